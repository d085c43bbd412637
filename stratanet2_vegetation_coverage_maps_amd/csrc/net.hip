// net.hip -- the launch sequences of the whole network behind one C-ABI call per pass (include/strata_hip.h, "the whole
// network behind ONE call per pass").  Host code only: it builds the descriptors of the entry points (sn2_sa, sn2_fp, sn2_head)
// exactly as the Python host does (hip_ops.sa_desc / fp_desc / head_desc, PointNet2._forward_impl / _backward_impl /
// _geometry) and calls those entry points in the same order.  Replaces the per-call host sequence behind PointNet2.forward
// (/root/reference/model/point_net2.py:106-153) and loss.backward() through it (/root/reference/learning/train.py:64): the
// reference's loop as written was bound by the ~60 Python -> C transitions of a step, not by the device.
#include "common.h"

#include <cstring>
#include <new>

namespace {

constexpr int WIDTHS[7] = {16, 16, 32, 64, 64, 34, 34};      // cout of sa1[0], sa1[1], sa2, sa3, fp3, fp2, fp1
constexpr int WIDTH_SUM = 260;
constexpr int GRAD_IMAGES = 32;                              // hip_ops.GRAD_IMAGES
constexpr int GL_MAX_PLOTS_HOST = 28;                        // sn2_global_level_forward's limit (fp.hip: GL_MAX_PLOTS)

// events of a forked geometry pass.  Chain b (level 2): FPS, ball query, work items [b_tables: all SA2 needs] -> the two small
// 3-NN tables [b_nn: all FP3 / FP2 need] -> their inverted indices [b_done].  Chain c: the per-point 3-NN table [c_nn: all FP1
// needs] -> its inverted index [c_done].  The inverted indices are read by the backward pass only, so a forward pass that
// consumes a deferred join waits for each table where it first reads it and for b_done / c_done at its very end.
struct NetCtx {
    hipEvent_t fork, b_tables, b_nn, b_done, c_nn, c_done, pack_fork, packed;
};

inline long rows_stat_limit() { return 64L * SN2_STAT_SLOTS; }

inline bool fps_fills_ws(int B, int N, int m) {              // hip_ops.fps_fills_ws == the condition inside sn2_fps_status
    const bool many_small = N <= 4096 && B > 32;
    return N > 2048 && !many_small && m > 16 && ((long)B * N) % 4 == 0 && N <= 131072;
}
inline bool three_nn_uses_grid(int S, int T) { return S >= 128 && S <= 8192 && T > 2048; }
inline bool fp_src_ws_wanted(const sn2_net_model* m, long R, int cb, bool force) {   // hip_ops.fp_desc: d._src_ws
    return m->source_side && cb > 0 && cb <= 16 && cb % 4 == 0 && (R > rows_stat_limit() || force);
}

// bump allocator over a caller-owned arena; base == nullptr: the "pointers" are offsets
struct Carver {
    uintptr_t base;
    size_t off = 0;
    explicit Carver(void* b) : base(reinterpret_cast<uintptr_t>(b)) {}
    template <class T>
    T* take(size_t n_elems) {
        off = (off + 255) & ~(size_t)255;
        T* p = reinterpret_cast<T*>(base + off);
        off += n_elems * sizeof(T);
        return p;
    }
    size_t bytes() const { return (off + 255) & ~(size_t)255; }
};

int check_dims(const sn2_net_model* m, const sn2_net_dims* d) {
    if (!m || !d) return SN2_EINVAL;
    if (d->B <= 0 || d->N <= 0 || d->M1 <= 0 || d->M2 <= 0 || d->M1 > d->N || d->M2 > d->M1) return SN2_EINVAL;
    if (m->max_neighbors <= 0) return SN2_EINVAL;
    const int c1 = m->max_neighbors < d->N ? m->max_neighbors : d->N, c2 = m->max_neighbors < d->M1 ? m->max_neighbors : d->M1;
    if (d->cap1 != c1 || d->cap2 != c2) return SN2_EINVAL;
    if ((long)d->B * d->N >= (1L << 31) / 64) return SN2_ELIMIT;
    const int cin[7] = {11, 16, 19, 35, 96, 80, 42};
    const sn2_net_layer* L[7] = {&m->sa1[0], &m->sa1[1], &m->sa2, &m->sa3, &m->fp3, &m->fp2, &m->fp1};
    for (int i = 0; i < 7; ++i)
        if (L[i]->cin != cin[i] || L[i]->cout != WIDTHS[i]) return SN2_ELIMIT;     // the reference architecture only
    return 0;
}

int check_model_ptrs(const sn2_net_model* m) {
    const sn2_net_layer* L[7] = {&m->sa1[0], &m->sa1[1], &m->sa2, &m->sa3, &m->fp3, &m->fp2, &m->fp1};
    for (int i = 0; i < 7; ++i)
        if (!L[i]->W || !L[i]->b || !L[i]->gamma || !L[i]->beta || !L[i]->running_mean || !L[i]->running_var) return SN2_EINVAL;
    if (!m->lin1_W || !m->lin1_b || !m->lin2_W || !m->lin2_b) return SN2_EINVAL;
    return 0;
}

// ---- descriptors ------------------------------------------------------------------------------------------------------
struct GradDst {
    float* flat;          // image 0 of the flat gradient, or nullptr (forward: no gradient fields)
    int stride;
    int frozen;           // backward: the forward ran BatchNorm on its running statistics (sn2_block.frozen_stats)
};

// hip_ops.BlockBuffers.fill: block `k` (0..6 in model order) of the network
void fill_block(sn2_block* blk, const sn2_net_layer* L, int k, const sn2_net_act* a, GradDst g) {
    int aux_off = 0, st_off = 0;
    for (int i = 0; i < k; ++i) aux_off += 4 * WIDTHS[i], st_off += SN2_STAT_SLOTS * 2 * WIDTHS[i];
    const int c = L->cout;
    blk->cin = L->cin, blk->cout = c;
    blk->W = L->W, blk->b = L->b, blk->gamma = L->gamma, blk->beta = L->beta;
    blk->running_mean = L->running_mean, blk->running_var = L->running_var;
    float* aux = a->aux + aux_off;
    blk->a = aux, blk->c = aux + c, blk->mean = aux + 2 * c, blk->invstd = aux + 3 * c;
    blk->stat_slots = a->stats + st_off;
    blk->mma_bf16 = L->mma_bf16 ? 1 : 0;
    blk->num_batches_tracked = L->num_batches_tracked;
    blk->frozen_stats = g.frozen;
    if (g.flat) {
        blk->dW = g.flat + L->gW, blk->db = g.flat + L->gb, blk->dgamma = g.flat + L->ggamma, blk->dbeta = g.flat + L->gbeta;
        blk->grad_replicas = GRAD_IMAGES, blk->grad_replica_stride = g.stride;
    } else {
        blk->dW = blk->db = blk->dgamma = blk->dbeta = nullptr;
        blk->grad_replicas = 1, blk->grad_replica_stride = 0;
    }
}

inline float* aux_row(const sn2_net_act* a, int k, int row) {      // row 0..3 = a, c, mean, invstd of block k
    int off = 0;
    for (int i = 0; i < k; ++i) off += 4 * WIDTHS[i];
    return a->aux + off + row * WIDTHS[k];
}

// PointNet2._sa1_desc / _sa2_desc
void sa1_desc(sn2_sa* s, const sn2_net_model* m, const sn2_net_dims* d, const sn2_net_geo* g, const sn2_net_act* a,
              GradDst gd, const float* dout) {
    memset(s, 0, sizeof(*s));
    s->B = d->B, s->Nsrc = d->N, s->M = d->M1, s->cap = d->cap1, s->cf = 8, s->nl = 2;
    s->feat = g->rows0, s->feat_stride = 12, s->spos = g->rows0 + 8, s->spos_stride = 12;
    s->cpos = g->pos1_aos, s->nbr = g->nbr1, s->cnt = g->cnt1, s->total = g->tot1, s->order = g->ord1;
    fill_block(&s->blk[0], &m->sa1[0], 0, a, gd);
    fill_block(&s->blk[1], &m->sa1[1], 1, a, gd);
    s->ext = a->ext1, s->arg = a->arg1, s->out = a->x1;
    s->dout = dout, s->dfeat = nullptr;
}
void sa2_desc(sn2_sa* s, const sn2_net_model* m, const sn2_net_dims* d, const sn2_net_geo* g, const sn2_net_act* a,
              GradDst gd, const float* dout, float* dfeat) {
    memset(s, 0, sizeof(*s));
    s->B = d->B, s->Nsrc = d->M1, s->M = d->M2, s->cap = d->cap2, s->cf = 16, s->nl = 1;
    s->feat = a->x1, s->feat_stride = 16, s->spos = g->pos1_aos, s->spos_stride = 4;
    s->cpos = g->pos2_aos, s->nbr = g->nbr2, s->cnt = g->cnt2, s->total = g->tot2, s->order = g->ord2;
    fill_block(&s->blk[0], &m->sa2, 2, a, gd);
    s->ext = a->ext2, s->arg = a->arg2, s->out = a->x2;
    s->dout = dout, s->dfeat = dfeat;
}

// hip_ops.fp_desc, forward fields; the backward fields are added by the caller
struct FpIn {
    int k;                                    // block index 3..6
    const sn2_net_layer* L;
    int Rp, Sp, ca, cb;
    const float* src; int src_stride; const float *src_a, *src_c;
    const int* knn_idx; const float* knn_w;
    const float* skip; int skip_stride;
    void* h;
    float* src_ws;
    int act_bf16;
};
void fp_desc(sn2_fp* p, const sn2_net_dims* d, const sn2_net_act* a, const FpIn& in, GradDst gd) {
    memset(p, 0, sizeof(*p));
    p->B = d->B, p->R_per_plot = in.Rp, p->S_per_plot = in.Sp, p->ca = in.ca, p->cb = in.cb;
    p->src = in.src, p->src_stride = in.src_stride, p->src_a = in.src_a, p->src_c = in.src_c;
    p->knn_idx = in.knn_idx, p->knn_w = in.knn_w;
    p->skip = in.cb > 0 ? in.skip : nullptr, p->skip_stride = in.cb > 0 ? in.skip_stride : 0;
    fill_block(&p->blk, in.L, in.k, a, gd);
    if ((long)d->B * in.Rp > rows_stat_limit()) p->blk.mma_bf16 = 0;       // no bfloat16 kernel for that many rows
    p->h = static_cast<float*>(in.h), p->h_stride = (in.L->cout + 3) / 4 * 4;
    p->src_ws = in.src_ws;
    p->act_bf16 = in.act_bf16;
}

FpIn sa3_in(const sn2_net_model* m, const sn2_net_dims* d, const sn2_net_geo* g, const sn2_net_act* a) {
    FpIn in{};
    in.k = 3, in.L = &m->sa3, in.Rp = d->M2, in.Sp = d->M2, in.ca = 32, in.cb = 3;
    in.src = a->x2, in.src_stride = 32;
    in.skip = g->pos2_aos, in.skip_stride = 4;
    in.h = a->h_sa3;
    return in;
}
FpIn fp3_in(const sn2_net_model* m, const sn2_net_dims* d, const sn2_net_geo* g, const sn2_net_act* a) {
    FpIn in{};
    in.k = 4, in.L = &m->fp3, in.Rp = d->M2, in.Sp = 1, in.ca = 64, in.cb = 32;
    in.src = a->x3, in.src_stride = 64;
    in.knn_idx = g->knn3_idx, in.knn_w = g->knn3_w;
    in.skip = a->x2, in.skip_stride = 32;
    in.h = a->h3;
    return in;
}
FpIn fp2_in(const sn2_net_model* m, const sn2_net_dims* d, const sn2_net_geo* g, const sn2_net_act* a, float* src_ws) {
    FpIn in{};
    in.k = 5, in.L = &m->fp2, in.Rp = d->M1, in.Sp = d->M2, in.ca = 64, in.cb = 16;
    in.src = a->h3, in.src_stride = 64, in.src_a = aux_row(a, 4, 0), in.src_c = aux_row(a, 4, 1);
    in.knn_idx = g->knn2_idx, in.knn_w = g->knn2_w;
    in.skip = a->x1, in.skip_stride = 16;
    in.h = a->h2;
    in.src_ws = src_ws;
    return in;
}
FpIn fp1_in(const sn2_net_model* m, const sn2_net_dims* d, const sn2_net_geo* g, const sn2_net_act* a, float* src_ws) {
    FpIn in{};
    in.k = 6, in.L = &m->fp1, in.Rp = d->N, in.Sp = d->M1, in.ca = 34, in.cb = 8;
    in.src = a->h2, in.src_stride = 36, in.src_a = aux_row(a, 5, 0), in.src_c = aux_row(a, 5, 1);
    in.knn_idx = g->knn1_idx, in.knn_w = g->knn1_w;
    in.skip = g->rows0, in.skip_stride = 12;
    in.h = a->h1;
    in.src_ws = src_ws;
    in.act_bf16 = d->act_bf16;
    return in;
}

// hip_ops.head_desc
void head_desc(sn2_head* h, const sn2_net_model* m, const sn2_net_dims* d, const sn2_net_act* a, int training) {
    memset(h, 0, sizeof(*h));
    h->R = d->B * d->N, h->cin = 34, h->f_stride = 36;
    h->f = static_cast<const float*>(a->h1), h->fa = aux_row(a, 6, 0), h->fc = aux_row(a, 6, 1);
    h->W1 = m->lin1_W, h->b1 = m->lin1_b, h->W2 = m->lin2_W, h->b2 = m->lin2_b;
    h->coverages = a->cov, h->proba = a->proba;
    h->grad_replicas = 1, h->grad_replica_stride = 0;
    h->act_bf16 = d->act_bf16;
    if (training && a->drop_mask) {
        h->drop_mask = a->drop_mask;
        h->drop_scale = m->drop_p < 1.f ? 1.f / (1.f - m->drop_p) : 0.f;
    } else {
        h->drop_mask = nullptr, h->drop_scale = 1.f;
    }
}

#define NET_HIP(expr)                                         \
    do {                                                      \
        hipError_t e__ = (expr);                              \
        if (e__ != hipSuccess) return (int)e__;               \
    } while (0)

int inverted_tables(const sn2_net_dims* d, const sn2_net_geo* g, int which, void* st) {
    // which: 1 = the two small tables (chain b), 2 = the per-point table (chain c), 3 = all
    if (which == 3) {               // (a late build: the message totals an eval-mode geometry pass did not make)
        SN2_TRY(sn2_count_sum(g->cnt1, d->B * d->M1, g->tot1, st));
        SN2_TRY(sn2_count_sum(g->cnt2, d->B * d->M2, g->tot2, st));
    }
    if (which & 1) {
        if (!g->inv3 || !g->inv2) return SN2_EINVAL;
        SN2_TRY(sn2_interp_index_perm(g->knn3_idx, g->knn3_w, nullptr, nullptr, d->B, d->M2, 1, g->inv3, st));
        SN2_TRY(sn2_interp_index_perm(g->knn2_idx, g->knn2_w, nullptr, nullptr, d->B, d->M1, d->M2, g->inv2, st));
    }
    if (which & 2) {
        if (!g->inv1) return SN2_EINVAL;
        SN2_TRY(sn2_interp_index_perm(g->knn1_idx, g->knn1_w, g->pos1_aos, g->rank1, d->B, d->N, d->M1, g->inv1, st));
    }
    return 0;
}

int check_geo(const sn2_net_dims* d, const sn2_net_geo* g) {
    if (!g || !g->xyz || !g->idx1 || !g->pos1_soa || !g->pos1_aos || !g->nbr1 || !g->cnt1 || !g->tot1 || !g->ord1 || !g->idx2 ||
        !g->pos2_soa || !g->pos2_aos || !g->nbr2 || !g->cnt2 || !g->tot2 || !g->ord2 || !g->pos3 || !g->knn3_idx || !g->knn3_w ||
        !g->knn2_idx || !g->knn2_w || !g->knn1_idx || !g->knn1_w || !g->rows0)
        return SN2_EINVAL;
    return 0;
}
int check_geo_workspaces(const sn2_net_dims* d, const sn2_net_geo* g) {     // what only the geometry pass itself needs
    if (three_nn_uses_grid(d->M2, d->M1) && !g->nn_ws2) return SN2_EINVAL;
    if (three_nn_uses_grid(d->M1, d->N) && !g->nn_ws1) return SN2_EINVAL;
    return 0;
}

int three_nn_any(const float* src, int B, int S, const float* dst, int T, int k, int* idx, float* w, int* ws, void* st) {
    if (three_nn_uses_grid(S, T)) return sn2_three_nn_xy(src, B, S, dst, T, k, idx, w, ws, st);
    return sn2_three_nn(src, B, S, dst, T, k, idx, w, nullptr, nullptr, st);
}

// PointNet2._geometry
int geometry_impl(const sn2_net_model* m, const sn2_net_dims* d, const sn2_net_geo* g, const sn2_net_io* io, int flags,
                  hipStream_t cur) {
    const int B = d->B, N = d->N, M1 = d->M1, M2 = d->M2;
    NetCtx* ctx = static_cast<NetCtx*>(io->ctx);
    const bool fork = (flags & SN2_NET_FORK) && ctx && io->stream_b && io->stream_c;
    const bool inverted = flags & SN2_NET_INVERTED;
    const int* start0 = io->fps_start;
    const int* start1 = io->fps_start ? io->fps_start + B : nullptr;
    int* ws1 = fps_fills_ws(B, N, M1) ? g->ws1 : nullptr;
    int* ws2 = fps_fills_ws(B, M1, M2) ? g->ws2 : nullptr;
    int waves = 0;
    if (flags & SN2_NET_SHARED) waves = B > 32 ? m->fps_waves_many : m->fps_waves_shared;
    SN2_TRY(sn2_fps_status(g->xyz, B, N, M1, start0, g->idx1, g->pos1_soa, g->pos1_aos, ws1, waves, io->fps_status, cur));
    hipStream_t sb = cur, sc = cur;
    if (fork) {
        sb = (hipStream_t)io->stream_b, sc = (hipStream_t)io->stream_c;
        NET_HIP(hipEventRecord(ctx->fork, cur));
        NET_HIP(hipStreamWaitEvent(sb, ctx->fork, 0));
        NET_HIP(hipStreamWaitEvent(sc, ctx->fork, 0));
    }
    // (b) the level-2 chain
    SN2_TRY(sn2_fps_status(g->pos1_soa, B, M1, M2, start1, g->idx2, g->pos2_soa, g->pos2_aos, ws2, 0, io->fps_status, sb));
    // (the message totals are the counts of the batch-statistics BatchNorms: only a pass a backward may follow -- `inverted` --
    // needs them; an eval pass over hundreds of plots spent 0.2 ms per level in the one-workgroup sum.  A forward that finds
    // tables without them builds them with the inverted indices: inverted_tables)
    SN2_TRY(sn2_ball_query(g->pos1_soa, B, M1, g->pos2_soa, M2, m->r2_sq, d->cap2, g->nbr2, g->cnt2, inverted ? g->tot2 : nullptr, ws2, sb));
    SN2_TRY(sn2_sa_order(g->cnt2, B, M2, g->ord2, sb));
    if (fork) NET_HIP(hipEventRecord(ctx->b_tables, sb));
    SN2_TRY(sn2_three_nn(g->pos3, B, 1, g->pos2_soa, M2, 1, g->knn3_idx, g->knn3_w, nullptr, nullptr, sb));
    SN2_TRY(three_nn_any(g->pos2_soa, B, M2, g->pos1_soa, M1, 3, g->knn2_idx, g->knn2_w, g->nn_ws2, sb));
    if (fork) NET_HIP(hipEventRecord(ctx->b_nn, sb));
    if (inverted) SN2_TRY(inverted_tables(d, g, 1, sb));
    // (c) the per-point table
    SN2_TRY(three_nn_any(g->pos1_soa, B, M1, g->xyz, N, 3, g->knn1_idx, g->knn1_w, g->nn_ws1, sc));
    if (fork) NET_HIP(hipEventRecord(ctx->c_nn, sc));
    if (inverted) SN2_TRY(inverted_tables(d, g, 2, sc));
    // the input-only pieces of the feature pass
    if (flags & SN2_NET_INPUT_ONLY) {
        if (!io->cloud) return SN2_EINVAL;
        SN2_TRY(sn2_pack_rows(io->cloud, g->xyz, B, 10, N, g->rows0, cur));
        if (d->p2_diam_pix > 0 && g->p2_pix && g->p2_mm)
            SN2_TRY(sn2_plot_pixels(io->cloud, 10L * N, B, N, d->p2_diam_pix, g->p2_mm, g->p2_pix, cur));
    }
    // (a)
    SN2_TRY(sn2_ball_query(g->xyz, B, N, g->pos1_soa, M1, m->r1_sq, d->cap1, g->nbr1, g->cnt1, inverted ? g->tot1 : nullptr, ws1, cur));
    SN2_TRY(sn2_sa_order(g->cnt1, B, M1, g->ord1, cur));
    if (fork) {
        NET_HIP(hipEventRecord(ctx->b_done, sb));
        NET_HIP(hipEventRecord(ctx->c_done, sc));
        if (!(flags & SN2_NET_DEFER_JOIN)) {
            NET_HIP(hipStreamWaitEvent(cur, ctx->b_done, 0));
            NET_HIP(hipStreamWaitEvent(cur, ctx->c_done, 0));
        }
    }
    return 0;
}

}  // namespace

extern "C" int sn2_net_ctx_create(void** out) {
    if (!out) return SN2_EINVAL;
    NetCtx* c = new (std::nothrow) NetCtx();
    if (!c) return SN2_EINVAL;
    hipEvent_t* ev[8] = {&c->fork, &c->b_tables, &c->b_nn, &c->b_done, &c->c_nn, &c->c_done, &c->pack_fork, &c->packed};
    for (int i = 0; i < 8; ++i) {
        hipError_t e = hipEventCreateWithFlags(ev[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(*ev[j]);
            delete c;
            return (int)e;
        }
    }
    *out = c;
    return 0;
}

extern "C" int sn2_net_ctx_destroy(void* ctx) {
    if (!ctx) return SN2_EINVAL;
    NetCtx* c = static_cast<NetCtx*>(ctx);
    (void)hipEventDestroy(c->fork);
    (void)hipEventDestroy(c->b_tables);
    (void)hipEventDestroy(c->b_nn);
    (void)hipEventDestroy(c->b_done);
    (void)hipEventDestroy(c->c_nn);
    (void)hipEventDestroy(c->c_done);
    (void)hipEventDestroy(c->pack_fork);
    (void)hipEventDestroy(c->packed);
    delete c;
    return 0;
}

extern "C" int sn2_net_geo_carve(const sn2_net_model* m, const sn2_net_dims* d, void* base, sn2_net_geo* g, size_t* bytes) {
    SN2_TRY(check_dims(m, d));
    if (!g || !bytes) return SN2_EINVAL;
    const size_t B = d->B, N = d->N, M1 = d->M1, M2 = d->M2;
    memset(g, 0, sizeof(*g));
    Carver c(base);
    g->idx1 = c.take<int>(B * M1), g->pos1_soa = c.take<float>(B * 3 * M1), g->pos1_aos = c.take<float>(B * M1 * 4);
    g->ws1 = fps_fills_ws(d->B, d->N, d->M1) ? c.take<int>((size_t)SN2_FPS_WS_WORDS(B, N)) : nullptr;
    g->nbr1 = c.take<int>(B * M1 * (size_t)d->cap1), g->cnt1 = c.take<int>(B * M1);
    g->tot1 = c.take<unsigned long long>(1), g->ord1 = c.take<int>(SN2_SA_ORDER_WORDS(B, M1));
    g->idx2 = c.take<int>(B * M2), g->pos2_soa = c.take<float>(B * 3 * M2), g->pos2_aos = c.take<float>(B * M2 * 4);
    g->ws2 = fps_fills_ws(d->B, d->M1, d->M2) ? c.take<int>((size_t)SN2_FPS_WS_WORDS(B, M1)) : nullptr;
    g->nbr2 = c.take<int>(B * M2 * (size_t)d->cap2), g->cnt2 = c.take<int>(B * M2);
    g->tot2 = c.take<unsigned long long>(1), g->ord2 = c.take<int>(SN2_SA_ORDER_WORDS(B, M2));
    g->knn3_idx = c.take<int>(B * M2 * 3), g->knn3_w = c.take<float>(B * M2 * 3);
    g->knn2_idx = c.take<int>(B * M1 * 3), g->knn2_w = c.take<float>(B * M1 * 3);
    g->knn1_idx = c.take<int>(B * N * 3), g->knn1_w = c.take<float>(B * N * 3);
    g->inv3 = c.take<float>(SN2_INTERP_WS_WORDS(B, M2, 1));
    g->inv2 = c.take<float>(SN2_INTERP_WS_WORDS(B, M1, M2));
    g->inv1 = c.take<float>(SN2_INTERP_WS_WORDS(B, N, M1));
    g->nn_ws2 = three_nn_uses_grid(d->M2, d->M1) ? c.take<int>(SN2_THREE_NN_XY_WS_WORDS(B, M2, M1)) : nullptr;
    g->nn_ws1 = three_nn_uses_grid(d->M1, d->N) ? c.take<int>(SN2_THREE_NN_XY_WS_WORDS(B, M1, N)) : nullptr;
    g->rows0 = c.take<float>(B * N * 12);
    if (d->p2_diam_pix > 0) g->p2_pix = c.take<int>(B * N), g->p2_mm = c.take<float>(B * 4);
    // xyz, pos3 (a constant zero vector) and rank1 (a view of ws1) are the caller's
    *bytes = c.bytes();
    return 0;
}

extern "C" int sn2_net_act_carve(const sn2_net_model* m, const sn2_net_dims* d, int training, void* base, sn2_net_act* a,
                                 size_t* bytes) {
    SN2_TRY(check_dims(m, d));
    if (!a || !bytes) return SN2_EINVAL;
    const size_t B = d->B, N = d->N, M1 = d->M1, M2 = d->M2;
    memset(a, 0, sizeof(*a));
    Carver c(base);
    a->aux = c.take<float>(4 * WIDTH_SUM), a->stats = c.take<float>((size_t)SN2_STAT_SLOTS * 2 * WIDTH_SUM);
    a->ext1 = c.take<float>(B * M1 * 16), a->arg1 = c.take<int>(B * M1 * 16), a->x1 = c.take<float>(B * M1 * 16);
    a->ext2 = c.take<float>(B * M2 * 32), a->arg2 = c.take<int>(B * M2 * 32), a->x2 = c.take<float>(B * M2 * 32);
    a->h_sa3 = c.take<float>(B * M2 * 64), a->h3 = c.take<float>(B * M2 * 64);
    a->x3 = c.take<float>(B * 64), a->arg3 = c.take<int>(B * 64);
    a->h2 = c.take<float>(B * M1 * 36);
    const bool fused_eval = !training && m->fuse_eval_head && !d->act_bf16 && m->source_side;
    if (!fused_eval) a->h1 = d->act_bf16 ? (void*)c.take<unsigned short>(B * N * 36) : (void*)c.take<float>(B * N * 36);
    if (fp_src_ws_wanted(m, (long)B * N, 8, fused_eval)) a->src_ws1 = c.take<float>(SN2_FP_SRC_WS_WORDS(B, N, M1, 34));
    if (fp_src_ws_wanted(m, (long)B * M1, 16, false)) a->src_ws2 = c.take<float>(SN2_FP_SRC_WS_WORDS(B, M1, M2, 34));
    *bytes = c.bytes();
    return 0;
}

extern "C" int sn2_net_bwd_carve(const sn2_net_model* m, const sn2_net_dims* d, void* arena_base, void* scratch_base,
                                 sn2_net_bwd* b, size_t* arena_bytes, size_t* scratch_bytes) {
    SN2_TRY(check_dims(m, d));
    if (!b || !arena_bytes || !scratch_bytes || m->n_flat <= 0) return SN2_EINVAL;
    const size_t B = d->B, N = d->N, M1 = d->M1, M2 = d->M2;
    memset(b, 0, sizeof(*b));
    // the zero-filled arena: hip_ops.grad_images_alloc + PointNet2._grad_arena (floats, every buffer a multiple of 4)
    const size_t stride = ((size_t)m->n_flat + 63) / 64 * 64;
    float* ar = static_cast<float*>(arena_base);
    size_t o = GRAD_IMAGES * stride;
    auto take = [&](size_t n) {
        float* p = ar + o;
        o += (n + 3) / 4 * 4;
        return p;
    };
    b->arena = ar, b->images = GRAD_IMAGES, b->image_stride = (int)stride;
    b->dy2 = take(B * M1 * 36), b->dy3 = take(B * M2 * 64), b->dx1 = take(B * M1 * 16), b->dx2 = take(B * M2 * 32);
    b->dx3 = take(B * 64), b->dy_sa3 = take(B * M2 * 64);
    b->arena_words = (long)o;
    *arena_bytes = o * sizeof(float);
    Carver c(scratch_base);
    if (d->act_bf16) {
        b->dy1 = c.take<unsigned short>(B * N * 36), b->du1 = c.take<unsigned short>(B * N * 36);
    } else {
        b->dy1 = c.take<float>(B * N * 36), b->du1 = c.take<float>(B * N * 36);
    }
    b->du2 = c.take<float>(B * M1 * 64), b->du3 = c.take<float>(B * M2 * 64);
    b->bn_ok = c.take<int>(4);
    if (fp_src_ws_wanted(m, (long)B * N, 8, false)) b->src_ws1 = c.take<float>(SN2_FP_SRC_WS_WORDS(B, N, M1, 34));
    if (fp_src_ws_wanted(m, (long)B * M1, 16, false)) b->src_ws2 = c.take<float>(SN2_FP_SRC_WS_WORDS(B, M1, M2, 34));
    *scratch_bytes = c.bytes();
    return 0;
}

extern "C" int sn2_net_geometry(const sn2_net_model* m, const sn2_net_dims* d, const sn2_net_geo* g, const sn2_net_io* io,
                                void* stream) {
    SN2_TRY(check_dims(m, d));
    if (!io) return SN2_EINVAL;
    SN2_TRY(check_geo(d, g));
    SN2_TRY(check_geo_workspaces(d, g));
    if ((io->flags & SN2_NET_INVERTED) && (!g->inv1 || !g->inv2 || !g->inv3)) return SN2_EINVAL;
    return geometry_impl(m, d, g, io, io->flags, (hipStream_t)stream);
}

// PointNet2._forward_impl
extern "C" int sn2_net_forward(const sn2_net_model* m, const sn2_net_dims* d, const sn2_net_geo* g, const sn2_net_act* a,
                               const sn2_net_io* io, void* stream) {
    SN2_TRY(check_dims(m, d));
    SN2_TRY(check_model_ptrs(m));
    if (!io || !a) return SN2_EINVAL;
    SN2_TRY(check_geo(d, g));
    if (!a->aux || !a->stats || !a->ext1 || !a->arg1 || !a->x1 || !a->ext2 || !a->arg2 || !a->x2 || !a->h_sa3 || !a->h3 || !a->x3 ||
        !a->arg3 || !a->h2 || !a->cov || !a->proba)
        return SN2_EINVAL;
    // mode: 0 eval, 1 training, SN2_BN_FROZEN_KEEP eval with a backward to come (the layers' own entry points take it as is);
    // `training` below = "a backward pass may follow" (tables, kept rows), `batch_stats` = model.training
    if (io->training < 0 || io->training > SN2_BN_FROZEN_KEEP) return SN2_EINVAL;
    const int mode = io->training, training = mode != 0 ? 1 : 0, batch_stats = mode == 1 ? 1 : 0, flags = io->flags;
    const int B = d->B, N = d->N, M2 = d->M2;
    hipStream_t cur = (hipStream_t)stream;
    NetCtx* ctx = static_cast<NetCtx*>(io->ctx);
    const bool fused_eval = !training && m->fuse_eval_head && !d->act_bf16 && m->source_side;
    if (!fused_eval && !a->h1) return SN2_EINVAL;
    if (training && !(g->inv1 && g->inv2 && g->inv3)) return SN2_EINVAL;
    const GradDst nograd{nullptr, 0, 0};
    bool join_b = false, join_c = false, wait_pack = false;
    bool have_rows0 = flags & SN2_NET_HAS_ROWS0;
    if (flags & SN2_NET_WITH_GEOMETRY) {
        if (!io->cloud) return SN2_EINVAL;
        SN2_TRY(check_geo_workspaces(d, g));
        int gflags = (flags & (SN2_NET_FORK | SN2_NET_SHARED)) | (training ? SN2_NET_INVERTED : 0);
        const bool fork = (flags & SN2_NET_FORK) && ctx && io->stream_b && io->stream_c;
        if (fork) {
            gflags |= SN2_NET_DEFER_JOIN;
            if (io->stream_pack && !have_rows0) {
                // the row packing needs the inputs only: beside the level-1 FPS instead of behind it
                hipStream_t sp = (hipStream_t)io->stream_pack;
                NET_HIP(hipEventRecord(ctx->pack_fork, cur));
                NET_HIP(hipStreamWaitEvent(sp, ctx->pack_fork, 0));
                SN2_TRY(sn2_pack_rows(io->cloud, g->xyz, B, 10, N, g->rows0, sp));
                NET_HIP(hipEventRecord(ctx->packed, sp));
                wait_pack = have_rows0 = true;
            }
        } else {
            gflags &= ~SN2_NET_FORK;
        }
        SN2_TRY(geometry_impl(m, d, g, io, gflags, cur));
        join_b = join_c = fork;
    } else {
        if (flags & SN2_NET_JOIN_PENDING) {
            if (!ctx) return SN2_EINVAL;
            join_b = join_c = true;
        }
        if (training && !(flags & SN2_NET_HAS_INVERTED)) {
            // tables prefetched in eval mode, forward in training mode: the backward pass needs the inverted indices
            if (join_b) NET_HIP(hipStreamWaitEvent(cur, ctx->b_done, 0));
            if (join_c) NET_HIP(hipStreamWaitEvent(cur, ctx->c_done, 0));
            join_b = join_c = false;
            SN2_TRY(inverted_tables(d, g, 3, cur));
        }
    }
    // ---- level 0 rows: [8 features | x y z 0]
    if (wait_pack) NET_HIP(hipStreamWaitEvent(cur, ctx->packed, 0));
    if (!have_rows0) {
        if (!io->cloud) return SN2_EINVAL;
        SN2_TRY(sn2_pack_rows(io->cloud, g->xyz, B, 10, N, g->rows0, cur));
    }
    // ---- SA1, SA2                                                                           (point_net2.py:131-132, 21-29)
    sn2_sa sa;
    sa1_desc(&sa, m, d, g, a, nograd, nullptr);
    SN2_TRY(sn2_sa_forward(&sa, mode, cur));
    if (join_b) NET_HIP(hipStreamWaitEvent(cur, ctx->b_tables, 0));     // chain b: the level-2 centroids, lists and work items
    sa2_desc(&sa, m, d, g, a, nograd, nullptr, nullptr);
    SN2_TRY(sn2_sa_forward(&sa, mode, cur));
    if (join_b) NET_HIP(hipStreamWaitEvent(cur, ctx->b_nn, 0));         // ... its two small 3-NN tables
    // ---- global level: SA3 -> plot max -> FP3                                                (:133-137, 37-42, 62-67)
    sn2_fp p3, pf3;
    fp_desc(&p3, d, a, sa3_in(m, d, g, a), nograd);
    fp_desc(&pf3, d, a, fp3_in(m, d, g, a), nograd);
    if (batch_stats && m->fuse_global_level && io->gl_xchg && io->gl_ctl && B <= GL_MAX_PLOTS_HOST && !m->sa3.mma_bf16 &&
        !m->fp3.mma_bf16) {
        SN2_TRY(sn2_global_level_forward(&p3, &pf3, a->x3, a->arg3, io->gl_xchg, io->gl_ctl, cur));
    } else {
        SN2_TRY(sn2_fp_forward(&p3, mode, cur));
        SN2_TRY(sn2_plot_max_forward(a->h_sa3, p3.blk.a, p3.blk.c, B, M2, 64, a->x3, a->arg3, cur));
        SN2_TRY(sn2_fp_forward(&pf3, mode, cur));
    }
    // ---- FP2                                                                                 (:138)
    sn2_fp p2;
    fp_desc(&p2, d, a, fp2_in(m, d, g, a, a->src_ws2), nograd);
    SN2_TRY(sn2_fp_forward(&p2, mode, cur));
    if (join_c) NET_HIP(hipStreamWaitEvent(cur, ctx->c_nn, 0));         // chain c: the per-point 3-NN table
    // ---- FP1 + head                                                                          (:139-151)
    sn2_fp p1;
    sn2_head hd;
    if (fused_eval) {
        if (!a->src_ws1) return SN2_EINVAL;
        sn2_net_act a2 = *a;
        a2.h1 = nullptr;
        fp_desc(&p1, d, &a2, fp1_in(m, d, g, &a2, a->src_ws1), nograd);
        head_desc(&hd, m, d, &a2, 0);
        SN2_TRY(sn2_fp_head_eval(&p1, &hd, cur));
    } else {
        fp_desc(&p1, d, a, fp1_in(m, d, g, a, a->src_ws1), nograd);
        SN2_TRY(sn2_fp_forward(&p1, mode, cur));
        head_desc(&hd, m, d, a, batch_stats);            // (dropout: model.training only)
        if (training && a->bwd_arena && a->bwd_arena_words > 0) hd.zero_fill = a->bwd_arena, hd.zero_fill_words = a->bwd_arena_words;
        SN2_TRY(sn2_head_forward(&hd, cur));
    }
    // the chains' ends (the inverted indices: the backward pass reads them on this stream; and the side streams' last reads of
    // the tables lie in front of whatever the caller does with the buffers next)
    if (join_b) NET_HIP(hipStreamWaitEvent(cur, ctx->b_done, 0));
    if (join_c) NET_HIP(hipStreamWaitEvent(cur, ctx->c_done, 0));
    return 0;
}

// PointNet2._backward_impl
extern "C" int sn2_net_backward(const sn2_net_model* m, const sn2_net_dims* d, const sn2_net_geo* g, const sn2_net_act* a,
                                const sn2_net_bwd* b, void* stream) {
    SN2_TRY(check_dims(m, d));
    SN2_TRY(check_model_ptrs(m));
    if (!a || !b) return SN2_EINVAL;
    SN2_TRY(check_geo(d, g));
    if (!g->inv1 || !g->inv2 || !g->inv3) return SN2_EINVAL;
    if (!a->aux || !a->stats || !a->h1 || !a->h2 || !a->h3 || !a->h_sa3 || !a->x3 || !a->arg3) return SN2_EINVAL;
    if (!b->arena || b->images != GRAD_IMAGES || b->image_stride < m->n_flat || !b->dy1 || !b->du1 || !b->du2 || !b->du3 || !b->bn_ok ||
        !b->dy2 || !b->dy3 || !b->dx1 || !b->dx2 || !b->dx3 || !b->dy_sa3 || b->arena_words <= 0)
        return SN2_EINVAL;
    const int B = d->B, M2 = d->M2;
    hipStream_t cur = (hipStream_t)stream;
    // one zero-filled arena: the images of the flat parameter gradient + every accumulate-into buffer of the chain
    if (!b->arena_is_zero) {
        sn2_fill_words(b->arena, 0u, (size_t)b->arena_words, cur);
        NET_HIP(hipGetLastError());
    }
    const GradDst gd{b->arena, b->image_stride, b->frozen_stats ? 1 : 0};
    // head
    sn2_head hd;
    head_desc(&hd, m, d, a, b->frozen_stats ? 0 : 1);       // (no dropout in an eval-mode forward)
    hd.coverages = hd.proba = nullptr;
    hd.dcoverages = b->dcov, hd.dproba = b->dproba, hd.dy = static_cast<float*>(b->dy1);
    hd.dW1 = b->arena + m->g_lin1_W, hd.db1 = b->arena + m->g_lin1_b, hd.dW2 = b->arena + m->g_lin2_W, hd.db2 = b->arena + m->g_lin2_b;
    hd.grad_replicas = GRAD_IMAGES, hd.grad_replica_stride = b->image_stride;
    SN2_TRY(sn2_head_backward(&hd, cur));
    // FP1's BatchNorm gradients fall out of lin1's (no extra pass over the B*N rows)
    SN2_TRY(sn2_head_bn_sums(&hd, m->fp1.gamma, m->fp1.beta, aux_row(a, 6, 2), aux_row(a, 6, 3), b->arena + m->fp1.ggamma,
                             b->arena + m->fp1.gbeta, b->bn_ok + 0, cur));
    // FP1 -> d(fp2 output)
    sn2_fp p1;
    fp_desc(&p1, d, a, fp1_in(m, d, g, a, b->src_ws1), gd);
    p1.dy = static_cast<const float*>(b->dy1), p1.dsrc = b->dy2, p1.dsrc_stride = 36, p1.dskip = nullptr, p1.dskip_stride = 0;
    p1.du_scratch = static_cast<float*>(b->du1), p1.scatter_ws = g->inv1, p1.scatter_ready = 1, p1.bn_sums_done = b->bn_ok + 0;
    p1.row_perm = (g->rank1 && p1.src_ws) ? g->rank1 : nullptr;
    SN2_TRY(sn2_fp_backward(&p1, cur));
    SN2_TRY(sn2_fp_bn_sums(&p1, m->fp2.gamma, m->fp2.beta, aux_row(a, 5, 2), aux_row(a, 5, 3), b->arena + m->fp2.ggamma,
                           b->arena + m->fp2.gbeta, b->bn_ok + 1, cur));
    // FP2 -> d(fp3 output), d x1
    sn2_fp p2;
    fp_desc(&p2, d, a, fp2_in(m, d, g, a, b->src_ws2), gd);
    p2.dy = b->dy2, p2.dsrc = b->dy3, p2.dsrc_stride = 64, p2.dskip = b->dx1, p2.dskip_stride = 16;
    p2.du_scratch = b->du2, p2.scatter_ws = g->inv2, p2.scatter_ready = 1, p2.bn_sums_done = b->bn_ok + 1;
    SN2_TRY(sn2_fp_backward(&p2, cur));
    SN2_TRY(sn2_fp_bn_sums(&p2, m->fp3.gamma, m->fp3.beta, aux_row(a, 4, 2), aux_row(a, 4, 3), b->arena + m->fp3.ggamma,
                           b->arena + m->fp3.gbeta, b->bn_ok + 2, cur));
    // FP3 -> d x2 and the per-row gradients of its interpolated part; the pool between FP3 and SA3 in one launch: d x3 = their sum
    // over the plot (k = 1 from the plot's one source), routed to the SA3 rows that attained the maximum, and SA3's BatchNorm sums
    // over those B x 64 entries
    sn2_fp pf3;
    fp_desc(&pf3, d, a, fp3_in(m, d, g, a), gd);
    pf3.dy = b->dy3, pf3.dsrc = b->dx3, pf3.dsrc_stride = 64, pf3.dskip = b->dx2, pf3.dskip_stride = 32;
    pf3.du_scratch = b->du3, pf3.scatter_ws = g->inv3, pf3.scatter_ready = -1, pf3.bn_sums_done = b->bn_ok + 2;
    SN2_TRY(sn2_fp_backward(&pf3, cur));
    SN2_TRY(sn2_global_pool_backward(b->du3, 64, a->arg3, a->h_sa3, aux_row(a, 3, 2), aux_row(a, 3, 3), B, M2, 64, b->dx3, b->dy_sa3,
                                     b->arena + m->sa3.ggamma, b->arena + m->sa3.gbeta, cur));
    sn2_fp p3;
    fp_desc(&p3, d, a, sa3_in(m, d, g, a), gd);
    p3.dy = b->dy_sa3, p3.dsrc = b->dx2, p3.dsrc_stride = 32, p3.bn_sums_done = b->bn_ok + 3;
    SN2_TRY(sn2_fp_backward(&p3, cur));
    // SA2 -> d x1 ; SA1
    sn2_sa sa;
    sa2_desc(&sa, m, d, g, a, gd, b->dx2, b->dx1);
    SN2_TRY(sn2_sa_backward(&sa, cur));
    sa1_desc(&sa, m, d, g, a, gd, b->dx1);
    SN2_TRY(sn2_sa_backward(&sa, cur));
    // the images of (dW, db) -> image 0 (unless the optimiser step folds them itself: sn2_adam_step_images)
    if (b->defer_grad_reduce) return 0;
    return sn2_grad_reduce(b->arena, m->n_flat, GRAD_IMAGES, b->image_stride, cur);
}
