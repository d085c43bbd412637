/* strata_hip.h -- C ABI of libstrata_hip.so: the MI355X (gfx950) implementation of the PointNet2 hot path of
 * IGNF/StrataNet2-Vegetation-Coverage-Maps.
 *
 * The reference has no FFI: its boundary is the Python API of model/point_net2.py and model/project_to_2d.py,
 * whose arithmetic lives in un-vendored wheels (torch-cluster 1.5.9, torch-geometric 1.7.2, torch-scatter 2.0.7).
 * Each entry point below names the reference call site (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (tensor.data_ptr()) unless it points to one of the structs below, which
 *     live in host memory and are only read during the call;
 *   - the caller owns every buffer including workspaces; nothing is allocated, freed or synchronised inside;
 *   - every call is asynchronous on `stream` (a hipStream_t; NULL = default stream);
 *   - fp32 data, int32 indices; indices are LOCAL to their plot (0..N-1) unless stated;
 *   - return value: 0 = ok, >0 = hipError_t of a failed launch, <0 = argument error (SN2_E*), never throws.
 *
 * Layouts
 *   "SoA"   (B,3,N)   x-row, y-row, z-row per plot          (what the reference DataLoader hands over as `xyz`)
 *   "AoS4"  (B*N,4)   x,y,z,0 per point
 *   rows    (R,C)     row-major feature rows
 */
#ifndef STRATA_HIP_H
#define STRATA_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SN2_VERSION 100
#define SN2_EINVAL (-1) /* bad size / null pointer             */
#define SN2_ELIMIT (-2) /* size outside what the kernels cover */

#define SN2_MAX_NEIGHBORS 2000 /* model/point_net2.py:24  max_num_neighbors */
#define SN2_STAT_SLOTS 1024    /* workgroups (= partial-sum slots) of a kernel that produces BatchNorm statistics */

int sn2_version(void);
/* diagnostic (tests/test_gpu_bf16.py): D (16,16) = A (16,48) B (48,16) with bfloat16 operands through a K = 32 and a K = 16 MFMA
 * chained on one accumulator -- modes 0..4: csrc/misc.hip, debug_mfma_chain_kernel */
int sn2_debug_mfma_chain(const float *a, const float *b, float *d, int mode, void *stream);
/* diagnostic: `blocks` one-wave workgroups idle for `clocks` shader clocks each (s_sleep loop, bounded); out: NULL or one int */
int sn2_debug_spin(int blocks, long long clocks, int *out, void *stream);

/* measured peaks (bench.py times them with HIP events in the same run as the metric, next to the datasheet figures):
 * sn2_debug_stream_probe: mode 0 = copy dst[i] = src[i] over n_floats floats (2 * 4 * n_floats bytes moved), 1 = read-only sum
 * (4 * n_floats bytes); sink: SN2_PROBE_SINK_WORDS floats of scratch.  sn2_debug_mfma_probe: independent chains of one matrix
 * instruction, 8 workgroups of 4 waves per CU, `iters` rounds -- mode 0 v_mfma_f32_16x16x4_f32, 1 v_mfma_f32_32x32x2_f32,
 * 2 v_mfma_f32_16x16x32_bf16, 3 v_mfma_f32_32x32x16_bf16; *flops (HOST pointer or NULL) = the flops of the launch. */
/* diagnostic (bench.py's roofline record): which kernels of the per-point layer's source-side backward sn2_fp_backward launches --
 * bit 0 the row pass, 1 the source pass, 2 the merge; 7 (default; 0 = back to it) = all, the only setting that yields gradients */
int sn2_debug_fp1_backward_parts(int mask);
/* test hook: which row pass the source-side FORWARD of the per-point layer launches -- 1 (default) the form whose input stream is
 * fetched one element per lane, four iterations ahead (fp_fwd_rows2_kernel, round 5), 0 the first form (every lane of a row loads
 * the row's inputs itself); the two must give the same bits */
int sn2_debug_fp_rows_form(int form);
/* test hook: which kernel builds the source table of the source-side forms -- 1 (default) on the matrix cores (64 rows per wave
 * through an LDS tile, round 5), 0 one row per lane with scalar weights; the fp32 MFMA accumulates k ascending like the fmaf chain */
int sn2_debug_fp_table_form(int form);
#define SN2_PROBE_SINK_WORDS 4096
int sn2_debug_stream_probe(const float *src, float *dst, size_t n_floats, int mode, float *sink, void *stream);
int sn2_debug_mfma_probe(int mode, int iters, float *sink, double *flops, void *stream);

/* ---- one (Linear -> ReLU -> BatchNorm1d) block, model/point_net2.py:45-53 -------------------------------- */
typedef struct sn2_block {
    int cin, cout;
    const float *W, *b;            /* Linear weight (cout,cin) row-major, bias (cout)                            */
    const float *gamma, *beta;     /* BatchNorm affine (cout)                                                   */
    float *running_mean, *running_var; /* (cout) updated in place when training (momentum 0.1, unbiased var)    */
    float *a, *c;                  /* out (cout): the block's output is  a*relu(W u + b) + c                    */
    float *mean, *invstd;          /* out (cout): batch statistics saved for backward (training)                */
    float *stat_slots;             /* workspace, SN2_STAT_SLOTS * 2 * cout floats (per-workgroup batch-statistics
                                      partials; no initialisation needed)                                       */
    float *dW, *db, *dgamma, *dbeta; /* backward outputs, ACCUMULATED; must be ZERO on entry of the backward call
                                        (dgamma/dbeta are read back inside it)                                   */
    int grad_replicas, grad_replica_stride; /* > 1: (dW, db) exist as grad_replicas zeroed images, image r at
                                        + r * grad_replica_stride floats; a workgroup adds into ONE image, so that the
                                        float atomics spread over the memory channels.  The gradient is the sum of the
                                        images (sn2_grad_reduce).  dgamma / dbeta always have one image.  0 or 1: one. */
    int mma_bf16;                  /* != 0: this block's contractions on the matrix cores (forward, input gradient, weight
                                      gradient) take bfloat16 operands (weights and activations rounded to nearest even,
                                      v_mfma_f32_16x16x32_bf16 / 16x16x16, fp32 accumulate); ReLU, BatchNorm, statistics, the
                                      extremum and everything that decides an index stay fp32 (BASELINE.json configs[4]).
                                      0: exact fp32 products (v_mfma_f32_16x16x4_f32), the reference's precision.          */
    long long *num_batches_tracked; /* BatchNorm1d's int64 counter (one element) or NULL: += 1 by the training forward,
                                       inside the statistics finalisation (no launch of its own)                           */
    int frozen_stats;              /* read by the BACKWARD entry points.  != 0: the forward pass ran this BatchNorm on its RUNNING
                                      statistics (a forward with training = SN2_BN_FROZEN_KEEP: model.eval() with gradients
                                      wanted, model/point_net2.py:45-53 under torch's autograd), so mean / invstd are constants of
                                      the backward: d pre-BN = gamma * invstd * dy, without the batch-mean and batch-variance
                                      terms; dgamma / dbeta are the same sums.  0: batch statistics (a training forward).           */
} sn2_block;
/* the `training` argument of sn2_sa_forward / sn2_fp_forward / sn2_net_io.training:
 *   0                   eval: BatchNorm on its running statistics, nothing kept for a backward pass (fewest bytes and launches);
 *   1                   training: batch statistics, running statistics and counters updated, everything kept;
 *   SN2_BN_FROZEN_KEEP  eval WITH a backward to come: running statistics (not updated), and everything a backward pass reads is
 *                       kept exactly as a training forward keeps it (the kernels of a training pass, the finalisation of an eval
 *                       pass); the backward calls must then carry sn2_block.frozen_stats = 1. */
#define SN2_BN_FROZEN_KEEP 2
/* flat[i] += sum_{r=1..replicas-1} flat[r*stride + i], i < n: folds the images of a flat gradient vector into image 0 */
int sn2_grad_reduce(float *flat, int n, int replicas, int stride, void *stream);

/* ---- geometry -------------------------------------------------------------------------------------------- */

/* (cloud (B,C,N), xyz (B,3,N)) -> rows0 (B*N,12) = [cloud rows 2..9 | x y z 0]
 * replaces the long-form/concat glue of PointNet2.forward, model/point_net2.py:107-124 (C must be 10). */
int sn2_pack_rows(const float *cloud, const float *xyz, int B, int C, int N, float *rows0, void *stream);

/* farthest point sampling -- torch_cluster.fps, model/point_net2.py:22.
 * start (B) local start index per plot or NULL (= 0, i.e. random_start=False).
 * idx (B,M) local indices in selection order; cpos_soa (B,3,M) and cpos_aos (B*M,4) = the selected positions.
 * order_ws: workspace of SN2_FPS_WS_WORDS(B,N) int32 (16-byte aligned, no initialisation: B*N ints of spatial order,
 * B*N float4 of sorted points, B cell-grid headers of 4104 words, B exchange areas of 4096 words for the
 * multi-workgroup kernel, 32 control words, B*N ints = every point's position in the spatial order [the inverse of the
 * first B*N words: what sn2_fp.row_perm takes]) enabling the bucketed kernel (exact same result, several times
 * faster at N = 32768), or NULL for the brute-force kernel.  After the call the workspace describes the sorted point
 * set and can be handed to sn2_ball_query over the same sources. */
#define SN2_FPS_WS_WORDS(B, N) (6L * (B) * (N) + (4104L + 4096L) * (B) + 32L)
int sn2_fps(const float *pos_soa, int B, int N, int M, const int *start, int *idx, float *cpos_soa,
            float *cpos_aos, int *order_ws, void *stream);
/* The same with a choice of kernel for the bucketed path.  waves = 0 (what sn2_fps passes): the shortest pass -- the
 * multi-workgroup kernel wherever the batch fits the chip (B * P workgroups resident at once), else one workgroup of 16 waves
 * per plot.  16 / 8: one workgroup of 16 / 8 waves per plot (8: a 9 % longer pass that leaves half of each occupied CU to
 * concurrent kernels: the setting of a pipelined loop where the pass runs beside another batch's feature kernels).
 * 4: four waves per plot, for passes over hundreds of plots of at most 16 384 points (two or more FPS workgroups per CU:
 * parcel inference); larger plots take 8.
 * 32 + P / 64 + P, P = 2, 4, 8: P workgroups of 16 / 8 waves per plot (each owns every P-th bucket of the plot's Morton order;
 * one exchange of tagged 8-byte granules through L2 per super-round; falls back to 16 when B * P workgroups do not fit).
 * 1: one sample per arg-max round (round 1's kernel, kept for cross-checks and timing comparisons).
 * Same indices whichever runs. */
int sn2_fps_waves(const float *pos_soa, int B, int N, int M, const int *start, int *idx, float *cpos_soa,
                  float *cpos_aos, int *order_ws, int waves, void *stream);
/* The same with a STATUS word.  The multi-workgroup kernel's workgroups wait for their peers, and HIP does not promise that
 * the B * P workgroups of a launch are resident together (kernels of other streams or processes may hold the CUs a peer
 * needs): every wait is bounded, a workgroup whose wait runs out leaves without writing, and a REPAIR launch that every
 * multi-workgroup pass is followed by (the single-workgroup kernel, same stream; it reads one control word and returns when
 * no wait gave up) samples all plots again.  The results are the reference's either way; *status (device, one 32-bit
 * word, caller-zeroed once, or NULL) accumulates the number of waits that gave up, so that a host that synchronises anyway
 * can tell that passes are being repeated (torch_cluster.fps has no such failure mode: model/point_net2.py:22). */
int sn2_fps_status(const float *pos_soa, int B, int N, int M, const int *start, int *idx, float *cpos_soa,
                   float *cpos_aos, int *order_ws, int waves, unsigned *status, void *stream);
/* tests only: sweeps a wait of the multi-workgroup FPS makes before it gives up (0 = the default, 2^18 ~ 0.2 s) */
int sn2_debug_fps_spin_limit(unsigned sweeps);

/* radius ball query -- torch_cluster.radius, model/point_net2.py:23-25.
 * For centroid i of plot b: all source points j of plot b with d2 < r2 (strict, canonical fp32 arithmetic; r2 = the
 * fp32 value of r*r evaluated in double, as torch_cluster compares),
 * ascending j, at most `cap` of them: nbr[(b*M+i)*cap + 0..cnt-1], cnt[b*M+i]. *total = sum of cnt (overwritten).
 * fps_ws: the workspace a bucketed sn2_fps call over the SAME sources left behind (cell lists: ~200 candidates per
 * centroid instead of N), or NULL for the full scan; same result either way. */
int sn2_ball_query(const float *src_soa, int B, int N, const float *cpos_soa, int M, float r2, int cap,
                   int *nbr, int *cnt, unsigned long long *total, const int *fps_ws, void *stream);

/* total = sum of cnt[0..n): the number of messages of a set of neighbour lists (what sn2_ball_query leaves in `total` for
 * the lists of one call; needed again when the lists of one call are handed on in parts). */
int sn2_count_sum(const int *cnt, int n, unsigned long long *total, void *stream);
/* the same for G consecutive lists of n counts each in one launch: total[h * total_stride] = sum of cnt[h*n .. (h+1)*n) */
int sn2_count_sum_group(const int *cnt, int G, int n, unsigned long long *total, size_t total_stride, void *stream);

/* k nearest sources (k = 1..3) + inverse squared distance weights -- the no_grad part of
 * torch_geometric.nn.knn_interpolate, model/point_net2.py:63.
 * idx (B*T,3), w (B*T,3): w = 1/max(d2,1e-16); unused slots (k<3 or S<k) get w = 0 and idx = idx[0].
 * ws + dst_fps_ws (optional, both or neither): ws = 16-byte aligned workspace of SN2_THREE_NN_WS_WORDS(B,S) 32-bit words,
 * dst_fps_ws = the workspace sn2_fps filled for the TARGET points (same B, N = T > 2048).  With them (and 128 <= S <= 8192)
 * a wave takes 64 spatially adjacent targets and searches a per-plot x,y grid of the sources instead of scanning all S
 * (same result, bit for bit); results are still written at the targets' original positions. */
#define SN2_THREE_NN_WS_WORDS(B, S) ((size_t)(B) * (4 * (size_t)(S) + 1032))
int sn2_three_nn(const float *src_soa, int B, int S, const float *dst_soa, int T, int k, int *idx, float *w,
                 void *ws, const int *dst_fps_ws, void *stream);
/* The same search with the targets sorted by the source grid's x,y cells inside the call (a counting sort, one workgroup
 * per plot): every wave then holds 64 targets of one or two adjacent cells, whatever order the targets come in.
 * 128 <= S <= 8192; ws = 16-byte aligned workspace of SN2_THREE_NN_XY_WS_WORDS(B,S,T) 32-bit words.  Same results as
 * sn2_three_nn, bit for bit. */
#define SN2_THREE_NN_XY_WS_WORDS(B, S, T) ((size_t)(B) * (4 * (size_t)(S) + 5 * (size_t)(T) + 1032))
int sn2_three_nn_xy(const float *src_soa, int B, int S, const float *dst_soa, int T, int k, int *idx, float *w,
                    void *ws, void *stream);

/* Input pipeline of a batch: load_cloud of the reference DataLoader, data_loader/loader.py:73-87 --
 * centre, append the fake ground points, keep xyz, [train: rotate about z, flip, add noise], rescale, gather the subsample --
 * one thread per output point.  raw (10,T): the B plots' raw points side by side (rows x, y, z, red, green, blue,
 * near_infrared, intensity, return_num, num_returns), offsets (B+1), centers (B,2); fake_xy (n_fake,2): the positions that
 * add_fake_empty_ground_points appends to every plot; idx (B,N): the subsample of each plot, values in
 * [0, raw points of the plot + n_fake); rot (B,2) fp64 = cos, sin of the rotation, flips (B,2); noise (6,noise_T) or NULL
 * = the clipped gaussian noise of x, y, red, green, blue, near_infrared for every point of every plot BEFORE subsampling
 * (fake points included), noise_offsets (B).  Outputs cloud (B,10,N), xyz (B,3,N).  All random draws are the caller's. */
int sn2_prepare_plots(const float *raw, long T, const int *offsets, const float *centers, const float *fake_xy, int n_fake,
                      const int *idx, int B, int N, int train, const double *rot, const int *flips, const float *noise,
                      const long *noise_offsets, long noise_T, float z_max, float *cloud, float *xyz, void *stream);

/* z-normalisation of a raw plot (offline preparation): z_i - min{ z_j : |xy_i - xy_j| <= radius } --
 * normalize_z_with_minz_in_a_radius, utils/load_data.py:237-249 (sklearn kd-tree radius query in x,y + a python loop).
 * x, y, z (n) fp32; the test is sklearn's: fp64 reduced distance dx*dx + dy*dy <= radius*radius, inclusive.
 * x_min..y_max: bounding box of the points (the caller has it from loading the plot).  ws: SN2_ZNORM_WS_WORDS(n, cells)
 * 32-bit words with cells = ((x_max-x_min)/radius + 2) * ((y_max-y_min)/radius + 2), 16-byte aligned.
 * zmin (n) and/or z_out (n) = fp32(z - zmin). */
#define SN2_ZNORM_WS_WORDS(n, cells) ((size_t)5 * (n) + 3 * (size_t)(cells) + 8)
int sn2_znorm(const float *x, const float *y, const float *z, int n, float radius, float x_min, float y_min, float x_max,
              float y_max, int *ws, float *zmin, float *z_out, void *stream);

/* ---- set abstraction: gather + shared MLP + BN + max -- SAModule/PointConv, model/point_net2.py:19,21-29 --- */
typedef struct sn2_sa {
    int B, Nsrc, M, cap;            /* plots, source points per plot, centroids per plot, stride of nbr        */
    int cf;                         /* feature channels of a source row (8 or 16); message = [feat | pos_j-pos_i] */
    int nl;                         /* blocks in local_nn: 1 or 2                                               */
    const float *feat; int feat_stride;  /* source feature rows (B*Nsrc, feat_stride)                           */
    const float *spos; int spos_stride;  /* source positions (x,y,z,.) rows                                      */
    const float *cpos;              /* centroid positions AoS4 (B*M,4)                                          */
    const int *nbr, *cnt;           /* from sn2_ball_query                                                      */
    const unsigned long long *total;/* number of messages E (device)                                            */
    const int *order;               /* from sn2_sa_order, or NULL: quads of consecutive centroids                      */
    sn2_block blk[2];
    float *ext; int *arg;           /* (B*M,cout): signed extremum of the last block's pre-BN activation and the
                                       neighbour slot attaining it (-1: no neighbours).  An EVAL forward keeps
                                       nothing for a backward: it leaves both arrays untouched (round 5: its kernel
                                       writes `out` itself)                                                     */
    float *out;                     /* (B*M,cout) = a*ext + c : the module output x                             */
    const float *dout;              /* backward in : d loss / d out (B*M,cout)                                  */
    float *dfeat;                   /* backward out: ACCUMULATED d loss / d feat (B*Nsrc,cf) or NULL            */
} sn2_sa;
/* Work items of the SA passes (position-only: part of the geometry pass).  A wave step is four 16-message tiles; a plot's
 * centroids are ranked by DESCENDING neighbour count (ties by ascending id -- deterministic) and cut into classes:
 *   SOLO (n > SN2_SA_SOLO_MIN: all four tiles, 64 messages per step), QUAD (n > SN2_SA_QUAD_MIN: ranks 4k..4k+3 share the
 *   steps, a tile each), OCT (n > SN2_SA_OCT_MIN: eight ranks, half a tile each, one step), HEX (the rest: sixteen ranks, a
 *   quarter tile each, one step).  (Ball sizes at C2: median 5, mean 25, maximum 261.)
 * Item k of plot b sits at position k*B + b of its table, i.e. heaviest first across plots.  order = SN2_SA_ORDER_WORDS(B,M)
 * ints (-1 = none): 4 B M ints of SOLO / QUAD positions, 4 each (solo: id | SN2_SA_SOLO_FLAG four times; quad: four ids);
 * 16 B SN2_SA_PACKED_ITEMS(M) ints of OCT / HEX positions, 16 each = the centroid of every quarter tile (an OCT's ids twice
 * each, | SN2_SA_OCT_FLAG); a trailer: [0] / [1] = the largest SOLO + QUAD / OCT + HEX item count of any plot. */
#define SN2_SA_SOLO_MIN 64
#define SN2_SA_QUAD_MIN 8
#define SN2_SA_OCT_MIN 4
#define SN2_SA_SOLO_FLAG 0x40000000
#define SN2_SA_OCT_FLAG 0x20000000
#define SN2_SA_PACKED_ITEMS(M) ((M) / 8 + 2)
#define SN2_SA_ORDER_WORDS(B, M) ((size_t)4 * (B) * (M) + (size_t)16 * (B) * SN2_SA_PACKED_ITEMS(M) + 8)
int sn2_sa_order(const int *cnt, int B, int M, int *order, void *stream);
/* G consecutive batches of B plots each in one launch pair: cnt (G*B*M), batch h's work items at order + h * stride_words
 * (stride_words >= SN2_SA_ORDER_WORDS(B,M)): what a geometry pass over several batches of a pipelined loop calls */
int sn2_sa_order_group(const int *cnt, int G, int B, int M, int *order, size_t stride_words, void *stream);
int sn2_sa_forward(const sn2_sa *p, int training, void *stream);
int sn2_sa_backward(const sn2_sa *p, void *stream);

/* ---- dense-row block with interpolated + skip inputs: FPModule / GlobalSAModule, model/point_net2.py:37-42,62-67
 * u_r = [ interp_r (ca) | skip_r (cb) ],  interp_r = sa*( sum_k w_k src[idx_k] / sum_k w_k ) + sc,
 * h = relu(W u + b) (R,cout) stored pre-BN; consumers apply (a,c).  knn_idx = NULL: interp_r = src row r. */
typedef struct sn2_fp {
    int B, R_per_plot, S_per_plot;  /* rows per plot (targets), source rows per plot                            */
    int ca, cb;                     /* interpolated channels, skip channels (cb may be 0)                       */
    const float *src; int src_stride; const float *src_a, *src_c; /* source rows (pre-BN) + their affine or NULL */
    const int *knn_idx; const float *knn_w;                        /* (B*R,3) from sn2_three_nn or NULL          */
    const float *skip; int skip_stride;                            /* (B*R, skip_stride) or NULL                 */
    sn2_block blk;
    float *h; int h_stride;         /* (B*R,h_stride) pre-BN activations; h_stride = cout rounded up to 4        */
    const float *dy;                /* backward in : d loss / d (a*h+c)  (B*R,h_stride)                          */
    float *dsrc; int dsrc_stride;   /* backward out: ACCUMULATED d loss / d (sa*src+sc) (B*S,dsrc_stride) or NULL */
    float *dskip; int dskip_stride; /* backward out: ACCUMULATED (B*R, >=cb) or NULL                             */
    float *du_scratch;              /* backward workspace (B*R, max(ca, h_stride)) when knn_idx and dsrc are given */
    float *scatter_ws;              /* backward workspace when knn_idx and dsrc are given: SN2_INTERP_WS_WORDS(B,R,S)
                                       32-bit words (inverted index of the 3-NN table)                           */
    int scatter_ready;              /* > 0: scatter_ws already holds the index (sn2_interp_index); 0: sn2_fp_backward builds it;
                                       < 0: sn2_fp_backward leaves the per-row input gradients in du_scratch ((B*R, ca) rows)
                                       and does NOT add them onto dsrc: the caller transposes the interpolation itself
                                       (sn2_global_pool_backward does, for the plot's one source)                 */
    const int *bn_sums_done;        /* non-NULL (the `ok` word of sn2_head_bn_sums / sn2_fp_bn_sums): blk.dgamma /
                                       blk.dbeta already hold this BatchNorm's gradients, sn2_fp_backward launches no
                                       pass over the rows for them.  NULL: it does                                  */
    float *src_ws;                  /* workspace SN2_FP_SRC_WS_WORDS(B,R,S,cout) floats or NULL.  Given with knn_idx on a
                                       layer of more than 64*SN2_STAT_SLOTS rows with cb % 4 == 0 (the per-point layer),
                                       everything linear in the interpolation is done once per SOURCE row: forward
                                       gathers rows of T = W_A (sa*src+sc) kept here; backward keeps the partial sums of G[s] = sum of
                                       w * d pre-activation over the rows interpolating s here, one row per 63 list
                                       entries (dsrc += G W_A, dW_A += G^T (sa*src+sc)).  NULL: every row rebuilds its interpolated input.     */
    int act_bf16;                   /* non-zero (only with src_ws, i.e. on the per-point layer, and with bn_sums_done): the rows
                                       of h, dy and du_scratch are bfloat16 (same ELEMENT strides: 72-byte rows at cout = 34)
                                       -- BASELINE.json configs[4]: the three per-point activation buffers are what the step
                                       streams; statistics, every sum and the weight gradients stay fp32               */
    const int *row_perm;            /* NULL, or (B*R) ints, a permutation of 0..R-1 per plot (with src_ws only): the backward
                                       pass keeps the d pre-activation row of target row r of plot b at row b*R + row_perm[b*R + r]
                                       of du_scratch, and the inverted index (sn2_interp_index_perm) lists those positions.
                                       With row_perm = the targets' positions along a space-filling curve (the last B*N words
                                       of sn2_fps's workspace) the rows a source gathers lie close together: the gather of the
                                       source-side backward runs out of L2 instead of fetching every row three times          */
} sn2_fp;
/* chunks of at most 63 list entries an inverted 3-NN index of R rows over S sources can have, per plot (+ one slot per
 * source: every list's last chunk may be short) */
#define SN2_INTERP_CHUNKS(R, S) ((3 * (size_t)(R) + 62) / 63 + (size_t)(S))
#define SN2_FP_SRC_WS_WORDS(B, R, S, cout) ((size_t)(B) * SN2_INTERP_CHUNKS(R, S) * ((((cout) + 3) / 4) * 4))
/* The transpose of knn_interpolate (its backward) is done as a gather through an inverted index of the 3-NN table:
 * source -> list of (target row, normalised weight).  The index depends on positions only, so it can be built ahead of
 * the backward pass (in the geometry pass) with sn2_interp_index; otherwise sn2_fp_backward builds it itself. */
#define SN2_INTERP_WS_WORDS(B, R, S) \
    ((size_t)(B) * (S) * (((R) + 2047) / 2048 + 6) + 6 * (size_t)(B) * (R) + 64 + 4 * (size_t)(B) * SN2_INTERP_CHUNKS(R, S))
/* src_pos: (B*S,4) x,y,z,- rows of the SOURCE positions or NULL.  Given, the index also holds the sources of every plot
 * in Morton order, and the source-side backward (sn2_fp.src_ws) walks them in that order, one stretch per XCD, so that
 * target rows shared by neighbouring sources stay in that XCD's L2. */
int sn2_interp_index(const int *knn_idx, const float *knn_w, const float *src_pos, int B, int R_per_plot,
                     int S_per_plot, float *ws, void *stream);
/* the same index over PERMUTED target rows: list entries name row_perm[b*R + r] instead of r (sn2_fp.row_perm; NULL = identity) */
int sn2_interp_index_perm(const int *knn_idx, const float *knn_w, const float *src_pos, const int *row_perm, int B,
                          int R_per_plot, int S_per_plot, float *ws, void *stream);
/* the same for G consecutive batches of B plots each in ONE set of launches (round 5): knn_idx / knn_w / src_pos / row_perm are the
 * group's arrays (G*B plots, batch-major), batch h's index is written to ws + h * ws_stride_words (an ordinary B-plot workspace:
 * its consumers do not change; ws_stride_words >= SN2_INTERP_WS_WORDS(B,R,S), a multiple of 4).  The geometry pass of a
 * pipelined loop covers several batches per launch: 12 launches per pass instead of 12 per batch. */
int sn2_interp_index_group(const int *knn_idx, const float *knn_w, const float *src_pos, const int *row_perm, int G, int B,
                           int R_per_plot, int S_per_plot, float *ws, size_t ws_stride_words, void *stream);
int sn2_fp_forward(const sn2_fp *p, int training, void *stream);
int sn2_fp_backward(const sn2_fp *p, void *stream);

/* per-plot max over R_per_plot rows of a*h+c -- global_max_pool, model/point_net2.py:39.
 * h has row stride C rounded up to 4.  out (B,C); arg (B,C) row attaining it; backward scatters dout into dy
 * (B*R, same stride), which the caller zeroed. */
int sn2_plot_max_forward(const float *h, const float *a, const float *c, int B, int R_per_plot, int C, float *out,
                         int *arg, void *stream);
int sn2_plot_max_backward(const float *dout, const int *arg, int B, int R_per_plot, int C, float *dy, void *stream);

/* The backward of the global level's pool in one launch (round 5): for FP3's per-row input gradients du (B*R, du_stride >= C) that
 * sn2_fp_backward left in its du_scratch (scatter_ready < 0) --
 *   dx (B,C) += sum over the plot's rows of du   (the transpose of knn_interpolate with k = 1 from the plot's one source, :137/:41),
 *   dy (B*R,C), zero-filled by the caller: dy[b R + arg[b][c]][c] = dx[b][c]   (the backward of global_max_pool, :39),
 *   dgamma, dbeta (C) += the BatchNorm sums over dy of the block whose pre-BatchNorm rows are h (B*R,C) with saved statistics
 *   mean, invstd -- B terms per channel, dy being zero elsewhere; hand the block's sn2_fp_backward a non-NULL bn_sums_done.
 * Replaces the gather inside sn2_fp_backward, sn2_plot_max_backward and the BatchNorm-sum pass of the global SA block.  C = 64. */
int sn2_global_pool_backward(const float *du, int du_stride, const int *arg, const float *h, const float *mean,
                             const float *invstd, int B, int R_per_plot, int C, float *dx, float *dy, float *dgamma,
                             float *dbeta, void *stream);

/* The reference architecture's global level in ONE launch, TRAINING mode (round 4): SA3 = MLP[35,64] on cat[x2, pos2]
 * (model/point_net2.py:133, 37-42), its BatchNorm, the plot's max (:39), FP3 = MLP[96,64] on cat[plot feature, x2] (:137,
 * 62-67) and its BatchNorm -- what sn2_fp_forward(sa3, 1), sn2_plot_max_forward and sn2_fp_forward(fp3, 1) do in five launches,
 * with the same staging, tiles and per-64-row statistics; only the two BatchNorms' sums cross workgroups (one workgroup of 16
 * waves per plot, 8-byte {tag, value} granules, every workgroup finalises the statistics itself in a fixed order).
 *   sa3, fp3: the descriptors the separate calls take (sa3: ca 32, cb 3, no 3-NN table, src = x2 (B*M2,32), skip = pos2 (B*M2,4);
 *             fp3: ca 64, cb 32, S_per_plot 1, src = x3, its 3-NN table, skip = x2; both cout 64, h_stride 64, fp32 operands);
 *             blk.stat_slots is not used;  x3 (B,64), arg3 (B,64): the plot feature and the rows attaining it;
 *   xchg: SN2_GLOBAL_XCHG_WORDS(B) 64-bit words and ctl: SN2_GLOBAL_CTL_WORDS 32-bit words, BOTH ZERO-FILLED ONCE and then left to
 *         the library (the launch epoch lives there: allocate them once, at the size of the largest batch, and never again while
 *         graphs that captured their address exist); launches that share them must be on one stream at a time.
 *   A wait for the peers' sums is bounded (HIP does not promise that the B workgroups of a launch are resident together): a
 *   workgroup whose wait runs out counts itself in ctl[1] and leaves; every workgroup takes a ticket on its way out, and the
 *   one that leaves LAST -- all its peers are gone -- finds the count changed and computes the whole level again, alone, with the
 *   same tiles and the same fixed-order sums: the results (rows, statistics, running statistics, counters) are those of an
 *   undisturbed launch, bit for bit, whatever happened (torch's BatchNorm has no such failure mode: model/point_net2.py:45-53).
 *   An undisturbed launch pays one fence and one atomic per workgroup for it.  ctl[1] stays as a sticky count a host may read
 *   where it synchronises anyway, to learn that launches are being repeated.
 * SN2_ELIMIT for other shapes, bfloat16 operands or more than 28 plots: use the separate calls. */
#define SN2_GLOBAL_XCHG_WORDS(B) ((size_t)2 * (size_t)(B) * 4 * 128)
#define SN2_GLOBAL_CTL_WORDS 8
int sn2_global_level_forward(const sn2_fp *sa3, const sn2_fp *fp3, float *x3, int *arg3, unsigned long long *xchg,
                             unsigned *ctl, void *stream);
/* tests only: the sweeps (~1 us each, default 2^18; 0 = back to it) an exchange wait of sn2_global_level_forward makes before
 * it gives up */
int sn2_debug_global_spin_limit(unsigned sweeps);

/* ---- pointwise head: lin1+ReLU, lin2, softmax/sigmoid/product -- model/point_net2.py:141-151 ----------------
 * f (R,34) pre-BN with affine (fa,fc); coverages (R,4), proba (R,4). */
typedef struct sn2_head {
    int R, cin, f_stride;           /* rows, 34, 36                                                              */
    const float *f, *fa, *fc;
    const float *W1, *b1, *W2, *b2; /* (16,34),(16),(5,16),(5)                                                   */
    float *coverages, *proba;
    const float *dcoverages, *dproba; /* backward in (R,4) each, either may be NULL                              */
    float *dy;                      /* backward out: d loss / d (fa*f+fc) (R,f_stride)                           */
    float *dW1, *db1, *dW2, *db2;   /* ACCUMULATED                                                               */
    int grad_replicas, grad_replica_stride; /* images of the four gradients, as in sn2_block                            */
    const int *drop_mask;           /* F.dropout between lin1 and lin2 (model/point_net2.py:142) in training: (R) words,
                                       bit j set = hidden channel j of the row is KEPT; NULL = no dropout (eval, p = 0) */
    float drop_scale;               /* 1/(1-p) applied to the kept channels (0 when p = 1)                         */
    int act_bf16;                   /* non-zero: the rows of f and dy are bfloat16 (as sn2_fp.act_bf16 of the block that wrote f) */
    float *zero_fill; long zero_fill_words; /* sn2_head_forward only, or NULL / 0: a buffer (16-byte aligned, a multiple of 4 words)
                                       that the forward kernel also clears -- the backward pass's zero-filled arena (sn2_net_bwd),
                                       so that no launch of its own has to clear it in front of the backward pass */
} sn2_head;
int sn2_head_forward(const sn2_head *p, void *stream);
/* EVAL only: the per-point layer (p: the 34 + 8 -> 34 block with its 3-NN table, source-side workspace p->src_ws required) and
 * the head (hd: f / f_stride unused -- the rows never leave the CU; fa, fc = p->blk.a, p->blk.c; no dropout, fp32 rows) in one
 * pass: fp_forward(p, eval) + head_forward(hd) without the (B*N, 36) activation buffer between them (model/point_net2.py:139-151
 * under model.eval(), predict.py:96-126).  Same operations in the same order as the two calls: the same bits. */
int sn2_fp_head_eval(const sn2_fp *p, const sn2_head *hd, void *stream);
int sn2_head_backward(const sn2_head *p, void *stream);
/* After sn2_head_backward: the gradients of the BatchNorm whose output the head reads (FP1's), obtained from lin1's weight
 * and bias gradients instead of a pass over all rows (derivation in fp.hip).  gamma, beta, mean, invstd: that BatchNorm's
 * parameters and saved batch statistics; dgamma, dbeta: ACCUMULATED, complete on return.  *ok (device int) = 1 when the
 * identity was used, 0 when some |gamma| <= 1e-4 made it unusable and the kernel summed over the rows itself (p->f = the
 * BatchNorm's input rows, p->dy = the gradient of its output).  Hand `ok` to the producer's sn2_fp_backward as bn_sums_done. */
int sn2_head_bn_sums(const sn2_head *p, const float *gamma, const float *beta, const float *mean, const float *invstd,
                     float *dgamma, float *dbeta, int *ok, void *stream);
/* The same for the BatchNorm whose output an FP block interpolates (its columns 0..ca-1), after that block's
 * sn2_fp_backward: the interpolation weights of a row sum to 1, so the identity carries over (p->src = the BatchNorm's
 * input rows, p->dsrc = the gradient of its output). */
int sn2_fp_bn_sums(const sn2_fp *p, const float *gamma, const float *beta, const float *mean, const float *invstd,
                   float *dgamma, float *dbeta, int *ok, void *stream);

/* ---- 2D projections -- model/project_to_2d.py -------------------------------------------------------------- */

/* P2 project_to_plotwise_coverages (:7-55): bbox-normalised diam_pix^2 grid per plot, per-pixel max of channels
 * 0,2,3 (first point wins ties), bare soil = 1 - low-veg max, mean over occupied pixels.
 * cloud_xy: plot b's normalised x row at cloud_xy + b*plot_stride, y row at + b*plot_stride + N.
 * keys (B*D*D*3) u64 workspace; pix (B*N) int32 out (x_pix*D + y_pix, bit-exact); arg (B*D*D*3) int32 out;
 * nocc (B) int32 out; pred (B,4) out. */
int sn2_plot_project_forward(const float *pred_pointwise, const float *cloud_xy, long plot_stride, int B, int N,
                             int D, unsigned long long *keys, int *pix, int *arg, int *nocc, float *pred,
                             void *stream);
/* The same in two parts, for callers that run their position-only kernels ahead of the feature kernels (the pixel id of a
 * point is a function of the plot's x, y only, project_to_2d.py:16-22): sn2_plot_pixels computes mm (B,4) = the plots'
 * (xmin, xmax, ymin, ymax) and pix (B*N) = the ids sn2_plot_project_forward would write; sn2_plot_project_forward_pix is the
 * rest of it from those ids -- keys: u64 workspace of SN2_P2_KEY_PARTS(N)*B*D*D*3 words (no initialisation: every slice of a
 * plot writes its own table, the finalisation takes their maximum), same arg / nocc / pred. */
#define SN2_P2_KEY_PARTS(N) (((N) + 4095) / 4096 < 1 ? 1 : (((N) + 4095) / 4096 > 64 ? 64 : ((N) + 4095) / 4096))
int sn2_plot_pixels(const float *cloud_xy, long plot_stride, int B, int N, int D, float *mm, int *pix, void *stream);
int sn2_plot_project_forward_pix(const float *pred_pointwise, const int *pix, int B, int N, int D,
                                 unsigned long long *keys, int *arg, int *nocc, float *pred, void *stream);
/* d pred (B,4) -> d pred_pointwise (B*N,4): every row written (a point gets its pixel's gradient iff it is the pixel's
 * arg-max; pix, arg, nocc from the forward). */
int sn2_plot_project_backward(const float *dpred, const int *arg, const int *nocc, const int *pix, int B, int N,
                              int D, float *dpointwise, void *stream);

/* P1 project_to_2d_rasters (:58-113): fixed grid, clipped; rasters (B,3,D,D) [low,med,high], image[y,x], NaN where
 * empty, rows flipped; pix (B*N) int32 out = y_pix*D + x_pix (unflipped).  coverages (B*N,4) row-major. */
int sn2_raster_project(const float *coverages, const float *cloud_xy, long plot_stride, int B, int N, int D,
                       int diam_meters, unsigned long long *keys, int *pix, float *rasters, void *stream);

/* Parcel mosaic merge -- the plot-after-plot weighted merge of overlapping plot rasters that predict.py:136-141 obtains
 * from rasterio.merge with the callback inference/geotiff_raster.py:294-347 (_weighted_average_of_rasters), over per-plot
 * rasters carrying the radial weight band of add_weights_band_to_rasters (:103-118).  The callback's rule is ORDER
 * DEPENDENT (its weight band also sums the weights of plots whose score is no-data in a pixel), so the kernel keeps the
 * order: one thread per (band, parcel pixel) of the window folds plots 0..B-1 in sequence.
 * rasters (B,3,D,D) as produced by sn2_raster_project (NaN = no data); weights (D,D) (NaN outside the disc);
 * offsets (B,2) int32 = (row, col) of each plot's top-left pixel in the parcel grid;
 * mean, wsum (3,H,W): the running mosaic's score bands and weight bands, NaN = no data (caller fills with NaN once);
 * only parcel pixels in rows [win_y0, win_y0+win_h) x cols [win_x0, win_x0+win_w) are visited (pass the bounding window
 * of the batch, or 0,0,H,W). */
int sn2_mosaic_merge(const float *rasters, const float *weights, const int *offsets, int B, int D, int H, int W,
                     float *mean, float *wsum, int win_y0, int win_x0, int win_h, int win_w, void *stream);

/* Mosaic finalisation -- finalize_merged_raster (inference/geotiff_raster.py:262-285) up to, not including, the GIS
 * admissibility band: keep the three score bands + one weight band, insert the hard medium-vegetation band
 * (insert_hard_med_veg_raster_band :119-144: the threshold among linspace(0,1,10001) whose hard coverage is closest to the
 * mean soft coverage, first minimum), then NaN -> 0 wherever at least one score exists and NaN everywhere else.
 * mean (3,H,W) and wsum (H,W) from sn2_mosaic_merge; hist_ws SN2_MOSAIC_HIST_WORDS ints, sum_ws one double (both scratch);
 * thr_out[0] = threshold, thr_out[1] = its index; out (5,H,W) = [Vb, Vm_soft, Vh, Vm_hard, weights]. */
#define SN2_MOSAIC_HIST_WORDS 10004
int sn2_mosaic_finalize(const float *mean, const float *wsum, int H, int W, int *hist_ws, double *sum_ws, float *thr_out,
                        float *out, void *stream);

/* ---- loss block of the timed training step: learning/loss_functions.py:9-57 combined as learning/train.py:58-62,
 *   total = get_absolute_loss(pred, gt) + m * get_NLL_loss(proba, pdf_all) + e * get_entropy_loss(proba)
 * pred (B,4) fp32 plot-wise coverages, gt (B,4) fp64, proba (R,4) fp32 pointwise class probabilities, pdf (R,3) fp64 the
 * KDE-mixture densities at the points' heights (the reference evaluates them on the CPU each step, :30-42; KDE fitting is
 * out of scope, so they are an input).  partials: 2*SN2_LOSS_BLOCKS fp64 workspace.  out[4] = total, absolute, NLL,
 * entropy.  Backward: grad_total = device scalar d(objective)/d(total); writes dpred (B,4), dproba (R,4).
 * A term that is switched off is SKIPPED (not multiplied by zero) and its inputs may be absent -- what the reference's loop,
 * which calls the three functions one by one (learning/train.py:58-60), needs: B = 0: no absolute term (pred, gt, dpred
 * unused); m == 0: no NLL (pdf unused: rows that are no probability vectors give no 0 * log(<= 0) = NaN); e == 0: no entropy;
 * R = 0: no pointwise term (proba, pdf, partials, dproba unused).  The skipped components of out[] are 0. */
/* KDE-mixture densities at the points' heights -- KdeMixture.predict, learning/kde_mixture.py:65-70 (three scipy
 * interp1d(kind="linear") over one knot vector), which get_NLL_loss evaluates on the CPU for all B*N points every step
 * (learning/loss_functions.py:30-42).  cloud (B,C,N) fp32, height = cloud[:, z_channel, :] * z_max formed in fp32;
 * X (K) ascending knots and Y (3,K) the three tables, fp64; pdf (B*N,3) fp64 = what sn2_loss_* take.  Heights outside
 * [X[0], X[K-1]] (scipy raises there) give NaN.  Fitting the KDEs (KDEpy) is out of scope: the tables are an input. */
int sn2_kde_lookup(const float *cloud, int B, int C, int N, int z_channel, float z_max, const double *X, const double *Y,
                   int K, double *pdf, void *stream);
#define SN2_LOSS_BLOCKS 1024
int sn2_loss_forward(const float *pred, const double *gt, int B, const float *proba, const double *pdf, int R, double m,
                     double e, double *partials, double *out, void *stream);
int sn2_loss_backward(const float *pred, const double *gt, int B, const float *proba, const double *pdf, int R, double m,
                      double e, const double *grad_total, float *dpred, float *dproba, void *stream);

/* Projection + loss of the training step (learning/train.py:54-62) in three launches instead of seven (round 5):
 * sn2_projected_loss_forward = sn2_plot_project_forward_pix (pixel ids `pix` from sn2_plot_pixels) + sn2_loss_forward in two
 * launches -- the scatter of the coverages and the pointwise loss sums side by side, then the per-plot finalisation whose last
 * workgroup adds the loss up -- and sn2_projected_loss_backward = sn2_loss_backward + sn2_plot_project_backward in one pass over
 * the points.  Same pred / arg / nocc / dcoverages / dproba bits as the separate calls; out[4] as sn2_loss_forward's to fp64
 * re-association.  keys: SN2_P2_KEY_PARTS(N)*B*D*D*3 u64; partials: SN2_PROJECTED_LOSS_WS doubles (no initialisation). */
#define SN2_PROJECTED_LOSS_WS (2 * 512 + 2)
int sn2_projected_loss_forward(const float *coverages, const int *pix, const float *proba, const double *pdf, const double *gt,
                               int B, int N, int D, double m, double e, unsigned long long *keys, int *arg, int *nocc,
                               float *pred, double *partials, double *out, void *stream);
int sn2_projected_loss_backward(const float *pred, const double *gt, int B, const float *proba, const double *pdf, int N, int D,
                                double m, double e, const double *grad_total, const int *arg, const int *nocc, const int *pix,
                                float *dcoverages, float *dproba, void *stream);

/* ---- optimiser step of the timed training step -- torch.optim.Adam as configured in learning/train.py:180-185
 * (L2 weight decay added to the gradient), on flat buffers; grad_scale multiplies the gradient first (1/world). */
int sn2_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int *step_dev /* two device ints: {steps taken so far (incremented here), 0} */,
                  float grad_scale, void *stream);

/* ==== the whole network behind ONE call per pass (round 5) ===================================================
 * PointNet2.forward of the reference (model/point_net2.py:106-153) is ~25 of the entry points above in a fixed order, and
 * loss.backward() through it (learning/train.py:64) ~10 more.  A host that issues them one by one from Python spends more
 * time between the launches than the device needs for the kernels (the reference's training loop as written: 3.5 ms per step
 * over ~2.4 ms of kernels).  sn2_net_geometry / sn2_net_forward / sn2_net_backward issue the same entry points, with the same
 * descriptors, in the same order, from inside the library: one call each.  Nothing else changes: every buffer is the
 * caller's, nothing is allocated, freed or synchronised, all launches go to `stream` (and, for the forked geometry pass, to
 * the caller's side streams, ordered by the events of a caller-owned context).  Results are those of the separate calls, bit
 * for bit (tests/test_gpu_executor.py).
 *
 * The reference architecture only (model/point_net2.py:81-99): SA1 MLP[11,16,16], SA2 MLP[19,32], global SA MLP[35,64],
 * FP3 MLP[96,64] k=1, FP2 MLP[80,34] k=3, FP1 MLP[42,34] k=3, lin1 34->16, lin2 16->5. */

/* one (Linear -> ReLU -> BatchNorm1d) block of the model: its parameters and buffers (device), and where its four gradients
 * live inside the flat parameter gradient (offsets in floats; parameters() order of the reference module) */
typedef struct sn2_net_layer {
    int cin, cout;
    const float *W, *b, *gamma, *beta;
    float *running_mean, *running_var;
    long long *num_batches_tracked;      /* or NULL */
    int gW, gb, ggamma, gbeta;
    int mma_bf16;                        /* sn2_block.mma_bf16 of this block */
} sn2_net_layer;

typedef struct sn2_net_model {
    sn2_net_layer sa1[2], sa2, sa3, fp3, fp2, fp1;      /* model/point_net2.py:84-93 */
    const float *lin1_W, *lin1_b, *lin2_W, *lin2_b;     /* :95-99 */
    int g_lin1_W, g_lin1_b, g_lin2_W, g_lin2_b;
    int n_flat;                          /* length of the flat parameter (and gradient) vector: 14 997 */
    float r1_sq, r2_sq;                  /* fp32(r*r) of the two ball queries, r*r evaluated in double (sn2_ball_query) */
    int max_neighbors;                   /* model/point_net2.py:24 (2000) */
    float drop_p;                        /* args.drop (:142) */
    int fuse_global_level;               /* training: sn2_global_level_forward where its limits allow (else the separate calls) */
    int fuse_eval_head;                  /* eval: sn2_fp_head_eval instead of sn2_fp_forward + sn2_head_forward */
    int source_side;                     /* hand out sn2_fp.src_ws (the per-point layer's source-side form) */
    int fps_waves_shared, fps_waves_many;/* sn2_fps_waves of the level-1 FPS of a pass that shares the chip (<= 32 plots / more) */
} sn2_net_model;

typedef struct sn2_net_dims {
    int B, N, M1, M2;                    /* plots, points per plot, centroids of the two levels (ceil(fp32(n) * fp32(ratio))) */
    int cap1, cap2;                      /* row stride of nbr1 / nbr2: min(max_neighbors, N) / min(max_neighbors, M1) */
    int act_bf16;                        /* the per-point activation buffers h1 / dy1 / du1 are bfloat16 (sn2_fp.act_bf16) */
    int p2_diam_pix;                     /* > 0: the geometry pass also computes the pixel ids of sn2_plot_pixels (geo.p2_*) */
} sn2_net_dims;

/* The position-only tables of one batch -- what SAModule / FPModule obtain from torch_cluster in the reference
 * (model/point_net2.py:22-25, 63) -- and the two input-only pieces a geometry pass may also produce.  Shapes as in the
 * entry points that fill them.  sn2_net_geo_carve lays them out in one caller-owned arena. */
typedef struct sn2_net_geo {
    const float *xyz;                                    /* (B,3,N): the batch's positions */
    int *idx1; float *pos1_soa, *pos1_aos; int *ws1;     /* sn2_fps level 1; ws1 = NULL where the FPS fills no workspace */
    int *nbr1, *cnt1; unsigned long long *tot1; int *ord1;
    int *idx2; float *pos2_soa, *pos2_aos; int *ws2;
    int *nbr2, *cnt2; unsigned long long *tot2; int *ord2;
    const float *pos3;                                   /* (B,3,1) zeros: the global feature sits at the origin (:41) */
    int *knn3_idx; float *knn3_w;                        /* (B*M2,3) */
    int *knn2_idx; float *knn2_w;                        /* (B*M1,3) */
    int *knn1_idx; float *knn1_w;                        /* (B*N,3)  */
    float *inv3, *inv2, *inv1;                           /* SN2_INTERP_WS_WORDS each */
    int *nn_ws2, *nn_ws1;                                /* SN2_THREE_NN_XY_WS_WORDS or NULL (full scan) */
    const int *rank1;                                    /* sn2_fp.row_perm of FP1 or NULL */
    float *rows0;                                        /* (B*N,12) sn2_pack_rows */
    int *p2_pix; float *p2_mm;                           /* sn2_plot_pixels or NULL */
} sn2_net_geo;

/* The buffers of one forward pass (kept for its backward pass when training). */
typedef struct sn2_net_act {
    float *aux;                          /* 4 * 260 floats: (a, c, mean, invstd) of the seven blocks, in model order */
    float *stats;                        /* SN2_STAT_SLOTS * 2 * 260 floats: their statistic slots */
    float *ext1; int *arg1; float *x1;   /* (B*M1,16) */
    float *ext2; int *arg2; float *x2;   /* (B*M2,32) */
    float *h_sa3, *h3;                   /* (B*M2,64) */
    float *x3; int *arg3;                /* (B,64) */
    float *h2;                           /* (B*M1,36) */
    void *h1;                            /* (B*N,36) fp32 / bfloat16 rows; unused by the fused eval pass */
    float *src_ws1, *src_ws2;            /* SN2_FP_SRC_WS_WORDS of FP1 / FP2, or NULL (sn2_fp.src_ws) */
    float *cov, *proba;                  /* OUT (B*N,4): coverages_pointwise, proba_pointwise -- set by the caller, not carved */
    const int *drop_mask;                /* sn2_head.drop_mask or NULL -- set by the caller */
    float *bwd_arena; long bwd_arena_words; /* set by the caller or NULL / 0 (training forward): the arena of the backward pass that will
                                            follow (sn2_net_bwd.arena, arena_words): the forward's last kernel clears it, and that
                                            backward pass is told so (sn2_net_bwd.arena_is_zero) */
} sn2_net_act;

/* The buffers of one backward pass: `arena` (zero-filled INSIDE sn2_net_backward) = 32 images of the flat parameter gradient,
 * image stride = n_flat rounded up to 64 floats, followed by the accumulate-into buffers; the rest is scratch. */
typedef struct sn2_net_bwd {
    const float *dcov, *dproba;          /* IN (B*N,4) each, either may be NULL -- set by the caller */
    float *arena; long arena_words;
    int images, image_stride;
    float *dy2, *dy3, *dx1, *dx2, *dx3, *dy_sa3;
    void *dy1, *du1;                     /* (B*N,36) rows of the activation type */
    float *du2, *du3;                    /* (B*M1,64), (B*M2,64) */
    int *bn_ok;                          /* 4 words */
    float *src_ws1, *src_ws2;
    int defer_grad_reduce;               /* set by the caller: leave the images unfolded (sn2_adam_step_images folds them) */
    int arena_is_zero;                   /* set by the caller: the forward pass cleared `arena` (sn2_net_act.bwd_arena) and nothing has
                                            touched it since: sn2_net_backward does not clear it again */
    int frozen_stats;                    /* set by the caller: the forward pass ran with io.training = SN2_BN_FROZEN_KEEP (every
                                            block's sn2_block.frozen_stats) */
} sn2_net_bwd;

#define SN2_NET_FORK 1          /* geometry: level-2 chain on io.stream_b, per-point 3-NN chain on io.stream_c (needs io.ctx) */
#define SN2_NET_SHARED 2        /* geometry: the pass shares the chip with other kernels (level-1 FPS: model.fps_waves_*) */
#define SN2_NET_INVERTED 4      /* geometry: also the inverted 3-NN tables (only a backward pass reads them) */
#define SN2_NET_DEFER_JOIN 8    /* geometry (with FORK): return without joining; the forward pass that follows joins */
#define SN2_NET_INPUT_ONLY 16   /* geometry: also rows0 (and the P2 pixel ids when dims.p2_diam_pix > 0) from io.cloud */
#define SN2_NET_HAS_ROWS0 32    /* forward: geo.rows0 is already packed */
#define SN2_NET_JOIN_PENDING 64 /* forward: the geometry pass on this ctx was launched with DEFER_JOIN: SA2 waits for chain b, FP1 for chain c */
#define SN2_NET_WITH_GEOMETRY 128 /* forward: run the geometry pass first (forked when io.ctx and the side streams are given), the
                                     row packing beside the level-1 FPS */
#define SN2_NET_HAS_INVERTED 256 /* forward (training): geo.inv* are already built (else they are built here) */

typedef struct sn2_net_io {
    const float *cloud;                  /* (B,10,N) features on the device, or NULL where not needed */
    const int *fps_start;                /* (2,B) start indices of the two FPS calls, or NULL (= 0) */
    unsigned *fps_status;                /* sn2_fps_status's word or NULL */
    unsigned long long *gl_xchg; unsigned *gl_ctl;   /* sn2_global_level_forward's exchange area, or NULL: separate launches */
    void *stream_b, *stream_c, *stream_pack;         /* side streams of a forked pass (hipStream_t), or NULL */
    void *ctx;                           /* sn2_net_ctx_create: the events that order them, or NULL (no fork) */
    int flags;                           /* SN2_NET_* */
    int training;                        /* model.training (0 / 1), or SN2_BN_FROZEN_KEEP: eval mode with a backward to come */
} sn2_net_io;

/* events of a forked geometry pass: created once by the caller (one per model and device), used by one pass at a time */
int sn2_net_ctx_create(void **ctx);
int sn2_net_ctx_destroy(void *ctx);
/* Lay the tables / buffers out in caller-owned arenas: every pointer of *out = base + its offset (256-byte aligned); *bytes = the
 * arena's size.  base = NULL: the pointers ARE the offsets (a host can cache them per shape).  Fields marked "set by the
 * caller" are left NULL. */
int sn2_net_geo_carve(const sn2_net_model *m, const sn2_net_dims *d, void *base, sn2_net_geo *out, size_t *bytes);
int sn2_net_act_carve(const sn2_net_model *m, const sn2_net_dims *d, int training, void *base, sn2_net_act *out, size_t *bytes);
int sn2_net_bwd_carve(const sn2_net_model *m, const sn2_net_dims *d, void *arena_base, void *scratch_base, sn2_net_bwd *out,
                      size_t *arena_bytes, size_t *scratch_bytes);
/* the position-only kernels of one batch: FPS x2, ball query x2, SA work items x2, 3-NN x3 [, inverted tables x3, rows0, P2 ids]
 * -- PointNet2._geometry; replaces the torch_cluster calls of model/point_net2.py:22-25, 63 */
int sn2_net_geometry(const sn2_net_model *m, const sn2_net_dims *d, const sn2_net_geo *g, const sn2_net_io *io, void *stream);
/* PointNet2.forward, model/point_net2.py:131-151 (training or eval by io->training) */
int sn2_net_forward(const sn2_net_model *m, const sn2_net_dims *d, const sn2_net_geo *g, const sn2_net_act *a,
                    const sn2_net_io *io, void *stream);
/* its backward pass (loss.backward(), learning/train.py:64): every parameter gradient into image 0 of b->arena */
int sn2_net_backward(const sn2_net_model *m, const sn2_net_dims *d, const sn2_net_geo *g, const sn2_net_act *a,
                     const sn2_net_bwd *b, void *stream);

/* sn2_grad_reduce + sn2_adam_step in one launch, for a step with no exchange between them: grad_images = the `replicas` images
 * of the flat gradient (image stride `stride` floats); image 0 holds the folded gradient afterwards (same additions, same order
 * as sn2_grad_reduce).  sn2_net_bwd.defer_grad_reduce makes sn2_net_backward leave the images unfolded for it. */
int sn2_adam_step_images(float *param, float *grad_images, int replicas, int stride, float *exp_avg, float *exp_avg_sq, int n,
                         float lr, float beta1, float beta2, float eps, float weight_decay, int *step_dev, float grad_scale,
                         void *stream);

#ifdef __cplusplus
}
#endif
#endif /* STRATA_HIP_H */
