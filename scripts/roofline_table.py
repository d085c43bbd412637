#!/usr/bin/env python3
"""Per-kernel evidence table (north_star: "rocprof HBM GB/s (ball_query, grouping, scatter-max) and MFMA utilisation (MLP
GEMMs) against the chip's peak") from the rocprofv3 outputs of scripts/refresh_profiles.sh:

    python scripts/roofline_table.py <bench.json> <kernel_stats.csv> <pmc_traffic.json> <sq_counter_collection.csv> <out prefix>

Columns per device kernel (one launch, C2 ref-arch sizes; E1, E2 = the measured message counts of the bench line):
  us            average duration, rocprofv3 --kernel-trace --stats of the UNPIPELINED eager run (one stream: in the pipelined run
                the position-only kernels overlap feature kernels and their durations include the contention)
  compulsory    HBM bytes that must move: index entries + first touch of every row read + results written
  L2 gather     bytes gathered again out of the (L2 / Infinity-Cache resident) row tables: NOT compulsory HBM traffic
  PMC           FETCH_SIZE x 2 + WRITE_SIZE of separate --pmc passes (gfx950 half-count of wide reads corrected: guide)
  GB/s, %HBM    compulsory / us against 8 TB/s
  MFMA flops    SQ_INSTS_VALU_MFMA_MOPS_{F32,BF16} x 512; model = the layer shapes (2 Cin Cout per row incl. padding not counted)
  TF/s, %peak   counter flops / us against 157.3 TF/s (fp32 MFMA) -- or 2500 (bf16)
  MFMA busy     SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 256 CUs x 4 SIMDs): share of SIMD-cycles with the matrix pipe busy
"""
import collections
import csv
import json
import re
import sys

bench, stats_csv, traffic_json, sq_csv, out = sys.argv[1:6]
b = json.loads([l for l in open(bench) if l.startswith("{")][-1])
cfg = b["config"]
B, N, E1, E2 = cfg["plots_per_gpu"], cfg["points_per_plot"], cfg["messages_sa1"], cfg["messages_sa2"]
M1, M2 = 1024, 256
R = B * N


def norm(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name).split("(")[0]


dur = {}
for r in csv.DictReader(open(stats_csv)):
    dur[norm(r["Name"])] = (float(r["AverageNs"]) * 1e-3, int(r["Calls"]))
traffic = json.load(open(traffic_json))
sq = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sq_csv)):
    sq[norm(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
sqm = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in sq.items()}

# kernel -> (what it is, compulsory bytes, L2-gather bytes, model flops on the matrix cores)
rows0, cpos = 48 * R, 16 * M1 * B
K = collections.OrderedDict()
K["ball_query_grid_kernel"] = ("radius ball query, level 1", 16 * R + 12 * M1 * B + 4 * E1 + 4 * M1 * B, None, 0)
K["sa_mfma_fwd_kernel<8, 2, 16, 16, 0, false, true>"] = ("SA1 statistics pass (gather + layer 1)", 4 * E1 + rows0 + cpos, 48 * E1, 2 * 11 * 16 * E1)
K["sa_mfma_fwd_kernel<8, 2, 16, 16, 1, false, true>"] = ("SA1 forward (gather + MLP + BN + max)", 4 * E1 + rows0 + cpos + 128 * M1 * B, 48 * E1, 864 * E1)
K["sa_mfma_bwd_kernel<8, 2, 16, 16, 2, false>"] = ("SA1 backward, layer 2", 4 * E1 + rows0 + cpos + 128 * M1 * B, 48 * E1, (864 + 512 + 512) * E1)
K["sa_mfma_bwd_kernel<8, 2, 16, 16, 3, false>"] = ("SA1 backward, layer 1", 4 * E1 + rows0 + cpos + 128 * M1 * B, 48 * E1, (864 + 512 + 352) * E1)
K["sa_mfma_fwd_kernel<16, 1, 32, 32, 1, false, true>"] = ("SA2 forward", 4 * E2 + 80 * M1 * B + 16 * M2 * B + 256 * M2 * B, 80 * E2, 1216 * E2)
K["sa_mfma_bwd_kernel<16, 1, 32, 32, 3, false>"] = ("SA2 backward", 4 * E2 + 80 * M1 * B + 16 * M2 * B + 256 * M2 * B, (80 + 64) * E2, (1216 + 1216 + 1024) * E2)
K["scatter_max_kernel<0>"] = ("plot-wise projection: scatter-max", (8 + 16 + 4) * R + 24 * 400 * B, None, 0)
K["fp_fwd_rows2_kernel<34, 8, 34, false>"] = ("FP1 forward, row pass (source-side form; round 5: pipelined input stream)", (24 + 32 + 144) * R, 3 * 144 * R, 0)
K["fp_bwd_rows_kernel<34, 8, 34, 512, false>"] = ("FP1 backward, row pass", (288 + 32 + 136) * R, None, 0)
K["fp_bwd_src_chunk_kernel<34, 8, 34, false>"] = ("FP1 backward, source pass (rows gathered through the chunked inverted index)", (24 + 136) * R, 2 * 136 * R, 0)
K["fp_bwd_src_merge_dw_kernel<34, 8, 34>"] = ("FP1 backward, partial rows -> G, dsrc, dW_A", (144 + 144 + 136 + 2 * 136) * M1 * B, None, 0)
K["head_fwd_mfma_kernel<false>"] = ("head forward (lin1, lin2 on the matrix cores)", (144 + 32) * R, None, 2 * (16 * 35 + 5 * 17) * R)
K["head_bwd_mfma_kernel<false>"] = ("head backward (six contractions on the matrix cores)", (144 + 32 + 144) * R, None,
                                     2 * (34 * 16 + 16 * 5 + 5 * 16 + 16 * 34 + 16 * 35 + 5 * 17) * R)
K["global_level_fwd_kernel"] = ("global level forward: SA3, BatchNorm, plot max, FP3, BatchNorm in one launch",
                                (128 + 16 + 256 + 256) * M2 * B, None, 2 * (35 * 64 + 33 * 64) * M2 * B)
K["three_nn_grid_kernel"] = ("3-NN of the N points among the level-1 centroids", 16 * R + 16 * M1 * B + 24 * R, None, 0)
K["pack_rows_kernel"] = ("row packing", (44 + 48) * R, None, 0)

lines, js = [], {}
hdr = ("| kernel | what | us | compulsory MB | L2 gather MB | PMC MB | PMC / compulsory | GB/s | % of 8 TB/s | MFMA GFLOP (counter) | "
       "model GFLOP | TFLOP/s | % of MFMA peak | MFMA busy % |")
lines += [hdr, "|" + "---|" * 14]
for k, (what, comp, gather, flops_model) in K.items():
    key = next((n for n in dur if n.startswith(k.split("<")[0]) and (k in n or "<" not in k)), None)
    if key is None:
        continue
    us = dur[key][0]
    t = traffic.get(key) or traffic.get(norm(key)) or {}
    pmc = t.get("hbm_bytes_per_launch_corrected")
    s = sqm.get(key, {})
    f32, bf16 = s.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512, s.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512
    flops = f32 + bf16
    peak = 2500.0 if bf16 > f32 else 157.3
    gui = s.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    busy = 100.0 * s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 256 * 4) if gui > 0 else None
    gbs = comp / (us * 1e-6) / 1e9
    tf = flops / (us * 1e-6) / 1e12
    js[k] = {"what": what, "us": round(us, 2), "compulsory_bytes": comp, "l2_gather_bytes": gather, "pmc_bytes": pmc,
             "pmc_over_compulsory": None if not pmc else round(pmc / comp, 2), "GBps": round(gbs, 1), "hbm_frac": round(gbs / 8000, 4),
             "mfma_flops_counter": flops, "mfma_flops_model": flops_model, "TFLOPs": round(tf, 2),
             "mfma_peak_frac": round(tf / peak, 4), "mfma_busy_pct": None if busy is None else round(busy, 2), "sq": s}
    lines.append(f"| `{k}` | {what} | {us:.1f} | {comp / 1e6:.1f} | {'-' if gather is None else f'{gather / 1e6:.1f}'} | "
                 f"{'-' if not pmc else f'{pmc / 1e6:.1f}'} | {'-' if not pmc else f'{pmc / comp:.2f}'} | {gbs:.0f} | {100 * gbs / 8000:.1f} | "
                 f"{flops / 1e9:.3f} | {flops_model / 1e9:.3f} | {tf:.2f} | {100 * tf / peak:.2f} | {'-' if busy is None else f'{busy:.1f}'} |")
open(out + ".md", "w").write(f"# Per-kernel roofline evidence ({cfg['workload']}; E1 = {E1}, E2 = {E2})\n\n" + "\n".join(lines) + "\n")
json.dump(js, open(out + ".json", "w"), indent=1)
print("\n".join(lines))
