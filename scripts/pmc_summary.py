"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as the MI355X guide prescribes) into
per-kernel HBM traffic.  Units: the counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide
coalesced streaming reads (MI355X_MICROARCH.md, HBM section), so the corrected read figure doubles it -- an upper
bound for kernels whose reads are narrow or L2-served.

    python scripts/pmc_summary.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv out.json
"""
import collections
import csv
import json
import re
import sys


def load(fn):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = load(sys.argv[1])
write, nw = load(sys.argv[2])
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out[k] = {"launches_sampled": nf.get(k, nw.get(k, 0)), "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
              "hbm_bytes_per_launch_raw": int((f + w) * 1024), "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch_corrected"])[:25]:
    print(f"{k[:60]:60s} fetch {v['FETCH_SIZE_KiB']:10.1f} KiB  write {v['WRITE_SIZE_KiB']:10.1f} KiB  corrected {v['hbm_bytes_per_launch_corrected'] / 1e6:8.2f} MB")
