"""Per-queue timeline of scripts/bench_inference.py from a rocprofv3 kernel trace (gpurun_out/.../*_kernel_trace.csv):
busy time, idle gaps and the largest kernels of every hardware queue over the LAST parcel of the run.
usage: python scripts/timeline_inference.py kernel_trace.csv"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["name"] = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
rows.sort(key=lambda r: r["s"])
# the last parcel: from the last-but-n mosaic_merge ... take the last 8 mosaic merges (8 launches per parcel at B = 256)
merges = [r for r in rows if "mosaic_merge" in r["name"]]
n_launch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
# the timed runs come before the per-entry timing run; take the parcel before the last one
last = merges[-2 * n_launch:-n_launch]
t1 = last[-1]["e"]
prev = merges[-2 * n_launch - 1]["e"] if len(merges) > 2 * n_launch else rows[0]["s"]
sel = [r for r in rows if r["s"] >= prev and r["e"] <= t1 + 1000]
t0 = sel[0]["s"]
print(f"parcel window {1e-6 * (t1 - t0):.2f} ms, {len(sel)} kernels")
byq = collections.defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append(r)
for q, ks in sorted(byq.items(), key=lambda kv: -sum(k["e"] - k["s"] for k in kv[1])):
    busy = sum(k["e"] - k["s"] for k in ks)
    names = collections.Counter()
    for k in ks:
        names[k["name"][:40]] += k["e"] - k["s"]
    top = ", ".join(f"{n} {1e-6 * v:.2f}" for n, v in names.most_common(6))
    print(f"queue {q}: {len(ks)} kernels, busy {1e-6 * busy:.2f} ms ({100 * busy / (t1 - t0):.0f} %), first {1e-6 * (ks[0]['s'] - t0):.2f} last {1e-6 * (ks[-1]['e'] - t0):.2f} | {top}")
