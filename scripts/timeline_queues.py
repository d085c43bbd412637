"""Per-hardware-queue busy time of a rocprofv3 kernel trace over its last WINDOW ms (default 20): which queue is the busy one,
what runs on it, how long it idles.  usage: python scripts/timeline_queues.py kernel_trace.csv [window_ms] [skip_tail_ms]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["name"] = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
t1 = max(r["e"] for r in rows) - int(skip * 1e6)
t0 = t1 - int(win * 1e6)
sel = [r for r in rows if r["s"] >= t0 and r["e"] <= t1]
print(f"window {win} ms ending {skip} ms before the last kernel: {len(sel)} kernels")
byq = collections.defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append(r)
for q, ks in sorted(byq.items(), key=lambda kv: -sum(k["e"] - k["s"] for k in kv[1])):
    ks.sort(key=lambda k: k["s"])
    busy = sum(k["e"] - k["s"] for k in ks)
    gaps = [b["s"] - a["e"] for a, b in zip(ks, ks[1:])]
    names = collections.Counter()
    for k in ks:
        names[k["name"][:36]] += k["e"] - k["s"]
    top = ", ".join(f"{n} {1e-6 * v:.2f}" for n, v in names.most_common(7))
    print(f"queue {q}: {len(ks)} kernels, busy {1e-6 * busy:.2f} ms ({100 * busy / (t1 - t0):.0f} %), median gap "
          f"{1e-3 * sorted(gaps)[len(gaps) // 2] if gaps else 0:.1f} us, gaps > 20 us: {sum(1 for g in gaps if g > 20000)} | {top}")
