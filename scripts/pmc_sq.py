"""Per-kernel means of a rocprofv3 --pmc SQ_* pass.  usage: pmc_sq.py <counter_collection.csv> [name filter ...]"""
import collections
import csv
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name).split("(")[0]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2:]
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    if flt and not any(f in k for f in flt):
        continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k[:58]:58s} waves {m.get('SQ_WAVES', 0):8.0f} wave_cyc {wc:12.0f}  wait_any {m.get('SQ_WAIT_ANY', 0) / wc:5.2f} "
          f"wait_inst {m.get('SQ_WAIT_INST_ANY', 0) / wc:5.2f} active {m.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.2f} "
          f"valu {m.get('SQ_ACTIVE_INST_VALU', 0) / wc:5.2f} lds {m.get('SQ_ACTIVE_INST_LDS', 0) / wc:5.2f} "
          f"vmem {m.get('SQ_ACTIVE_INST_VMEM', 0) / wc:5.2f}  insts_valu/wave {m.get('SQ_INSTS_VALU', 0) / max(m.get('SQ_WAVES', 1), 1):8.0f}")
