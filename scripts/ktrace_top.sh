#!/bin/bash
# kernel trace of the serial eager bench; top N kernels by time per step: bash scripts/ktrace_top.sh OUTDIR N [bench flags]
out=$1; n=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -- python3 $root/bench.py --serial --eager --steps 20 --warmup 3 --no-cpu-baseline --no-secondary "$@" > $root/$out/prof.log 2>&1 || { tail -20 $root/$out/prof.log; exit 1; }
cd $root
find $out/prof -name "*kernel_stats.csv" | xargs -I{} cp {} $out/kstats.csv
rm -rf $out/prof
python3 - "$out/kstats.csv" "$n" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
# steps the run really made (timed + warm-up + the instrumented ones around them, whatever the flags were): every step
# launches the Adam kernel exactly once
steps = next((float(r["Calls"]) for r in rows if r["Name"].startswith("adam_kernel")), 26.0)
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3
print(f"all kernels: {tot:.1f} us per step")
for r in rows[:int(sys.argv[2])]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).split("(")[0].replace("void ", "")
    print(f'{name[:64]:64s} calls/step {float(r["Calls"]) / steps:5.1f} avg {float(r["AverageNs"]) / 1e3:8.1f} us  per step {float(r["TotalDurationNs"]) / steps / 1e3:8.1f} us')
PY
