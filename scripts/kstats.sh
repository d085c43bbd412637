#!/bin/bash
# rocprofv3 kernel statistics of ANY bench.py invocation: `bash scripts/kstats.sh OUTDIR [bench flags]` (on the GPU box;
# SN2_KSTATS_PROG=scripts/bench_inference.py profiles that program instead).
# Writes OUTDIR/kstats.csv (the --stats summary) and prints the kernels by total time.  The program follows "--" directly.
out=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
prog=${SN2_KSTATS_PROG:-bench.py}      # another program of the repository, e.g. scripts/bench_inference.py
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -- python3 $root/$prog "$@" > $root/$out/prof.log 2>&1 || { tail -20 $root/$out/prof.log; exit 1; }
cd $root
find $out/prof -name "*kernel_stats.csv" | xargs -I{} cp {} $out/kstats.csv
rm -rf $out/prof
python3 - "$out/kstats.csv" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms")
for r in rows[:60]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).split("(")[0].replace("void ", "")
    print(f'{name[:70]:70s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms  {r["Percentage"]:>6s} %')
PY
