#!/bin/bash
# kernel trace of the serial eager bench (one stream: a kernel's duration is its own): prints the kernels matching $2
# usage (on the GPU box): bash scripts/ktrace.sh OUTDIR 'pattern' [extra bench flags]
out=$1; pat=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -- python3 $root/bench.py --serial --eager --steps 20 --warmup 3 --no-cpu-baseline --no-secondary "$@" > $root/$out/prof.log 2>&1 || { tail -20 $root/$out/prof.log; exit 1; }
cd $root
find $out/prof -name "*kernel_stats.csv" | xargs -I{} cp {} $out/kstats.csv
rm -rf $out/prof
python3 - "$out/kstats.csv" "$pat" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if re.search(sys.argv[2], r["Name"]):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).split("(")[0].replace("void ", "")
        print(f'{name:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
