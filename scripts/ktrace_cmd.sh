#!/bin/bash
# kernel trace of an arbitrary python script: bash scripts/ktrace_cmd.sh OUTDIR N script.py [args]  -> top N kernels by total time
out=$1; n=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof -- python3 $root/"$@" > $root/$out/prof.log 2>&1 || { tail -20 $root/$out/prof.log; exit 1; }
cd $root
find $out/prof -name "*kernel_stats.csv" | xargs -I{} cp {} $out/kstats.csv
rm -rf $out/prof
python3 - "$out/kstats.csv" "$n" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2])]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).split("(")[0].replace("void ", "")
    print(f'{name[:70]:70s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms {100 * float(r["TotalDurationNs"]) / tot:5.1f} %')
PY
