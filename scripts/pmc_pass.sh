#!/bin/bash
# one rocprofv3 --pmc pass over the serial eager bench; prints scripts/pmc_sq.py's per-kernel means for kernels matching $2..
# usage (on the GPU box): bash scripts/pmc_pass.sh OUTDIR "COUNTER1 COUNTER2 ..." [name filter ...]
out=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $root/$out/pmc -- python3 $root/bench.py --serial --eager --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $root/$out/pmc.log 2>&1 || { tail -20 $root/$out/pmc.log; exit 1; }
cd $root
cp $(find $out/pmc -name "*counter_collection.csv" | tail -1) $out/pmc_counters.csv
rm -rf $out/pmc
python3 scripts/pmc_sq.py $out/pmc_counters.csv "$@"
python3 - $out/pmc_counters.csv "$@" <<'PY'
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name).split("(")[0]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if sys.argv[2:] and not any(f in k for f in sys.argv[2:]):
        continue
    print(k[:60], {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
