#!/bin/bash
# rocprofv3 kernel stats of a short serial eager bench run; prints the top kernels (scratch tool for tuning sessions)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_top
rm -rf $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --serial --eager --steps 5 --warmup 2 --no-cpu-baseline "$@" > $out.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_top -name "*kernel_stats.csv" | tail -1)
python - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:${TOPN:-45}]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(4), "%9.1f" % (float(r["AverageNs"])/1000), r["Percentage"])
PY
