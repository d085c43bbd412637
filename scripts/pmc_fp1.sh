#!/bin/bash
# SQ counters of FP1's forward kernels (scripts/time_fp1.py) in two rocprofv3 --pmc passes; usage (GPU box): bash scripts/pmc_fp1.sh OUTDIR
out=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
n=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  n=$((n+1))
  timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $root/$out/pmc$n -- python3 $root/scripts/time_fp1.py > $root/$out/pmc$n.log 2>&1 || { tail -5 $root/$out/pmc$n.log; continue; }
  cp $(find $root/$out/pmc$n -name "*counter_collection.csv" | tail -1) $root/$out/pmc_counters$n.csv
  rm -rf $root/$out/pmc$n
done
cd $root
python3 - $out <<'PY'
import collections, csv, glob, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/pmc_counters*.csv")):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        if "fp_fwd_rows" in name or "fp_src_table" in name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:24s} {sum(v) / len(v):16.0f}   ({len(v)} launches)")
PY
