#!/bin/bash
# The measurement set behind profiles/rNN_* and DESIGN.md section 5, on the GPU box, in two calls (a gpurun call is 20 min at most):
#   bash scripts/refresh_profiles.sh r04 bench      the bench lines (default run with its secondary legs, the driver's flags, variants)
#   bash scripts/refresh_profiles.sh r04 prof       rocprofv3 kernel statistics, PMC passes, roofline table (same code, same box type)
# Outputs under gpurun_out/refresh_<part>/; copy what is to be kept into profiles/.
# rocprofv3: the program directly after "--"; counters in their own passes, without any trace option.
set -o pipefail
R=${1:-r04}
PART=${2:-bench}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/refresh_$PART
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
line() { grep '^{' "$1" | tail -1; }
if [ "$PART" = bench ]; then
timeout -k 10 700 python bench.py > $OUT/bench.log 2>&1 && line $OUT/bench.log > $OUT/${R}_bench.json && echo "bench done" &&
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $OUT/driver.log 2>&1 && line $OUT/driver.log > $OUT/${R}_bench_driver_flags.json && echo "driver-flags done" &&
timeout -k 10 300 python bench.py --only-leg exchange_world1 > $OUT/split.log 2>&1 && line $OUT/split.log > $OUT/${R}_bench_split_exchange.json && echo "exchange done" &&
timeout -k 10 200 python bench.py --host-inputs --no-cpu-baseline --no-secondary > $OUT/host.log 2>&1 && line $OUT/host.log > $OUT/${R}_bench_host_inputs.json && echo "host done" &&
timeout -k 10 200 python bench.py --arch 3sa --no-cpu-baseline > $OUT/3sa.log 2>&1 && line $OUT/3sa.log > $OUT/${R}_bench_3sa.json && echo "3sa done" &&
timeout -k 10 300 python bench.py --points 131072 --plots 8 --no-cpu-baseline > $OUT/c5.log 2>&1 && line $OUT/c5.log > $OUT/${R}_bench_131072pts.json && echo "c5 done" &&
timeout -k 10 300 python bench.py --dtype bf16 --points 131072 --plots 8 --no-cpu-baseline > $OUT/c5b.log 2>&1 && line $OUT/c5b.log > $OUT/${R}_bench_131072pts_bf16.json && echo "c5 bf16 done" &&
timeout -k 10 300 python scripts/bench_inference.py > $OUT/inf.log 2>&1 && line $OUT/inf.log > $OUT/${R}_bench_inference.json && echo "inference done"
else
cd /tmp && export TMPDIR=/tmp &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary > $OUT/prof.log 2>&1 &&
line $OUT/prof.log > $OUT/${R}_bench_under_rocprof.json && cp $(find $OUT/prof -name "*kernel_stats.csv" | tail -1) $OUT/${R}_kernel_stats.csv && echo "kernel stats done" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -- python3 $ROOT/bench.py --serial --eager --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/prof_serial.log 2>&1 &&
cp $(find $OUT/prof_serial -name "*kernel_stats.csv" | tail -1) $OUT/${R}_kernel_stats_serial.csv && echo "serial kernel stats done" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_inf -- python3 $ROOT/scripts/bench_inference.py --repeat 2 > $OUT/prof_inf.log 2>&1 &&
cp $(find $OUT/prof_inf -name "*kernel_stats.csv" | tail -1) $OUT/${R}_kernel_stats_inference.csv && echo "inference kernel stats done" &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --serial --eager --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_fetch.log 2>&1 && echo "fetch pass done" &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --serial --eager --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_write.log 2>&1 && echo "write pass done" &&
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --serial --eager --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_sq.log 2>&1 && cp $(find $OUT/pmc_sq -name "*counter_collection.csv" | tail -1) $OUT/${R}_pmc_sq.csv && echo "sq pass done" &&
cd $ROOT && python scripts/pmc_summary.py $(find $OUT/pmc_fetch -name "*counter_collection.csv" | tail -1) $(find $OUT/pmc_write -name "*counter_collection.csv" | tail -1) $OUT/${R}_pmc_traffic.json && echo "pmc summary done" &&
python scripts/roofline_table.py $ROOT/profiles/${R}_bench.json $OUT/${R}_kernel_stats_serial.csv $OUT/${R}_pmc_traffic.json $OUT/${R}_pmc_sq.csv $OUT/${R}_roofline_table > /dev/null && echo "roofline table done"
rm -rf $OUT/prof $OUT/prof_serial $OUT/prof_inf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
fi
ls -la $OUT
