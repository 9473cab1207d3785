#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 400 python tools/ab_bench.py --tag wpe -v base -v MM_ENGINE_LIB=$R/microservice_matchmaking_amd/csrc/libmm_engine_b.so -v base -v MM_ENGINE_LIB=$R/microservice_matchmaking_amd/csrc/libmm_engine_b.so -- --mode 5v5 --steps 8 --warmup 2 --no-pcie --no-cfg3 --no-prediction > $OUT/ab_wpe.txt 2>&1
cat $OUT/ab_wpe.txt
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-boundary --no-pcie --no-cfg3 --no-prediction"
rm -rf /tmp/prof_5v5 && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_5v5 -- $BENCH --mode 5v5 > /dev/null 2> $OUT/rocprof.err
DB=$(find /tmp/prof_5v5 -name "*_results.db" | head -1)
python $R/tools/rocpd_stats.py "$DB" > $OUT/kernel_stats_5v5.csv
python $R/tools/rocpd_passes.py "$DB" kt_init kt_build kt_fc kt_f kt_f2 kt_chase kt_emit kt_late > $OUT/kernel_passes_5v5.txt
cat $OUT/kernel_passes_5v5.txt | cut -c1-300
