#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03g
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "launch_shape or team" > $OUT/pytest_team.log 2>&1
tail -2 $OUT/pytest_team.log | cut -c1-200
timeout 600 python tools/ab_bench.py --tag pull -v base -v MM_TEAM_F2=0 -v MM_TEAM_F2=16 -v MM_TEAM_LATE=12 -v MM_TEAM_LATE=20 -v base -- --mode 5v5 --steps 8 --warmup 2 --no-pcie --no-cfg3 --no-prediction > $OUT/ab_pull.txt 2>&1
cat $OUT/ab_pull.txt
timeout 150 python tests/stress.py 40 7700000 team > $OUT/stress_team.log 2>&1
MM_TEAM_F2=0 timeout 150 python tests/stress.py 40 7800000 team > $OUT/stress_team_f20.log 2>&1
tail -1 $OUT/stress_team.log $OUT/stress_team_f20.log | cut -c1-300
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-boundary --no-pcie --no-cfg3 --no-prediction"
for V in 32 0; do
rm -rf /tmp/prof_5v5 && MM_TEAM_F2=$V timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_5v5 -- $BENCH --mode 5v5 > /dev/null 2> $OUT/rocprof.err
DB=$(find /tmp/prof_5v5 -name "*_results.db" | head -1)
python $R/tools/rocpd_passes.py "$DB" kt_init kt_build kt_fc kt_f kt_f2 kt_chase kt_emit kt_late > $OUT/kernel_passes_5v5_f2_$V.txt
cat $OUT/kernel_passes_5v5_f2_$V.txt | cut -c1-300
done
