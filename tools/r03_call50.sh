#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03j
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "launch_shape or team" 2>&1 | tail -2 | cut -c1-200
timeout 400 python tools/ab_bench.py --tag c4 -v base -v MM_TEAM_LATE=12 -v MM_TEAM_F2=24 -v base -- --mode 5v5 --steps 8 --warmup 2 --no-pcie --no-cfg3 --no-prediction > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
B="--mode 5v5 --steps 1 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-cfg3 --no-prediction --no-boundary"
MM_PAIR_DEBUG=1 MM_TEAM_LATE=0 timeout 200 python bench.py $B > /dev/null 2> $OUT/dbg_base.err
grep "kt_fc's chaser" $OUT/dbg_base.err | tail -7 | cut -c1-300
MM_PAIR_DEBUG=1 MM_TEAM_LATE=80 timeout 200 python bench.py $B > /dev/null 2> $OUT/dbg_late80.err
grep "mm-team-late" $OUT/dbg_late80.err | head -2 | cut -c1-420
timeout 100 python tests/stress.py 40 8000000 team 2>&1 | tail -1 | cut -c1-200
