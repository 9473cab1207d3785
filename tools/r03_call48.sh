#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03i
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
B="--mode 5v5 --steps 1 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-cfg3 --no-prediction --no-boundary"
MM_PAIR_DEBUG=1 MM_TEAM_LATE=0 timeout 200 python bench.py $B > /dev/null 2> $OUT/dbg_base.err
MM_PAIR_DEBUG=1 MM_TEAM_LATE=0 MM_TEAM_F2=0 timeout 200 python bench.py $B > /dev/null 2> $OUT/dbg_f20.err
MM_PAIR_DEBUG=1 MM_TEAM_LATE=0 MM_TEAM_F2=0 MM_TEAM_FUSED=0 timeout 200 python bench.py $B > /dev/null 2> $OUT/dbg_f20_nofused.err
grep "kt_fc's chaser" $OUT/dbg_base.err | tail -7
grep "kt_fc's chaser" $OUT/dbg_f20.err | tail -7
grep "kt_fc's chaser" $OUT/dbg_f20_nofused.err | tail -7
