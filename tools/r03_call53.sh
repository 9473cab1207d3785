#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03k
mkdir -p $OUT
cd $R
B="--mode 5v5 --steps 1 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-cfg3 --no-prediction --no-boundary"
MM_PAIR_DEBUG=1 MM_TEAM_LATE=0 timeout 200 python bench.py $B > /dev/null 2> $OUT/dbg_base.err
grep "kt_fc's chaser" $OUT/dbg_base.err | tail -14 | cut -c1-330
