#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03l
mkdir -p $OUT
cd $R
timeout 600 python tools/ab_bench.py --tag cap -v base -v MM_TEAM_CAP=256 -v MM_TEAM_CAP=1024 -v MM_TEAM_CAP=2048 -v MM_TEAM_REBUILD=4 -v MM_TEAM_REBUILD=16 -v MM_TEAM_BATCH=32 -v base -- --mode 5v5 --steps 8 --warmup 2 --no-pcie --no-cfg3 --no-prediction > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
