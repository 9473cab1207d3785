#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03p
mkdir -p $OUT
cd $R
timeout 500 python tools/ab_bench.py --tag em1 -v base -v MM_TEAM_EMIT_MAX=64 -v MM_TEAM_EMIT_MAX=128 -v base -- --mode 5v5 --steps 8 --warmup 2 --no-pcie --no-cfg3 --no-prediction > $OUT/ab1.txt 2>&1
cat $OUT/ab1.txt
timeout 500 python tools/ab_bench.py --tag em10 -v base -v MM_TEAM_EMIT_MAX=64 -v MM_TEAM_EMIT_MAX=128 -v MM_TEAM_FUSED=0 -v MM_TEAM_LIVE=0 -- --players 10000000 --mode 5v5 --steps 3 --warmup 1 --no-pcie --no-cfg3 --no-prediction > $OUT/ab10.txt 2>&1
cat $OUT/ab10.txt
