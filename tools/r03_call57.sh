#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03m
mkdir -p $OUT
cd $R
C="--mode 5v5 --steps 3 --warmup 1 --no-cpu-baseline --no-boundary"
run() { name=$1; shift; timeout 300 python bench.py $C "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; grep -v amdgpu.ids $OUT/$name.err | tail -2 | cut -c1-200; }
run pcie --no-stream --shared-players 0 --concurrent-pools 1
run shared10m --no-stream --no-pcie --no-prediction --concurrent-pools 1
run prediction --no-stream --no-pcie --concurrent-pools 1
run concurrent --no-stream --no-pcie --shared-players 0
run streams --no-secondary
