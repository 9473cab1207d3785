#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03n
mkdir -p $OUT
cd $R
for i in 1 2 3; do
timeout 400 python bench.py --mode 5v5 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/full5v5_$i.json 2> $OUT/full5v5_$i.err; echo "full5v5 $i rc=$?"; grep -v amdgpu.ids $OUT/full5v5_$i.err | tail -2 | cut -c1-200
done
