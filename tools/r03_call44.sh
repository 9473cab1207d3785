#!/bin/bash
set -u
mkdir -p gpurun_out/r03d
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "launch_shape" > gpurun_out/r03d/pytest_shape.log 2>&1
tail -3 gpurun_out/r03d/pytest_shape.log | cut -c1-200
MM_TEAM_FUSED=1 MM_TEAM_F2=4 timeout 150 python tests/stress.py 50 7500000 team > gpurun_out/r03d/stress_team_fused.log 2>&1
tail -1 gpurun_out/r03d/stress_team_fused.log | cut -c1-300
timeout 600 python tools/ab_bench.py --tag both -v base -v MM_TEAM_FUSED=1 -v MM_TEAM_FUSED=1,MM_TEAM_LATE=12 -v MM_TEAM_FUSED=1,MM_TEAM_LATE=20 -v MM_TEAM_FUSED=1,MM_TEAM_F2=24 -- --mode 5v5 --steps 8 --warmup 2 --no-pcie --no-cfg3 --no-prediction > gpurun_out/r03d/ab_both.txt 2>&1
cat gpurun_out/r03d/ab_both.txt
MM_TEAM_FUSED=1 timeout 150 python tests/stress.py 40 7600000 team > gpurun_out/r03d/stress_team_fused2.log 2>&1
tail -1 gpurun_out/r03d/stress_team_fused2.log | cut -c1-300
