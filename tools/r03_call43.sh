#!/bin/bash
# one gpurun call: kt_fc (kt_f + chase in one launch) and the emitters in the chase's launch, A/B on cfg-3
set -u
mkdir -p gpurun_out/r03c
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "team or launch_shape or 5v5" > gpurun_out/r03c/pytest_team.log 2>&1
tail -3 gpurun_out/r03c/pytest_team.log
timeout 600 python tools/ab_bench.py --tag live -v base -v MM_TEAM_LIVE=0 -v MM_TEAM_FUSED=1 -v MM_TEAM_LIVE=0,MM_TEAM_FUSED=1 -v MM_TEAM_LATE=12 -v MM_TEAM_F2=16 -- --mode 5v5 --steps 8 --warmup 2 --no-pcie --no-cfg3 --no-prediction > gpurun_out/r03c/ab_live.txt 2>&1
cat gpurun_out/r03c/ab_live.txt
timeout 150 python tests/stress.py 50 7400000 team > gpurun_out/r03c/stress_team.log 2>&1
MM_TEAM_FUSED=1 MM_TEAM_F2=4 timeout 150 python tests/stress.py 40 7500000 team > gpurun_out/r03c/stress_team_fused.log 2>&1
tail -1 gpurun_out/r03c/stress_team.log gpurun_out/r03c/stress_team_fused.log
