#!/bin/bash
# one gpurun call: fused emission A/B on cfg-3, kt_late thresholds, kt_late cycle counts, gpu tests
set -u
mkdir -p gpurun_out/r03b
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 600 python tools/ab_bench.py --tag fused -v base -v MM_TEAM_FUSED=0 -v MM_TEAM_LATE=12 -v MM_TEAM_LATE=20 -v MM_TEAM_LATE=0 -- --mode 5v5 --steps 8 --warmup 2 --no-pcie --no-cfg3 --no-prediction > gpurun_out/r03b/ab_fused.txt 2>&1
MM_PAIR_DEBUG=1 MM_TEAM_LATE=80 timeout 300 python bench.py --mode 5v5 --steps 1 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-cfg3 --no-prediction --no-boundary > gpurun_out/r03b/late80_debug.json 2> gpurun_out/r03b/late80_debug.err
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r03b/pytest_gpu.log 2>&1
timeout 300 python tools/ab_bench.py --tag pair42 -v base -- --steps 10 --warmup 3 --no-pcie --no-cfg3 --no-prediction > gpurun_out/r03b/ab_pair.txt 2>&1
timeout 200 python tests/stress.py 60 7300000 team > gpurun_out/r03b/stress_team.log 2>&1
tail -3 gpurun_out/r03b/ab_fused.txt
