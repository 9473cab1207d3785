#!/bin/bash
# the builder-side soak of tests/stress.py --fuzz-knobs — random COMBINATIONS of mm_tuning fields per scenario,
# every tick bit-exact against the oracle.  Usage (one gpurun call): bash tools/r06_soak.sh <seconds per leg> <legs>
cd ${GRAFT_REPO_ROOT:-.}
SECS=${1:-280}; LEGS=${2:-4}
OUT=gpurun_out/r06_stress_fuzz_knobs.txt
: > $OUT
for i in $(seq 1 $LEGS); do
  ( timeout $((SECS + 120)) python tests/stress.py $SECS $((6300 + i)) --fuzz-knobs 2>&1 | grep -v amdgpu.ids | tail -4 ) >> $OUT
  ( timeout $((SECS + 120)) python tests/stress.py $SECS $((6400 + i)) team --fuzz-knobs 2>&1 | grep -v amdgpu.ids | tail -4 ) >> $OUT
done
cat $OUT
