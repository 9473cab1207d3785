#!/bin/bash
# a few counters per kernel for a couple of ticks of bench.py (each counter set in a run of its own: --kernel-trace + --pmc only)
#   usage (GPU box, repo root): bash tools/quick_pmc.sh 5v5|1v1 <out file> "<counters set 1>" ["<counters set 2>" ...]
set -u
MODE=${1:-5v5}; OUTF=${2:-gpurun_out/pmc.txt}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
: > $R/$OUTF
for SET in "$@"; do
  rm -rf /tmp/qc_$MODE
  timeout 150 rocprofv3 --kernel-trace --pmc $SET -d /tmp/qc_$MODE -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-boundary --no-pcie --no-prediction --mode $MODE > /dev/null 2> /tmp/qc_$MODE.err
  DB=$(find /tmp/qc_$MODE -name "*_results.db" | head -1)
  python $R/tools/rocpd_pmc.py $DB 2>/dev/null | grep -E "kt_|kp_|kernel," >> $R/$OUTF
done
cat $R/$OUTF
