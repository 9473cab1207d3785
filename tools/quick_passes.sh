#!/bin/bash
# kernel trace of a few ticks of bench.py and the per-pass durations of the last one (no PMC, no bench line)
#   usage (GPU box, repo root): bash tools/quick_passes.sh 5v5|1v1 <out file> [env assignments ...]
set -u
MODE=${1:-5v5}; OUTF=${2:-gpurun_out/passes.txt}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
if [ $MODE = 1v1 ]; then FIRST=kp_init; KERNELS="kp_rounds kp_round kp_group kp_late kp_nx_init kp_finish"
else FIRST=kt_init; KERNELS="kt_build kt_fc kt_f kt_f2 kt_chase kt_late"; fi
rm -rf /tmp/qp_$MODE
timeout 150 env "$@" rocprofv3 --kernel-trace -d /tmp/qp_$MODE -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-boundary --no-pcie --no-prediction --mode $MODE > /dev/null 2> /tmp/qp_$MODE.err
DB=$(find /tmp/qp_$MODE -name "*_results.db" | head -1)
python $R/tools/rocpd_passes.py $DB $FIRST $KERNELS > $R/$OUTF
cat $R/$OUTF
