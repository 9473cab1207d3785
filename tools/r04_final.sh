#!/bin/bash
# the round's evidence in one gpurun call: profiles + PMC traffic (collect_profiles.sh), the driver's bench line,
# the 60 s cfg-5 streams, the 10M 5v5 pool, normal ratings, the heaviest chain of the 10M 1v1 pool alone, the micro-benchmarks,
# the gpu tests and the random-scenario stress
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
bash tools/collect_profiles.sh r04 both > $OUT/collect.log 2>&1
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_bench_full_1m_1v1.json 2> $OUT/bench_full.err
MM_PAIR_PERSIST=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-stream --no-secondary --no-cpu-baseline > $OUT/r04_bench_1m_1v1_one_launch_per_pass.json 2> $OUT/bench_persist0.err
python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --no-saturation --stream-seconds 60 > $OUT/r04_bench_stream60.json 2> $OUT/bench_stream60.err
python bench.py --players 10000000 --mode 5v5 --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-cfg3 --no-prediction > $OUT/r04_bench_10m_5v5.json 2> $OUT/bench_10m.err
python bench.py --dist normal --steps 10 --warmup 3 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-cfg3 --no-prediction > $OUT/r04_bench_1m_1v1_normal.json 2> $OUT/bench_normal.err
python bench.py --dist normal --mode 5v5 --steps 8 --warmup 2 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-cfg3 --no-prediction > $OUT/r04_bench_1m_5v5_normal.json 2> $OUT/bench_normal5.err
( cd /tmp && rm -rf /tmp/prof_h && rocprofv3 --kernel-trace -d /tmp/prof_h -- python $R/tools/heaviest_chain_tick.py 10000000 2 2> $OUT/heaviest.err > $OUT/r04_heaviest_chain_10m.txt; DB=$(find /tmp/prof_h -name "*_results.db" | head -1); python $R/tools/rocpd_passes.py $DB kp_init kp_rounds kp_round kp_group kp_late kp_nx_init kc_scatter >> $OUT/r04_heaviest_chain_10m.txt )
MM_PAIR_DEBUG=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stream --no-secondary 2>&1 > /dev/null | grep -E "kp_rounds:|tile1 cycles" | tail -14 > $OUT/r04_pair_phase_timers.txt
MM_PAIR_DEBUG=1 timeout 120 python bench.py --mode 5v5 --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-prediction 2>&1 > /dev/null | grep "mm-team" | tail -14 | grep -v chaser > $OUT/r04_team_phase_timers.txt
MM_TEAM_LATE=0 MM_PAIR_DEBUG=1 timeout 120 python bench.py --mode 5v5 --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-prediction 2>&1 > /dev/null | grep "mm-team" | tail -14 | grep chaser >> $OUT/r04_team_phase_timers.txt
for v in "mend:MM_X=1" "scratch:MM_TEAM_FIXMAX=0"; do n=${v%%:*}; e=${v#*:}; env $e timeout 120 python bench.py --mode 5v5 --steps 10 --warmup 3 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-prediction 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'variant':'$n','env':'$e','value':d['value'],'ms_per_step':d['ms_per_step'],'walk_ms':d['kernel_ms']['walk'],'exact':d['exactness']['ok']}))" >> $OUT/r04_ab_team_mend.jsonl; done
bash tools/quick_pmc.sh 5v5 gpurun_out/r04/r04_pmc_insts_1m_5v5.csv "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE" > /dev/null 2>&1
for u in xwg_hop token_handoff; do hipcc --offload-arch=gfx950 -O3 tools/ubench/$u.hip -o /tmp/$u 2>/dev/null && timeout 100 /tmp/$u > $OUT/r04_ubench_$u.txt; done
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/r04_pytest_gpu.log 2>&1
timeout 200 python tests/stress.py 60 8100000 >> $OUT/r04_pytest_gpu.log 2>&1
timeout 200 python tests/stress.py 60 8200000 team >> $OUT/r04_pytest_gpu.log 2>&1
tail -4 $OUT/r04_pytest_gpu.log | cut -c1-200
