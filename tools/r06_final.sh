#!/bin/bash
# the round's evidence in one gpurun call: profiles + PMC traffic (collect_profiles.sh), the driver's bench line, the timelines
# (idle gaps), the 60 s cfg-5 streams, the 10M 5v5 pool, normal ratings, the heaviest chain of the 10M 1v1 pool alone, the phase
# timers, the gpu tests and the random-scenario stress
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=r06
OUT=$R/gpurun_out/$T
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
bash tools/collect_profiles.sh $T both > $OUT/collect.log 2>&1
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${T}_bench_full_1m_1v1.json 2> $OUT/bench_full.err
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-stream --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', [round(x,3) for x in d['ms_per_step_min_median_max']], 'walk', [round(x,3) for x in d['walk_ms_min_median_max']], 'degraded', d['degraded'], 'exact', d['exactness']['ok'])" >> $OUT/${T}_bench_headline_repeats.txt; done
MM_PAIR_PERSIST=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-stream --no-secondary --no-cpu-baseline > $OUT/${T}_bench_1m_1v1_one_launch_per_pass.json 2> $OUT/bench_persist0.err
python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --no-saturation --stream-seconds 60 > $OUT/${T}_bench_stream60.json 2> $OUT/bench_stream60.err
python bench.py --players 10000000 --mode 5v5 --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-cfg3 --no-prediction > $OUT/${T}_bench_10m_5v5.json 2> $OUT/bench_10m.err
python bench.py --dist normal --steps 10 --warmup 3 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-cfg3 --no-prediction > $OUT/${T}_bench_1m_1v1_normal.json 2> $OUT/bench_normal.err
python bench.py --dist normal --mode 5v5 --steps 8 --warmup 2 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-cfg3 --no-prediction > $OUT/${T}_bench_1m_5v5_normal.json 2> $OUT/bench_normal5.err
( cd /tmp && for M in 1v1 5v5; do rm -rf /tmp/qg_$M; timeout 200 rocprofv3 --kernel-trace -d /tmp/qg_$M -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-boundary --no-pcie --no-prediction --mode $M > /dev/null 2> /tmp/qg_$M.err; DB=$(find /tmp/qg_$M -name "*_results.db" | head -1); if [ $M = 1v1 ]; then F=kp_init; else F=kt_init; fi; python $R/tools/rocpd_gaps.py $DB $F 6 > $OUT/${T}_timeline_gaps_1m_$M.txt 2>&1; done )
( cd /tmp && rm -rf /tmp/prof_h && rocprofv3 --kernel-trace -d /tmp/prof_h -- python $R/tools/heaviest_chain_tick.py 10000000 2 2> $OUT/heaviest.err > $OUT/${T}_heaviest_chain_10m.txt; DB=$(find /tmp/prof_h -name "*_results.db" | head -1); python $R/tools/rocpd_passes.py $DB kp_init kp_rounds kp_round kp_group kp_late kp_nx_init kc_scatter >> $OUT/${T}_heaviest_chain_10m.txt )
MM_PAIR_DEBUG=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stream --no-secondary 2>&1 > /dev/null | grep -E "kp_rounds:|tile1 cycles|g0 fast" | tail -16 > $OUT/${T}_pair_phase_timers.txt
MM_PAIR_DEBUG=1 timeout 120 python bench.py --mode 5v5 --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-prediction 2>&1 > /dev/null | grep "mm-team" | tail -14 | grep -v chaser > $OUT/${T}_team_phase_timers.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/${T}_pytest_gpu.log 2>&1
timeout 200 python tests/stress.py 40 9100000 >> $OUT/${T}_pytest_gpu.log 2>&1
timeout 200 python tests/stress.py 40 9200000 team >> $OUT/${T}_pytest_gpu.log 2>&1
# round 6: random COMBINATIONS of the tuning fields per scenario (the long soak: tools/r06_soak.sh, profiles/r06_stress_fuzz_knobs.txt)
timeout 200 python tests/stress.py 60 9300000 --fuzz-knobs >> $OUT/${T}_pytest_gpu.log 2>&1
timeout 200 python tests/stress.py 60 9400000 team --fuzz-knobs >> $OUT/${T}_pytest_gpu.log 2>&1
# the N > 1 branch of the bench line on the box's one GPU (two ranks, gloo, real HIP engines): not a scaling number
python bench.py --gpus 2 --same-device --steps 3 --warmup 1 --no-secondary --no-stream > $OUT/${T}_bench_2ranks_same_device.json 2> $OUT/bench_2ranks.err
tail -4 $OUT/${T}_pytest_gpu.log | cut -c1-200
cat $OUT/${T}_bench_headline_repeats.txt
