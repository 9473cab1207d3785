#!/bin/bash
# the round's evidence in one gpurun call: profiles + PMC traffic (collect_profiles.sh), the driver's bench line,
# the 60 s cfg-5 streams, the 10M 5v5 pool, normal ratings, the gpu tests and the random-scenario stress
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
bash tools/collect_profiles.sh r03 both > $OUT/collect.log 2>&1
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r03_bench_full_1m_1v1.json 2> $OUT/bench_full.err
python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --stream-seconds 60 > $OUT/r03_bench_stream60.json 2> $OUT/bench_stream60.err
python bench.py --players 10000000 --mode 5v5 --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-cfg3 --no-prediction > $OUT/r03_bench_10m_5v5.json 2> $OUT/bench_10m.err
python bench.py --dist normal --steps 10 --warmup 3 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-cfg3 --no-prediction > $OUT/r03_bench_1m_1v1_normal.json 2> $OUT/bench_normal.err
python bench.py --dist normal --mode 5v5 --steps 8 --warmup 2 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-cfg3 --no-prediction > $OUT/r03_bench_1m_5v5_normal.json 2> $OUT/bench_normal5.err
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/r03_pytest_gpu.log 2>&1
timeout 200 python tests/stress.py 60 7100000 >> $OUT/r03_pytest_gpu.log 2>&1
timeout 200 python tests/stress.py 90 7200000 team >> $OUT/r03_pytest_gpu.log 2>&1
tail -4 $OUT/r03_pytest_gpu.log | cut -c1-200
