cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MM_PAIR_DEBUG=1 timeout 60 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/dbg_1m.err > gpurun_out/bench_1m_v3.json < /dev/null
python -c "
import json; d=json.load(open('gpurun_out/bench_1m_v3.json')); print(d['kernel_ms'], d['ms_per_step'])"
tail -14 gpurun_out/dbg_1m.err | cut -c1-360 | grep -v "^\[mm-pair\] g[1-5] fast"
timeout 120 python -m pytest tests/test_gpu_parity.py -x -q -k "1v1 or golden or edge or device or stream" < /dev/null 2>&1 | tail -3
