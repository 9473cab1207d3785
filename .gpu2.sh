cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MM_PAIR_DEBUG=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/dbg_1m.err > gpurun_out/bench_1m_v2.json < /dev/null
cat gpurun_out/bench_1m_v2.json
tail -7 gpurun_out/dbg_1m.err
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "1v1 or golden or edge or device or stream" < /dev/null 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_1m -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_1m.log 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT; python tools/rocpd_stats.py $(find gpurun_out/prof_1m -name "*.db" | head -1) < /dev/null
