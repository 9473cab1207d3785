cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q < /dev/null 2>&1 | tail -6 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
timeout 200 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err < /dev/null; cat gpurun_out/bench_default.json | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_kt.log 2>&1 < /dev/null
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_fetch.log 2>&1 < /dev/null
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof_write -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_write.log 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
for d in prof_kt prof_fetch prof_write; do ls -la gpurun_out/$d/*/ 2>/dev/null | tail -2; done
python tools/rocpd_stats.py $(find gpurun_out/prof_kt -name "*.db" | head -1) < /dev/null | head -12
