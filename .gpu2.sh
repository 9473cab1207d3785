cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tune in 0 0x83; do
  echo "== tune $tune"
  MM_PAIR_TUNE=$tune MM_PAIR_DEBUG=1 timeout 120 python bench.py --players 65536 --steps 3 --warmup 2 --no-cpu-baseline 2> gpurun_out/dbg_$tune.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['passes_max'])" < /dev/stdin
  tail -7 gpurun_out/dbg_$tune.err
done
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_b -- python $GRAFT_REPO_ROOT/bench.py --players 65536 --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_b.log 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT; python tools/rocpd_stats.py $(find gpurun_out/prof_b -name "*.db" | head -1) < /dev/null
