mkdir -p gpurun_out/s10
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "team or 5v5 or starving or golden or random or stream" 2>&1 | tail -3
for fm in 96 128 200 100000; do
  echo "FIXMAX $fm"
  bash tools/quick_passes.sh 5v5 gpurun_out/s10/passes_$fm.txt MM_TEAM_FIXMAX=$fm | grep -E "kt_fc|kt_f,|kt_chase,|kt_late|span"
done
MM_PAIR_DEBUG=1 python bench.py --mode 5v5 --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-prediction 2>&1 >/dev/null | grep "mm-team" | tail -14 > gpurun_out/s10/team_timers.txt; grep "g6" gpurun_out/s10/team_timers.txt | cut -c1-1200
python bench.py --mode 5v5 --steps 10 --warmup 3 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-prediction > gpurun_out/s10/b5.json 2> gpurun_out/s10/b5.err; python -c "
import json
d=json.loads(open('gpurun_out/s10/b5.json').read().strip().splitlines()[-1])
print(d['value']/1e6, d['ms_per_step'], d['kernel_ms'], d['exactness']['ok'])
"
