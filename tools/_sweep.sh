mkdir -p gpurun_out/s14
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "team or 5v5 or starving or golden or random or stream" 2>&1 | tail -3
bash tools/quick_passes.sh 5v5 gpurun_out/s14/passes.txt MM_X=1 | grep -E "kt_fc|kt_f,|kt_chase,|kt_late|span"
MM_TEAM_LATE=0 MM_PAIR_DEBUG=1 python bench.py --mode 5v5 --steps 3 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-prediction 2>&1 >/dev/null | grep "mm-team" | tail -14 | grep "g0 kt_fc" | cut -c1-600
python bench.py --mode 5v5 --steps 10 --warmup 3 --no-cpu-baseline --no-stream --no-secondary --no-pcie --no-prediction > gpurun_out/s14/b5.json 2> gpurun_out/s14/b5.err; python -c "
import json
d=json.loads(open('gpurun_out/s14/b5.json').read().strip().splitlines()[-1])
print(d['value']/1e6, d['ms_per_step'], d['kernel_ms'], d['exactness']['ok'])
"
