mkdir -p gpurun_out/s9
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "team or 5v5 or starving or golden" 2>&1 | tail -3
for fm in 0 48 96 100000; do
  echo "TB8 FIXMAX $fm"
  bash tools/quick_passes.sh 5v5 gpurun_out/s9/passes_$fm.txt MM_TEAM_FIXMAX=$fm | grep -E "kt_fc|kt_f,|span"
done
for lib in TB4 TB2; do for fm in 96 100000; do
  echo "$lib FIXMAX $fm"
  bash tools/quick_passes.sh 5v5 gpurun_out/s9/passes_${lib}_$fm.txt MM_TEAM_FIXMAX=$fm MM_ENGINE_LIB=$PWD/tools/_lib_$lib.so | grep -E "kt_fc|kt_f,|span"
done; done
