cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06i_smoke.txt 2>&1
bash tools/r06_soak.sh 220 2 > /dev/null 2>&1
cp gpurun_out/r06_stress_fuzz_knobs.txt gpurun_out/r06i_soak2.txt
