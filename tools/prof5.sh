cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf /tmp/p5 && rocprofv3 --kernel-trace -d /tmp/p5 -- python $R/bench.py --mode 5v5 --steps 2 --warmup 1 --no-cpu-baseline --no-stream --no-secondary > /dev/null 2>&1
DB=$(find /tmp/p5 -name "*_results.db" | head -1)
python $R/tools/rocpd_passes.py $DB kt_init kt_build kt_f kt_f2 kt_chase kt_emit
