#!/bin/bash
# One gpurun call's worth of profiling for a round: kernel stats, per-pass durations and the two PMC
# passes (FETCH_SIZE / WRITE_SIZE, separately, --kernel-trace only: MI355X_MICROARCH.md) of bench.py
# for BASELINE cfg-2 (1v1) and cfg-3 (5v5), plus the bench lines themselves.
#   usage (on the GPU box, from the repo root):  bash tools/collect_profiles.sh r02 [1v1|5v5|both]
# Writes gpurun_out/<tag>/...; copy what is to be judged into profiles/ (tools/make_traffic.py output
# goes to profiles/traffic_latest*.json directly: bench.py reads it and checks the kernel source hash).
set -u
TAG=${1:-r03}
WHAT=${2:-both}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stream --no-secondary --no-boundary --no-probe"
for MODE in 1v1 5v5; do
  [ "$WHAT" = both ] || [ "$WHAT" = "$MODE" ] || continue
  if [ $MODE = 1v1 ]; then FIRST=kp_init; KERNELS="kp_rounds kp_round kp_group kp_late kp_nx_init kp_init kp_finish"; PFX="kp_,kc_"; TJ=traffic_latest.json
  else FIRST=kt_init; KERNELS="kt_build kt_fc kt_f kt_f2 kt_chase kt_late"; PFX="kt_"; TJ=traffic_latest_5v5.json; fi
  rm -rf /tmp/prof_$MODE && rocprofv3 --kernel-trace -d /tmp/prof_$MODE -- $BENCH --mode $MODE > /dev/null 2> "$OUT/rocprof_$MODE.err"
  DB=$(find /tmp/prof_$MODE -name "*_results.db" | head -1)
  python "$R/tools/rocpd_stats.py" "$DB" > "$OUT/${TAG}_kernel_stats_1m_$MODE.csv"
  python "$R/tools/rocpd_passes.py" "$DB" $FIRST $KERNELS > "$OUT/${TAG}_kernel_passes_1m_$MODE.txt"
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$MODE && rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$MODE -- $BENCH --mode $MODE > /dev/null 2>> "$OUT/rocprof_$MODE.err"
    DB=$(find /tmp/pmc_$MODE -name "*_results.db" | head -1)
    python "$R/tools/rocpd_pmc.py" "$DB" $C > "$OUT/${TAG}_pmc_$(echo $C | tr A-Z a-z)_1m_$MODE.csv" 2>> "$OUT/rocprof_$MODE.err"
  done
  # 4 timed + 1 warm-up tick were profiled
  python "$R/tools/make_traffic.py" "$OUT/${TAG}_pmc_fetch_size_1m_$MODE.csv" "$OUT/${TAG}_pmc_write_size_1m_$MODE.csv" $MODE 1000000 5 "$PFX" "$OUT/${TAG}_kernel_stats_1m_$MODE.csv" > "$OUT/$TJ"
  cp "$OUT/$TJ" "$R/profiles/$TJ"
  python "$R/bench.py" --mode $MODE --steps 20 --warmup 5 > "$OUT/${TAG}_bench_1m_$MODE.json" 2> "$OUT/bench_$MODE.err"
  tail -c 600 "$OUT/${TAG}_kernel_passes_1m_$MODE.txt"; head -6 "$OUT/${TAG}_kernel_stats_1m_$MODE.csv"
done
