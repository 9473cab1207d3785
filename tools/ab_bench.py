#!/usr/bin/env python
"""A/B runs of bench.py in ONE gpurun call: each variant is a set of environment variables (the
engine's build-free knobs: MM_PAIR_BATCH, MM_PAIR_FUSED, MM_PAIR_TUNE, MM_TEAM_BATCH, MM_TEAM_CAP,
MM_FORCE_GENERIC) and/or extra bench.py arguments; prints one table row per variant and writes
the raw bench lines to gpurun_out/ab_<tag>.jsonl.

  python tools/ab_bench.py --tag batch -v base -v MM_PAIR_BATCH=16 -v MM_PAIR_BATCH=64 -- --steps 5 --warmup 2
  python tools/ab_bench.py --tag team  -v base -v MM_TEAM_BATCH=8,MM_TEAM_CAP=256 -- --mode 5v5

`--script` replaces bench.py (tests/bench_dryrun_worker.py runs the same thing on the CPU shim)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="ab")
    ap.add_argument("-v", "--variant", action="append", required=True,
                    help="'base' or comma-separated NAME=VALUE environment settings")
    ap.add_argument("--script", default=os.path.join(ROOT, "bench.py"))
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("bench_args", nargs="*", help="after --: passed to bench.py")
    a = ap.parse_args()
    base_args = ["--no-cpu-baseline", "--no-stream", "--no-secondary"] + a.bench_args
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    out_path = os.path.join(ROOT, "gpurun_out", "ab_%s.jsonl" % a.tag)
    rows = []
    with open(out_path, "w") as out:
        for v in a.variant:
            env = dict(os.environ)
            if v != "base":
                for kv in v.split(","):
                    k, _, val = kv.partition("=")
                    env[k] = val
            best = None
            for _ in range(a.repeat):
                p = subprocess.run([sys.executable, a.script] + base_args, env=env, stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, cwd=ROOT)
                lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
                if p.returncode != 0 or not lines:
                    print("variant %s FAILED (exit %d): %s" % (v, p.returncode, p.stderr.decode()[-400:]))
                    continue
                d = json.loads(lines[-1])
                d["variant"] = v
                out.write(json.dumps(d) + "\n")
                if best is None or d["ms_per_step"] < best["ms_per_step"]:
                    best = d
            if best:
                rows.append(best)
    print("%-44s %12s %12s %16s %8s" % ("variant", "ms/step", "walk ms", "matched pl/s", "passes"))
    for d in rows:
        print("%-44s %12.3f %12.3f %16.0f %8d" % (d["variant"], d["ms_per_step"], d["kernel_ms"]["walk"], d["value"],
                                                   d["passes_max"]))
    print("raw lines:", out_path)


if __name__ == "__main__":
    main()
