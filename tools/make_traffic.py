#!/usr/bin/env python
"""profiles/traffic_latest*.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py`: sums the walk's kernels per tick and applies the gfx950 corrections of
MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request: read side doubled; both in KB).
Usage: python tools/make_traffic.py <fetch.csv> <write.csv> <mode> <players> <ticks> <kernel prefixes, comma separated> [kernel stats csv] > out.json
The CSVs are the output of tools/rocpd_pmc.py (kernel,counter,dispatches,sum); the optional seventh argument is the
--kernel-trace summary of the same command (tools/rocpd_stats.py: kernel,calls,total_ns,...), from which the file also
carries every walk kernel's duration per tick (bench.py's roofline.by_kernel; round 6)."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402  (bench.py refuses a file measured on other kernel sources)


def load(path, prefixes):
    out = {}
    for row in csv.DictReader(open(path)):
        k = row["kernel"].replace("void ", "")          # template instances are listed as "void kp_round<2048u>"
        if any(k.startswith(p) for p in prefixes):
            out[k] = (float(row["sum"]), int(row["dispatches"]))
    return out


def load_stats(path, prefixes, ticks):
    out = {}
    for row in csv.DictReader(open(path)):
        k = row["kernel"].replace("void ", "")
        if any(k.startswith(p) for p in prefixes):
            out[k] = {"us_per_tick": float(row["total_ns"]) / ticks / 1e3, "calls_per_tick": int(row["calls"]) / ticks,
                      "avg_us": float(row["avg_ns"]) / 1e3}
    return out


def main(fetch_csv, write_csv, mode, players, ticks, prefixes, stats_csv=None):
    ticks = float(ticks)
    prefixes = prefixes.split(",")
    f, w = load(fetch_csv, prefixes), load(write_csv, prefixes)
    fetch = sum(v[0] for v in f.values()) / ticks
    write = sum(v[0] for v in w.values()) / ticks
    per = {k: {"fetch_raw": f.get(k, (0, 0))[0] / ticks, "write_raw": w.get(k, (0, 0))[0] / ticks,
               "dispatches_per_tick": f.get(k, w.get(k, (0, 0)))[1] / ticks} for k in sorted(set(f) | set(w))}
    print(json.dumps({
        "workload_players": int(players), "mode": mode, "ticks_profiled": ticks,
        "source_hash": kernel_source_hash(),
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) of bench.py; sums over "
                  "the walk's kernels (%s*) divided by the ticks" % "*, ".join(prefixes),
        "fetch_size_kb_per_tick_raw": fetch, "write_size_kb_per_tick_raw": write,
        "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request: read side doubled (MI355X_MICROARCH.md "
                      "section HBM); WRITE_SIZE uncalibrated, taken as is; both x1024 (KB)",
        "walk_hbm_bytes_per_tick": (2.0 * fetch + write) * 1024.0,
        "per_kernel_kb_per_tick": per,
        "per_kernel_time": load_stats(stats_csv, prefixes, ticks) if stats_csv else None,
        "note": "Infinity-Cache hits are counted by these counters; the working set of a 1M-player pool is cache resident",
    }, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:8])
