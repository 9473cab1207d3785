#!/usr/bin/env python
"""Kernel-time summary (the `--stats` table) from a rocprofv3 rocpd SQLite database.

rocprofv3 on ROCm 7.2 writes `<pid>_results.db` by default; this prints per-kernel
calls / total / average / min / max (ns) and the vgpr / sgpr / LDS figures, as CSV.
Usage: python tools/rocpd_stats.py gpurun_out/prof/<host>/<pid>_results.db > profiles/x.csv
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x), max(grid_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,pct,vgpr,sgpr,lds_bytes,workgroup_x,grid_x")
    for r in rows:
        name = r[0].split("(")[0]
        print("%s,%d,%d,%.1f,%d,%d,%.2f,%s,%s,%s,%s,%s" % (name, r[1], r[2], r[3], r[4], r[5],
                                                        100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10]))


if __name__ == "__main__":
    main(sys.argv[1])
