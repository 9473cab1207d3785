#!/usr/bin/env python
"""Sum a rocprofv3 --pmc counter per kernel from a rocpd SQLite database.
Usage: python tools/rocpd_pmc.py <db> [counter-name-substring]"""
import sqlite3
import sys


def main(path, want=None):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info('pmc_events')")]
    sys.stderr.write("pmc_events columns: %s\n" % cols)
    rows = c.execute("select * from pmc_events limit 1").fetchall()
    sys.stderr.write("sample: %s\n" % (rows,))
    name_col = "counter_name" if "counter_name" in cols else "pmc_name" if "pmc_name" in cols else None
    val_col = "value" if "value" in cols else "counter_value"
    kname = "name" if "name" in cols else "kernel_name" if "kernel_name" in cols else None
    q = "select %s, %s, count(*), sum(%s) from pmc_events group by 1, 2 order by 4 desc" % (kname, name_col, val_col)
    print("kernel,counter,dispatches,sum")
    for r in c.execute(q):
        if want and want not in str(r[1]):
            continue
        print("%s,%s,%d,%s" % (str(r[0]).split("(")[0], r[1], r[2], r[3]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
