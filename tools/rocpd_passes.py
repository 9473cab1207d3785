#!/usr/bin/env python
"""Per-launch durations (us) of the named kernels over the LAST tick in a rocprofv3 rocpd
database: shows how a multi-launch walk's cost moves from pass to pass.
Usage: python tools/rocpd_passes.py <results.db> <first kernel of a tick> <kernel> [<kernel> ...]"""
import collections
import sqlite3
import sys


def main(path, first, names):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, duration from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if r[0].split("(")[0] == first]
    seq = rows[idx[-1]:] if idx else rows
    per = collections.defaultdict(list)
    for n, s, d in seq:
        per[n.split("(")[0]].append(d / 1000.0)
    for k in names:
        v = per.get(k, [])
        print("%s,%d,sum_us=%.0f,first10=%s,every10th=%s" % (k, len(v), sum(v), [round(x) for x in v[:10]],
                                                          [round(x) for x in v[10::10]]))
    t0 = seq[0][1]
    tend = max(s + d for n, s, d in seq)
    print("span_ms,%.3f,kernels_ms,%.3f" % ((tend - t0) / 1e6, sum(d for n, s, d in seq) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
