#!/usr/bin/env python
"""The timeline of the LAST tick in a rocprofv3 rocpd database (--kernel-trace): every kernel in start order with the
idle gap in front of it, and the sum of the gaps by the kernel that follows them — where the device stands still inside
a tick (host looks, launch latency) as opposed to where it works.
Usage: python tools/rocpd_gaps.py <results.db> <first kernel of a tick> [min gap us to list, default 8]"""
import collections
import sqlite3
import sys


def short(n):
    return n.split("(")[0].replace("void ", "")


def main(path, first, min_gap=8.0):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, duration from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if short(r[0]) == first]
    seq = rows[idx[-1]:] if idx else rows
    end_prev = None
    gaps = collections.defaultdict(lambda: [0, 0.0])
    busy = 0.0
    t0 = seq[0][1]
    for n, s, d in seq:
        if end_prev is not None:
            gap = (s - end_prev) / 1000.0
            if gap > 0:
                gaps[short(n)][0] += 1
                gaps[short(n)][1] += gap
            if gap >= min_gap:
                print("t=%8.1f us  gap %6.1f us before %s (%.1f us)" % ((s - t0) / 1000.0, gap, short(n), d / 1000.0))
        busy += d / 1000.0
        end_prev = max(end_prev or 0, s + d)
    span = (end_prev - t0) / 1000.0
    print("span %.1f us, kernels %.1f us, idle %.1f us in %d gaps" % (span, busy, sum(v[1] for v in gaps.values()),
                                                                      sum(v[0] for v in gaps.values())))
    for k, (cnt, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1]):
        print("  before %-28s %4d gaps %8.1f us  (%.1f us each)" % (k, cnt, tot, tot / cnt))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 8.0)
