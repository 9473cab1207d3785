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
    for k in names:                      # a name also stands for its template instances ("void kp_round<2048u>")
        v = per.get(k, [])
        if not v:
            for full in sorted(per):
                if full.replace("void ", "").split("<")[0] == k:
                    print("%s,%d,sum_us=%.0f,avg_us=%.1f" % (full.replace("void ", ""), len(per[full]), sum(per[full]),
                                                              sum(per[full]) / len(per[full])))
            merged = [(s, n.split("(")[0], d) for n, s, d in seq if n.split("(")[0].replace("void ", "").split("<")[0] == k]
            v = [d / 1000.0 for s, n, d in sorted(merged)]
        print("%s,%d,sum_us=%.0f,first10=%s,every10th=%s" % (k, len(v), sum(v), [round(x) for x in v[:10]],
                                                          [round(x) for x in v[10::10]]))
    t0 = seq[0][1]
    tend = max(s + d for n, s, d in seq)
    print("span_ms,%.3f,kernels_ms,%.3f" % ((tend - t0) / 1e6, sum(d for n, s, d in seq) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
