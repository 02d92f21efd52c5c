#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a per-kernel stats table."""
import collections
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    ks = {r[0]: r[1] for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
    d = collections.defaultdict(list)
    for k, s, e, gx, wx in cur.execute("select kernel_id, start, end, grid_size_x, workgroup_size_x from rocpd_kernel_dispatch"):
        d[ks.get(k, str(k))].append(((e - s) / 1e3, gx, wx))
    total = sum(x[0] for v in d.values() for x in v)
    print("%-64s %7s %13s %11s %11s %11s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for k, v in sorted(d.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        t = [x[0] for x in v]
        print("%-64s %7d %13.1f %11.1f %11.1f %11.1f %6.2f" % (k[:64], len(t), sum(t), sum(t) / len(t), min(t), max(t), 100 * sum(t) / total))


if __name__ == "__main__":
    main(sys.argv[1])
