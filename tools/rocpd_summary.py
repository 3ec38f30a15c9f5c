#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (what `rocprofv3 --kernel-trace --stats` writes in this image: *_results.db) into
the per-kernel stats table kept under profiles/.

usage: python tools/rocpd_summary.py gpurun_out/<dir>/<x>_results.db [> profiles/rNN_<what>.stats.txt]
"""
import sqlite3
import sys


def summarise(path, top=40):
    c = sqlite3.connect(path)
    rows = c.execute(
        "SELECT S.display_name, COUNT(*), SUM(K.end-K.start), MIN(K.end-K.start), MAX(K.end-K.start) "
        "FROM rocpd_kernel_dispatch K JOIN rocpd_info_kernel_symbol S ON S.id=K.kernel_id AND S.guid=K.guid "
        "GROUP BY S.display_name ORDER BY 3 DESC").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = ["%8s %10s %10s %10s %7s  %s" % ("calls", "avg_us", "min_us", "max_us", "pct", "kernel")]
    for name, n, tot, mn, mx in rows[:top]:
        out.append("%8d %10.2f %10.2f %10.2f %7.2f  %s" % (n, tot / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total,
                                                            name[:200]))
    out.append("total kernel time: %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))
    return "\n".join(out)


if __name__ == "__main__":
    print(summarise(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40))
