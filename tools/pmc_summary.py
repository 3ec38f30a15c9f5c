#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on
gfx950: TCC has 4 slots, they cost 3 + 2), corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes:
rocprofv3 reports KiB, and on gfx950 FETCH_SIZE counts a 128-byte request of a 16 B/lane streaming read as 64 B -> x2.
WRITE_SIZE is uncalibrated (reported as is; the decode kernels write KBs against MBs read).

usage: python tools/pmc_summary.py <fetch.db> <write.db> <config-name> <out.json> [<stats.txt>]
"""
import json
import os
import sqlite3
import sys

FETCH_CORRECTION = 2.0  # gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B while the requests are 128 B


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    q = ("SELECT kernel_name, COUNT(*), AVG(value), AVG(duration) FROM counters_collection "
         "WHERE counter_name=? GROUP BY kernel_name")
    return {name: dict(calls=n, kib=v, avg_us=d / 1e3) for name, n, v, d in c.execute(q, (counter,))}


def main():
    fetch_db, write_db, config, out = sys.argv[1:5]
    f = per_kernel(fetch_db, 'FETCH_SIZE')
    w = per_kernel(write_db, 'WRITE_SIZE')
    rows = []
    for name, r in f.items():
        if 'tllm::kernels' not in name:
            continue
        wr = w.get(name, dict(kib=0.0))
        rows.append(dict(kernel=name, calls=r['calls'], fetch_bytes=r['kib'] * 1024 * FETCH_CORRECTION,
                         write_bytes=wr['kib'] * 1024, avg_us_under_pmc=r['avg_us']))
    rows.sort(key=lambda r: -r['fetch_bytes'] * r['calls'])
    gem = [r for r in rows if 'gemv_kernel' in r['kernel'] or 'gemv_ksplit_kernel' in r['kernel']]
    layer_calls = sorted(r['calls'] for r in gem)[len(gem) // 2]  # the median: robust against the shared instance
    # per-layer kernels run once per layer and step; a template instance shared with the head GEMV (fp16 QKV) has a few more calls
    layer = [r for r in gem if r['calls'] >= 0.9 * layer_calls]
    tot = sum((r['fetch_bytes'] + r['write_bytes']) * r['calls'] for r in layer)
    summary = dict(
        counters='FETCH_SIZE (x2 gfx950 correction, KiB -> bytes) and WRITE_SIZE (KiB -> bytes), separate passes',
        gemv_layer_hbm_bytes_per_launch=tot / sum(r['calls'] for r in layer),
        gemv_layer_kernels=len(layer),
        # the dominant kernel (RMSNorm -> gate|up GEMV -> SwiGLU) is the layer GEMV that fetches most
        gate_up_hbm_bytes_per_launch=max(r['fetch_bytes'] + r['write_bytes'] for r in layer), kernels=rows)
    data = {}
    if os.path.exists(out):
        data = json.load(open(out))
    data[config] = summary
    json.dump(data, open(out, 'w'), indent=1)
    print("%10s %14s %12s  %s" % ("calls", "fetch_MB(x2)", "write_KB", "kernel"))
    for r in rows:
        print("%10d %14.3f %12.1f  %s" % (r['calls'], r['fetch_bytes'] / 1e6, r['write_bytes'] / 1e3, r['kernel'][:160]))
    print("gemv_layer HBM bytes per launch (mean over %d kernels): %.0f" % (len(layer), summary['gemv_layer_hbm_bytes_per_launch']))


if __name__ == '__main__':
    main()
