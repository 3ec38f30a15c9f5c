#!/usr/bin/env python3
"""Per-kernel averages of whatever counters one or more rocprofv3 --pmc passes collected, for the kernels whose name matches a
regular expression (tools/prefill_pmc.sh).  Counters are summed over the chip by rocprofv3 (SQ_*: over the SIMDs / CUs,
GRBM_GUI_ACTIVE: over the 8 XCDs); the derived lines say what they divide by.
usage: python tools/pmc_kernel_summary.py '<regex>' <results.db> [<results2.db> ...]"""
import re
import sqlite3
import sys

SIMDS, XCDS = 1024, 8


def main():
    pat = re.compile(sys.argv[1])
    rows = {}
    for db in sys.argv[2:]:
        c = sqlite3.connect(db)
        q = ("SELECT kernel_name, counter_name, COUNT(*), AVG(value), AVG(duration) FROM counters_collection "
             "GROUP BY kernel_name, counter_name")
        for name, counter, n, v, d in c.execute(q):
            if not pat.search(name):
                continue
            r = rows.setdefault(name, dict(calls=n, us=d / 1e3))
            r[counter] = v
    print('per-launch averages; durations are those under the counter pass')
    for name, r in sorted(rows.items(), key=lambda kv: -kv[1]['us']):
        print('\n' + name[:160])
        print(f"  launches {r['calls']}, avg duration {r['us']:.2f} us")
        for k in sorted(r):
            if k not in ('calls', 'us'):
                print(f'  {k:32s} {r[k]:.6g}')
        gui = r.get('GRBM_GUI_ACTIVE')
        if gui:
            cyc = gui / XCDS
            if r.get('SQ_VALU_MFMA_BUSY_CYCLES'):
                print(f"  -> matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / {XCDS} x {SIMDS} SIMDs) = "
                      f"{r['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * SIMDS):.3f}")
        wc = r.get('SQ_WAVE_CYCLES')
        if wc:
            for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS',
                      'SQ_ACTIVE_INST_VMEM'):
                if r.get(k) is not None:
                    print(f"  -> {k} / SQ_WAVE_CYCLES = {r[k] / wc:.3f}")


if __name__ == '__main__':
    main()
