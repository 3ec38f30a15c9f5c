#!/usr/bin/env python3
"""Per-kernel matrix-pipe counters of the prefill GEMMs from a rocprofv3 --pmc pass (tools/mfma_pmc.sh):
SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_I8, SQ_WAIT_INST_LDS, SQ_LDS_BANK_CONFLICT, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES,
GRBM_GUI_ACTIVE - averaged per launch, with the derived figures BASELINE.json's north_star asks for:
  achieved clock      = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration     (rocprofv3 sums the counter over the 8 XCDs)
  MFMA utilisation    = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)     (busy cycles are summed over the SIMDs)
  int8 MFMA ops check = SQ_INSTS_VALU_MFMA_MOPS_I8 x 512 ops vs 2 M N K
usage: python tools/mfma_pmc_summary.py <results.db> [<results2.db> ...]"""
import sqlite3
import sys

SIMDS = 1024
XCDS = 8


def main():
    rows = {}
    for db in sys.argv[1:]:
        c = sqlite3.connect(db)
        q = "SELECT kernel_name, counter_name, COUNT(*), AVG(value), AVG(duration) FROM counters_collection GROUP BY kernel_name, counter_name"
        for name, counter, n, v, d in c.execute(q):
            if 'gemm' not in name:
                continue
            r = rows.setdefault(name, dict(calls=n, us=d / 1e3))
            r[counter] = v
    print('counters are per-launch averages; durations are under the counter pass (a few % slower than unprofiled)')
    for name, r in sorted(rows.items(), key=lambda kv: -kv[1]['us']):
        print('\n' + name[:200])
        print(f"  launches {r['calls']}, avg duration {r['us']:.2f} us")
        for k in sorted(r):
            if k not in ('calls', 'us'):
                print(f'  {k:32s} {r[k]:.6g}')
        gui = r.get('GRBM_GUI_ACTIVE')
        if gui:
            gui = gui / XCDS
            # not the shader clock: the in-kernel measurement (s_memtime against the 100 MHz counter, tools/gemm_sweep.py CLOCKS=1,
            # profiles/r02_sqgemm_ablation.txt) reads 0.2 - 0.4 GHz lower under the same kernels
            print(f"  -> GRBM_GUI_ACTIVE / {XCDS} XCDs / duration = {gui / r['us'] / 1e3:.3f} GHz (GRBM-domain cycles; the shader clock "
                  f"held is measured in-kernel, see profiles/r02_sqgemm_ablation.txt)")
            busy = r.get('SQ_VALU_MFMA_BUSY_CYCLES')
            if busy:
                print(f'  -> MFMA busy = MFMA_BUSY_CYCLES / (GUI_ACTIVE / {XCDS} x {SIMDS} SIMDs) = {busy / (gui * SIMDS):.3f} of the GRBM cycles')
        mops = r.get('SQ_INSTS_VALU_MFMA_MOPS_I8')
        if mops:
            print(f'  -> int8 MFMA ops = MOPS_I8 x 512 = {mops * 512:.4g}')
        mops = r.get('SQ_INSTS_VALU_MFMA_MOPS_F16')
        if mops:
            print(f'  -> fp16 MFMA flops = MOPS_F16 x 512 = {mops * 512:.4g}')


if __name__ == '__main__':
    main()
