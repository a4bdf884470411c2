#!/usr/bin/env python3
"""Per-kernel SQ summary from two rocprofv3 PMC passes of tools/phase_profile.py (separate runs, 8 SQ counters each):
  pass A: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY
  pass B: SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE
usage: pmc_sq_summary.py passA.db passB.db   (largest dispatch of each kernel)"""
import re
import sqlite3
import sys


def largest(dbp):
    db = sqlite3.connect(dbp)
    rows = db.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection").fetchall()
    by = {}
    for k, d, c, v in rows:
        k = re.sub(r"\(anonymous namespace\)::", "", k)
        k = re.sub(r"^void\s+", "", k)
        k = re.sub(r"\(.*$", "", k)
        by.setdefault(k, {}).setdefault(d, {})
        by[k][d][c] = by[k][d].get(c, 0) + v
    return {k: max(ds.values(), key=lambda c: max(c.values())) for k, ds in by.items()}


def main(a, b):
    A, B = largest(a), largest(b)
    print(f"{'kernel (largest dispatch)':52s} {'waves':>8s} {'VALU/wave':>9s} {'VALUbusy%':>9s} {'LDS%':>5s} {'+confl%':>7s} "
          f"{'parked%':>7s} {'issue-stall%':>12s} {'active%':>7s}")
    for k in sorted(A, key=lambda k: -A[k].get("SQ_WAVE_CYCLES", 0)):
        if k not in B or "GRBM_GUI_ACTIVE" not in B[k] or not B[k].get("SQ_WAVES"):
            continue
        x = dict(A[k]); x.update(B[k])
        wc, gui, nw = x["SQ_WAVE_CYCLES"], x["GRBM_GUI_ACTIVE"] / 8, x["SQ_WAVES"]
        if gui < 20000:
            continue
        print(f"{k[:52]:52s} {int(nw):8d} {x['SQ_INSTS_VALU'] / nw:9.0f} {100 * x['SQ_INSTS_VALU'] * 4 / 1024 / gui:9.0f} "
              f"{100 * x['SQ_ACTIVE_INST_LDS'] / 256 / gui:5.0f} {100 * x['SQ_LDS_BANK_CONFLICT'] / 256 / gui:7.0f} "
              f"{100 * x['SQ_WAIT_ANY'] / wc:7.0f} {100 * x['SQ_WAIT_INST_ANY'] / wc:12.0f} {100 * x['SQ_ACTIVE_INST_ANY'] / wc:7.0f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
