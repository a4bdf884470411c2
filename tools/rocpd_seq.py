#!/usr/bin/env python3
"""The launches of a rocprofv3 rocpd database (kernel-trace) in time order: start (us from the first listed launch), duration,
gap to the previous listed launch's end, grid, kernel.  `--match RE` keeps kernels whose name matches, `--last N` the last N
of them, `--queues` adds the queue / stream columns the database holds."""
import argparse
import re
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--match", default=".")
    ap.add_argument("--last", type=int, default=400)
    ap.add_argument("--first", type=int, default=0)
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    extra = [c for c in ("grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z", "workgroup_x", "workgroup_size_x",
                         "queue_id", "stream_id", "queue", "stream") if c in cols]
    rows = cur.execute(f"select {name_col}, start, end{''.join(', ' + c for c in extra)} from kernels order by start").fetchall()
    rx = re.compile(a.match)
    rows = [r for r in rows if rx.search(r[0])]
    rows = rows[a.first:a.first + a.last] if a.first else rows[-a.last:]
    if not rows:
        print("no launches; columns:", cols)
        return
    t0, prev_end = rows[0][1], None
    print("columns:", extra)
    for r in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", r[0])
        n = re.sub(r"^void\s+", "", n)
        n = re.sub(r"\(.*$", "", n)
        gap = 0.0 if prev_end is None else (r[1] - prev_end) / 1e3
        print(f"{(r[1] - t0) / 1e3:10.1f} {(r[2] - r[1]) / 1e3:9.1f} {gap:8.1f}  {' '.join(str(v) for v in r[3:]):40s} {n[:60]}")
        prev_end = max(prev_end or 0, r[2])


if __name__ == "__main__":
    main()
