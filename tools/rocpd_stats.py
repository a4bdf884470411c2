#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a per-kernel stats table
(calls, total ms, avg us, min, max, % of kernel time) — the same content as `--stats` CSV output."""
import re
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", name)
        n = re.sub(r"^void\s+", "", n)
        n = re.sub(r"\(.*$", "", n)
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    lines = [f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{n[:70]:70s} {a[0]:7d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e3:10.2f} {a[2]/1e3:9.2f} {a[3]/1e3:9.2f} {100*a[1]/total:6.2f}")
    lines.append(f"{'TOTAL':70s} {sum(a[0] for a in agg.values()):7d} {total/1e6:10.3f}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
