#!/usr/bin/env python3
"""Per-kernel sums/averages of a PMC counter from a rocprofv3 rocpd sqlite database
(view counters_collection).  usage: rocpd_pmc.py db [out.txt]"""
import re
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    agg = {}
    for name, cname, val in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", name)
        n = re.sub(r"^void\s+", "", n)
        n = re.sub(r"\(.*$", "", n)
        a = agg.setdefault((n, cname), [0, 0.0])
        a[0] += 1
        a[1] += float(val)
    lines = [f"{'kernel':60s} {'counter':18s} {'dispatches':>10s} {'sum':>16s} {'avg_per_dispatch':>18s}"]
    for (n, cname), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{n[:60]:60s} {cname:18s} {a[0]:10d} {a[1]:16.1f} {a[1]/a[0]:18.2f}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
