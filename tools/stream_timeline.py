#!/usr/bin/env python3
"""Per-stream occupancy of the pipelined bench from a rocprofv3 kernel trace (rocpd database): for the steady-state
steps, the union of kernel intervals per stream (queue), the scale-space stream's idle gaps and what follows them.
usage: python tools/stream_timeline.py results.db [first_step last_step]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns:", cols)
key = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = db.execute(f"select start, end, name, {key or '0'} from kernels order by start").fetchall()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void\s+", "", n)
    return re.sub(r"\(.*$", "", n)[:60]


marks = [i for i, r in enumerate(rows) if "k_level_front2" in r[2] and "unsigned char" in r[2]]
a = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 3
b = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) - 3
t0, t1 = rows[marks[a]][0], rows[marks[b]][0]
steps = b - a
seg = [r for r in rows if r[0] >= t0 and r[0] < t1]
print(f"steps {a}..{b}: {(t1 - t0) / 1e6 / steps:.3f} ms per step, {len(seg) / steps:.0f} launches per step")
by = collections.defaultdict(list)
for s, e, n, q in seg:
    by[q].append((s, e, n))
for q, L in sorted(by.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    L.sort()
    busy = 0; cs, ce = L[0][0], L[0][1]
    gaps = []
    for s, e, n in L[1:]:
        if s > ce:
            gaps.append((s - ce, short(n), ce))
            busy += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    names = collections.Counter(short(n) for _, _, n in L)
    print(f"stream {q}: {len(L) / steps:.0f} launches/step, busy {busy / 1e6 / steps:.2f} ms/step, sum {sum(e - s for s, e, _ in L) / 1e6 / steps:.2f} ms/step; "
          f"top: {', '.join(k for k, _ in names.most_common(3))}")
    g = collections.defaultdict(lambda: [0, 0])
    for d, n, _ in gaps:
        g[n][0] += 1; g[n][1] += d
    for n, (c, d) in sorted(g.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"      idle before {n:60s} x{c / steps:5.1f}/step {d / 1e3 / steps:8.1f} us/step")
