import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select start, end, name from kernels order by start").fetchall()
# last frame: take the last N kernels between the final two 'k_level_front2<4' launches
idx = [i for i, r in enumerate(rows) if "k_level_front2" in r[2] and "Lh" in r[2] or ("unsigned char" in r[2] and "front2" in r[2])]
print("front0 launches:", len(idx))
a, b = idx[-2], idx[-1]
seg = rows[a:b]
busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
for s, e, _ in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("kernels per frame:", len(seg), "wall(us) between frame starts:", (rows[b][0] - rows[a][0]) / 1e3, "busy union(us):", busy / 1e3, "sum(us):", sum(e - s for s, e, _ in seg) / 1e3)
import re, collections
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in seg:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void\s+", "", n); n = re.sub(r"\(.*$", "", n)
    agg[n][0] += 1; agg[n][1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{n[:60]:60s} {c:3d} {t/1e3:9.1f} us")
# full timeline of the frame: offset of the start, duration, gap to the latest end seen so far
print("--- timeline (us): start, dur, gap_since_prev_end, name")
t0 = seg[0][0]; last_end = seg[0][0]
for s_, e_, n_ in seg:
    n_ = re.sub(r"\(anonymous namespace\)::", "", n_); n_ = re.sub(r"^void\s+", "", n_); n_ = re.sub(r"\(.*$", "", n_)
    print(f"{(s_-t0)/1e3:9.1f} {(e_-s_)/1e3:7.1f} {(s_-last_end)/1e3:7.1f}  {n_[:70]}")
    last_end = max(last_end, e_)
