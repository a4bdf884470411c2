#!/bin/bash
# Runs on the GPU box: builds tools/ubench/fetch_calib.hip and runs it under three rocprofv3 counter passes — FETCH_SIZE, the
# memory-side read requests of the L2 (TCC_EA0_RDREQ) and their size classes (32 / 64 / 128 bytes) — and prints, per access
# pattern, each figure against the bytes the pattern must fetch at 32 / 64 / 128-byte granularity
# -> gpurun_out/fetch_calibration.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=/tmp/fetch_calib_out
rm -rf $O; mkdir -p $O $R/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/ubench/fetch_calib.hip -o /tmp/fetch_calib 2>&1 | grep -v warning | head -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $O/a -o calib -- /tmp/fetch_calib > $O/table.txt 2> $O/err_a.txt
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/b -o calib -- /tmp/fetch_calib > /dev/null 2> $O/err_b.txt
rocprofv3 --pmc TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/c -o calib -- /tmp/fetch_calib > /dev/null 2> $O/err_c.txt
python - "$O" <<'PY' > $R/gpurun_out/fetch_calibration.txt
import sqlite3, sys, glob
O = sys.argv[1]
def counters(sub):
    """[{counter: value}] per dispatch, in dispatch order"""
    f = glob.glob(f"{O}/{sub}/**/calib_results.db", recursive=True)
    if not f:
        return []
    db = sqlite3.connect(f[0])
    rows = db.execute("select dispatch_id, counter_name, value from counters_collection order by dispatch_id").fetchall()
    by = {}
    for d, c, v in rows:
        by.setdefault(d, {})[c] = by.setdefault(d, {}).get(c, 0.0) + float(v)
    return [by[d] for d in sorted(by)]
A, B, Cc = counters("a"), counters("b"), counters("c")
pats = [l.split() for l in open(O + "/table.txt") if l.startswith("k_")]
# the library's dispatches only (hipMemset's fill kernel comes first)
skip = len(A) - len(pats)
print("rocprofv3 counters against known byte counts; every pattern reads a fresh 1 GiB region of an 8 GiB buffer once (tools/ubench/fetch_calib.hip)")
print("RDREQ = TCC_EA0_RDREQ_sum (memory-side read requests of the L2), n32 / n64 / n128 = its size classes; exact = 32 n32 + 64 n64 + 128 n128")
print(f"{'pattern':24s} {'useful MB':>10s} {'at 32 B':>9s} {'at 64 B':>9s} {'at 128 B':>9s} | {'FETCH_SIZE MB':>13s} {'RDREQ M':>9s} {'n32 M':>8s} {'n64 M':>8s} {'n128 M':>8s} {'exact MB':>10s} | {'exact/at32':>10s} {'exact/at64':>10s} {'exact/at128':>11s} {'FETCH/exact':>11s}")
for i, (k, lanes, useful, b32, b64, b128) in enumerate(pats):
    a = A[skip + i] if skip + i < len(A) else {}
    b = B[skip + i] if skip + i < len(B) else {}
    c = Cc[skip + i] if skip + i < len(Cc) else {}
    fs = a.get("FETCH_SIZE", 0.0) * 1024.0
    rd, n32, n64, n128 = b.get("TCC_EA0_RDREQ_sum", 0.0), b.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_64B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    exact = 32 * n32 + 64 * n64 + 128 * n128
    mb = lambda v: float(v) / 1e6
    r = lambda x, y: (x / float(y)) if float(y) else float('nan')
    print(f"{k:24s} {mb(useful):10.1f} {mb(b32):9.1f} {mb(b64):9.1f} {mb(b128):9.1f} | {fs / 1e6:13.1f} {rd / 1e6:9.2f} {n32 / 1e6:8.2f} {n64 / 1e6:8.2f} {n128 / 1e6:8.2f} {exact / 1e6:10.1f} | "
          f"{r(exact, b32):10.3f} {r(exact, b64):10.3f} {r(exact, b128):11.3f} {r(fs, exact):11.3f}")
PY
cat $R/gpurun_out/fetch_calibration.txt
