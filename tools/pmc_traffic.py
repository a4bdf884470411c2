#!/usr/bin/env python3
"""Build profiles/r01_pmc_traffic.json from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs
of `bench.py --frames 64 --micro-batch 64 --steps 1 --warmup 0 --no-cpu-baseline` with AKZ_PIPELINE=0).
usage: pmc_traffic.py fetch.db write.db out.json [frames_per_launch]"""
import json
import re
import sqlite3
import sys

FED_BYTES_PER_PIXEL_STEP = 12


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
    agg = {}
    for name, val in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", name)
        n = re.sub(r"^void\s+", "", n)
        n = re.sub(r"\(.*$", "", n)
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += float(val)
    return agg


def main(fetch_db, write_db, out, frames=64):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --frames 64 "
                     "--micro-batch 64 --steps 1 --warmup 0, AKZ_PIPELINE=0; units KiB; FETCH_SIZE doubled (gfx950 "
                     "16-byte-read correction, MI355X_MICROARCH.md)"}
    fed_f = sum(v[1] for k, v in f.items() if k.startswith("k_fed_"))
    fed_w = sum(v[1] for k, v in w.items() if k.startswith("k_fed_"))
    fed_n = sum(v[0] for k, v in f.items() if k.startswith("k_fed_"))
    res["fed"] = {"launches": fed_n, "fetch_size_kib": round(2 * fed_f, 1), "write_size_kib": round(fed_w, 1),
                  "hbm_bytes_per_launch": round((2 * fed_f + fed_w) * 1024 / max(1, fed_n)),
                  "frames_per_launch": int(frames)}
    for k in sorted(f, key=lambda k: -f[k][1])[:12]:
        if k.startswith("k_fed_"):
            continue
        res[k] = {"dispatches": f[k][0], "fetch_size_kib": round(2 * f[k][1], 1),
                  "write_size_kib": round(w.get(k, [0, 0.0])[1], 1)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["fed"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else 64)
