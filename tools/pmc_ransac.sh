#!/bin/bash
# Runs on the GPU box: VALU instruction counts and kernel durations of BASELINE configs[3] (10 000 eight-point hypotheses x
# 4 poses x 1 000 matches, exhaustive scoring; every VALU instruction of these kernels is f64 arithmetic or its address /
# control overhead) -> gpurun_out/pmc_ransac.json (copy to profiles/rNN_pmc_ransac.json: bench.py reads it for
# configs_extra["configs[3]"].roofline).   usage: bash tools/pmc_ransac.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r04}
O=/tmp/pmc_ransac
CALLS=3
rm -rf $O; mkdir -p $O $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -d $O/pmc -o r -- python $R/tools/ransac_once.py $CALLS > $O/pmc.log 2>&1
rocprofv3 --kernel-trace -d $O/trace -o r -- python $R/tools/ransac_once.py $CALLS > $O/trace.log 2>&1
python - "$O" "$CALLS" "$T" <<'PY' > $R/gpurun_out/pmc_ransac.json
import glob, json, re, sqlite3, sys
O, calls, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
def clean(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name); n = re.sub(r"^void\s+", "", n); return re.sub(r"\(.*$", "", n)
db = sqlite3.connect(glob.glob(O + "/pmc/**/r_results.db", recursive=True)[0])
ins, waves = {}, {}
for name, cname, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
    d = ins if cname == "SQ_INSTS_VALU" else waves if cname == "SQ_WAVES" else None
    if d is not None:
        a = d.setdefault(clean(name), [0, 0.0]); a[0] += 1; a[1] += float(val)
db = sqlite3.connect(glob.glob(O + "/trace/**/r_results.db", recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
nm = "name" if "name" in cols else "kernel_name"
dur = {}
for name, s, e in db.execute(f"select {nm}, start, end from {view}"):
    a = dur.setdefault(clean(name), [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
out = {"tag": tag, "calls": calls, "workload": "BASELINE configs[3]: rs_essential_batch, 10 000 eight-point hypotheses x 4 poses x 1 000 matches (30 % outliers), threshold 1e-7, exhaustive scoring",
       "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES and rocprofv3 --kernel-trace (separate passes) of tools/ransac_once.py", "kernels": {}}
for k in sorted(ins, key=lambda k: -ins[k][1]):
    if not k.startswith("k_"):
        continue
    out["kernels"][k] = {"launches_per_call": ins[k][0] / calls, "valu_insts_per_call": ins[k][1] / calls,
                         "waves_per_call": waves.get(k, [0, 0.0])[1] / calls, "kernel_us_per_call": dur.get(k, [0, 0.0])[1] / calls}
out["valu_insts_per_call"] = sum(v["valu_insts_per_call"] for v in out["kernels"].values())
out["kernel_us_per_call"] = sum(v["kernel_us_per_call"] for v in out["kernels"].values())
print(json.dumps(out, indent=1))
PY
head -c 1800 $R/gpurun_out/pmc_ransac.json
