"""Builds cv_amd/lib/libakz.so (the C-ABI library of include/akz.h) with hipcc for gfx950.

The flags are part of the parity contract: -ffp-contract=off (rustc never contracts to FMA),
no fast-math, IEEE-correct f32 divide/sqrt (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libakz.so")
SOURCES = ["akz_api.hip", "akz_scale_space.hip", "akz_arith.hip", "akz_keypoints.hip", "hm_match.hip", "rs_ransac.hip", "akz_color.hip",
           "akz_comm.hip", "akz_plan.cpp"]
# akz_scale_space.hip is compiled once per combination of the reference's three un-vendored arithmetic orders
# (-DAKZ_ARITH=k, include/akz.h AKZ_ARITH_*); akz_arith.hip routes a context to its copy
ARITH_VARIANTS = {"akz_scale_space.hip": range(8)}
HEADERS = ["akz_common.h", "akz_ctx.h", "../../include/akz.h", "../../include/akz_portable_math.h", "../../include/akz_ransac_math.h", "../../include/akz_p3p_math.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    # an object that is missing or newer than the library (an interrupted or partial build): link again.  (On the GPU box
    # the objects do not travel — only when SOME are present is their state a statement about the library.)
    objs = [os.path.join(HERE, "lib", f) for f in os.listdir(os.path.join(HERE, "lib")) if f.endswith(".o")]
    return any(os.path.getmtime(o) > t for o in objs)


def _object_is_current(obj, cmd):
    """An object is reused when it is newer than every file its compiler listed as a dependency (the -MD file beside it) and
    was produced by the same command line."""
    dep, tag = obj + ".d", obj + ".cmd"
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(tag)):
        return False
    with open(tag) as f:
        if f.read() != " ".join(cmd):
            return False
    t = os.path.getmtime(obj)
    with open(dep) as f:
        text = f.read().replace("\\\n", " ")
    files = [w for part in text.split(":", 1)[1:] for w in part.split() if not w.endswith(":")]
    return bool(files) and all(os.path.exists(w) and os.path.getmtime(w) <= t for w in files)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objs = []
    jobs = []
    for s in SOURCES:
        for k in ARITH_VARIANTS.get(s, [None]):
            tag = "" if k is None else f"_a{k}"
            o = os.path.join(HERE, "lib", s.replace(".", "_") + tag + ".o")
            extra = os.environ.get("AKZ_EXTRA_FLAGS", "").split()   # experiments only (e.g. -DKNN_ABLATE=1)
            extra += [] if k is None else [f"-DAKZ_ARITH={k}"]
            cmd = [hipcc()] + FLAGS + extra + (["-x", "hip"] if s.endswith(".hip") else []) + ["-MD", "-MF", o + ".d", "-c", os.path.join(CSRC, s), "-o", o]
            objs.append(o)
            if not force and _object_is_current(o, cmd):
                continue
            if os.path.exists(o + ".cmd"):
                os.remove(o + ".cmd")
            jobs.append((s + tag, cmd, o))
    # (at most as many compilers at once as the host has cores: the eight copies of the largest file would otherwise all
    # start together on a small box)
    failed = False
    limit = max(2, os.cpu_count() or 2)
    for j0 in range(0, len(jobs), limit):
        procs = []
        for name, cmd, _ in jobs[j0:j0 + limit]:
            if verbose:
                print(" ".join(cmd))
            procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        failed = _wait(procs, verbose) or failed
    if not failed:
        for _, cmd, o in jobs:
            with open(o + ".cmd", "w") as f:
                f.write(" ".join(cmd))
    if failed:
        raise RuntimeError("libakz build failed")
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    subprocess.check_call(cmd)
    return LIB


def _wait(procs, verbose):
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if out and (verbose or p.returncode != 0):
            sys.stderr.write(out.decode(errors="replace"))
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"hipcc failed on {s}\n")
    return failed


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
