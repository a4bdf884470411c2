#!/usr/bin/env python3
"""Properties of the COMPILED kernels that the source cannot promise (the compiler's choice of form moves some kernels by
integer factors): disassembles the gfx950 code object inside a built cv_amd/lib/*.o and checks

  k_rsb_score_p3p   the exact statement of rs_w2c_inlier (the only v_rsq_f64 of the kernel) sits behind an exec-mask branch
                    inside the match loop — not flattened into selects — and the loop is unrolled twice (two groups of
                    ds_read_b128 in its body).  Flattened: 3.7 times the kernel's time; not unrolled: +10 to +25 %.

`python tools/check_isa.py` prints one line per check and exits 1 on a failure (tests/test_abi.py runs it)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    """{kernel symbol: [instruction lines]} of the gfx950 code object bundled in a host object file."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(d, "copy.o")])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        text = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = out.setdefault(m.group(1), [])
        elif cur is not None and line.strip():
            cur.append(line.strip())
    return out


def check_score_p3p(kernels):
    name = [k for k in kernels if "k_rsb_score_p3p" in k]
    if len(name) != 1:
        return False, f"k_rsb_score_p3p: {len(name)} symbols"
    ins = kernels[name[0]]
    ops = [l.split()[0] for l in ins]
    rsq = [i for i, o in enumerate(ops) if o == "v_rsq_f64_e32"]
    if not rsq:
        return False, "k_rsb_score_p3p: no v_rsq_f64 (the exact statement is gone?)"
    ok = True
    for i in rsq:
        # the nearest preceding LDS read of a match starts the residual; a branch on exec must lie between it and the root
        j = max((k for k in range(i) if ops[k].startswith("ds_read_b128")), default=None)
        if j is None or not any(o in ("s_cbranch_execz", "s_cbranch_execnz") for o in ops[j:i]):
            ok = False
    # back edge of the match loop: the last s_cbranch before the epilogue that jumps backwards; its body holds the reads
    n_groups = sum(1 for k in range(1, len(ops)) if ops[k].startswith("ds_read_b128") and not ops[k - 1].startswith("ds_read")
                   and not (k >= 2 and ops[k - 2].startswith("ds_read")))
    detail = f"k_rsb_score_p3p: {len(rsq)} copies of the exact statement, {'all' if ok else 'NOT all'} behind an exec branch; {n_groups} match-read groups"
    return ok and n_groups >= 3, detail      # (two matches in the unrolled body + the remainder iteration)


def main():
    obj = os.path.join(ROOT, "cv_amd", "lib", "rs_ransac_hip.o")
    kernels = disassemble(obj)
    good = True
    for chk in (check_score_p3p,):
        ok, msg = chk(kernels)
        print(("ok   " if ok else "FAIL ") + msg)
        good = good and ok
    return 0 if good else 1


if __name__ == "__main__":
    sys.exit(main())
