"""akz_comm_* (the exchange step of the frame-sharded front-end) against a stand-in for librccl.so.1.

No N > 1 RCCL transfer has ever run on hardware the builder could reach, so the send / receive argument layout of the ring
shift and the all-gather is held here: tests/stubs/rccl_stub.cpp implements the nine symbols the library resolves, logs
every call and moves the data between the communicators of ONE process, which then plays every rank of a world of 1, 2,
3, 4 and 8 (BASELINE configs[4]: eight communicators, eight streams) on one GPU.  The CPU half (no device) checks that the dlopen / dlsym path reaches a librccl.so.1 at all."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_DIR = os.path.join(ROOT, "tests", "stubs", "build")
STUB = os.path.join(STUB_DIR, "librccl.so.1")
SRC = os.path.join(ROOT, "tests", "stubs", "rccl_stub.cpp")
U8, U32 = 1, 3   # ncclUint8, ncclUint32


def build_stub():
    if os.path.exists(STUB) and os.path.getmtime(STUB) >= os.path.getmtime(SRC):
        return
    os.makedirs(STUB_DIR, exist_ok=True)
    hipcc = "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "-O2", "-fPIC", "-shared", "-std=c++17", "-x", "hip", "--offload-arch=gfx950",
                           "-Wno-unused-result", "-Wno-unused-value", SRC, "-o", STUB])


def run_driver(mode):
    from cv_amd import build
    build.build()
    build_stub()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = STUB_DIR + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stubs", "comm_stub_driver.py"), mode], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_library_reaches_a_librccl_through_dlopen():
    out = run_driver("cpu")
    assert out["unique_id_status"] == 0
    assert out["unique_id_prefix"] == "akz-rccl-stub"          # the id came from the stub, i.e. through dlopen + dlsym
    assert out["create_status"] in (0, -2)                        # AKZ_E_NO_DEVICE on a box without a GPU


def _data_calls(log):
    return [r for r in log if r[0] in (0, 1, 2)]


@pytest.mark.gpu
def test_ring_shift_and_all_gather_for_world_1_to_8():
    out = run_driver("gpu")
    NF, CAP = 3, 8
    assert sorted(int(k) for k in out["worlds"]) == [1, 2, 3, 4, 8]
    for world in (1, 2, 3, 4, 8):
        w = out["worlds"][str(world)]
        assert w["shift_ok"] and w["allgather_ok"], world          # the bytes arrived where the matcher reads them
        assert w["unfinished_after_shift"] == 0 and w["unfinished_after_allgather"] == 0
        assert w["group_depth"] == 0
        assert w["world_seen"] == [world] * world
        # the ring shift: per rank ONE group of send(descs), send(counts) to rank + 1 and recv, recv from rank - 1
        log = w["shift_log"]
        for r in range(world):
            mine = [x for x in _data_calls(log) if x[1] == r]
            nxt, prv = (r + 1) % world, (r - 1) % world
            assert [(x[0], x[2], x[3], x[6]) for x in mine] == [
                (0, nxt, U8, NF * CAP * 64), (0, nxt, U32, NF), (1, prv, U8, NF * CAP * 64), (1, prv, U32, NF)], (world, r, mine)
            assert all(x[4] == 1 for x in mine)                   # inside a group
        assert sum(1 for x in log if x[0] == 3) == world and sum(1 for x in log if x[0] == 4) == world
        # the all-gather: per rank one group of two collectives, element counts per RANK (not per world)
        log = w["allgather_log"]
        for r in range(world):
            mine = [x for x in _data_calls(log) if x[1] == r]
            assert [(x[0], x[3], x[6]) for x in mine] == [(2, U8, NF * CAP * 64), (2, U32, NF)]
            assert all(x[4] == 1 for x in mine)
    w2 = out["worlds"]["2"]
    assert w2["fail_status"] == -8 and w2["fail_group_depth"] == 0          # AKZ_E_COMM, and the group was closed
    assert w2["fail_log_tail"][-1][0] == 4
    assert "injected" in w2["fail_error"]
    assert w2["after_fail_ok"]
    assert w2["create_fail_status"] == -8 and not w2["create_fail_handle"]
    w1 = out["worlds"]["1"]
    assert w1["timed_calls"] == 700 and w1["timed_bytes"] == 700 * (NF * CAP * 64 + 4 * NF) and w1["timed_ms_positive"]
