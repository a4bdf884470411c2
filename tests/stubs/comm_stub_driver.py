"""Driver of tests/test_comm_stub.py: runs in a subprocess whose LD_LIBRARY_PATH starts with tests/stubs/build, so that
the dlopen("librccl.so.1") of cv_amd/lib/libakz.so finds the stub.  No torch here (PyTorch carries its own librccl.so.1,
which would win).  Prints one JSON object."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
L = C.CDLL(os.path.join(ROOT, "cv_amd", "lib", "libakz.so"))
vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
L.akz_comm_unique_id.argtypes = [vp]
L.akz_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
L.akz_comm_destroy.argtypes = [vp]
L.akz_comm_shift_blocks.argtypes = [vp, vp, vp, u32, u32, vp, vp, vp]
L.akz_comm_allgather_blocks.argtypes = [vp, vp, vp, u32, u32, vp, vp, vp]
L.akz_comm_sync.argtypes = [vp]
L.akz_comm_timing.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), i32]
L.akz_comm_world.argtypes = [vp]
L.akz_comm_last_error_string.restype = C.c_char_p
out = {}
idb = (C.c_uint8 * 128)()
out["unique_id_status"] = L.akz_comm_unique_id(idb)
out["unique_id_prefix"] = bytes(idb[:13]).decode(errors="replace")

if mode == "cpu":
    h = vp()
    out["create_status"] = L.akz_comm_create(idb, 0, 1, 0, C.byref(h))
    print(json.dumps(out))
    sys.exit(0)

stub = C.CDLL("librccl.so.1")      # the copy libakz loaded (same soname: the loader hands back the same object)
stub.rccl_stub_log.argtypes = [vp, i32]
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, i32]
hip.hipMemset.argtypes = [vp, i32, C.c_size_t]
H2D, D2H = 1, 2


def dev(a):
    p = vp()
    assert hip.hipMalloc(C.byref(p), a.nbytes) == 0
    assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, H2D) == 0
    return p


def host(p, shape, dtype):
    a = np.zeros(shape, dtype)
    assert hip.hipMemcpy(a.ctypes.data, p, a.nbytes, D2H) == 0
    return a


def read_log():
    buf = np.zeros((4096, 8), np.int32)
    n = stub.rccl_stub_log(buf.ctypes.data, 4096)
    return buf[:n].tolist()


NF, CAP = 3, 8
res = {}
for world in (1, 2, 3, 4, 8):      # 8 = BASELINE configs[4]: eight communicators in one process
    stub.rccl_stub_reset()
    assert L.akz_comm_unique_id(idb) == 0
    comms = []
    for r in range(world):
        h = vp()
        st = L.akz_comm_create(idb, r, world, 0, C.byref(h))
        assert st == 0, (st, L.akz_comm_last_error_string())
        comms.append(h)
    rng = np.random.default_rng(world)
    descs = [rng.integers(0, 256, (NF, CAP, 64), dtype=np.uint8) for _ in range(world)]
    counts = [rng.integers(0, CAP + 1, (NF,), dtype=np.uint32) for _ in range(world)]
    d_descs = [dev(a) for a in descs]
    d_counts = [dev(a) for a in counts]
    d_rd = [dev(np.full((NF, CAP, 64), 0xEE, np.uint8)) for _ in range(world)]
    d_rc = [dev(np.full((NF,), 0xEEEEEEEE, np.uint32)) for _ in range(world)]
    for r in range(world):
        st = L.akz_comm_shift_blocks(comms[r], d_descs[r], d_counts[r], NF, CAP, d_rd[r], d_rc[r], None)
        assert st == 0, st
    for r in range(world):
        assert L.akz_comm_sync(comms[r]) == 0
    w = {"unfinished_after_shift": stub.rccl_stub_unfinished(), "group_depth": stub.rccl_stub_group_depth()}
    w["shift_ok"] = all(np.array_equal(host(d_rd[r], (NF, CAP, 64), np.uint8), descs[(r - 1) % world]) and
                        np.array_equal(host(d_rc[r], (NF,), np.uint32), counts[(r - 1) % world]) for r in range(world))
    w["shift_log"] = read_log()
    stub.rccl_stub_reset()
    d_ad = [dev(np.full((world, NF, CAP, 64), 0xEE, np.uint8)) for _ in range(world)]
    d_ac = [dev(np.full((world, NF), 0xEEEEEEEE, np.uint32)) for _ in range(world)]
    for r in range(world):
        assert L.akz_comm_allgather_blocks(comms[r], d_descs[r], d_counts[r], NF, CAP, d_ad[r], d_ac[r], None) == 0
    for r in range(world):
        assert L.akz_comm_sync(comms[r]) == 0
    w["unfinished_after_allgather"] = stub.rccl_stub_unfinished()
    w["allgather_ok"] = all(np.array_equal(host(d_ad[r], (world, NF, CAP, 64), np.uint8), np.stack(descs)) and
                            np.array_equal(host(d_ac[r], (world, NF), np.uint32), np.stack(counts)) for r in range(world))
    w["allgather_log"] = read_log()
    w["world_seen"] = [L.akz_comm_world(c) for c in comms]
    if world == 2:
        # a failure inside the group: the second data call of rank 0's shift fails
        stub.rccl_stub_reset()
        stub.rccl_stub_fail_after(1)
        w["fail_status"] = L.akz_comm_shift_blocks(comms[0], d_descs[0], d_counts[0], NF, CAP, d_rd[0], d_rc[0], None)
        w["fail_group_depth"] = stub.rccl_stub_group_depth()
        w["fail_error"] = L.akz_comm_last_error_string().decode()
        w["fail_log_tail"] = read_log()[-2:]
        # ... and the communicator still works afterwards
        stub.rccl_stub_reset()
        for r in range(world):
            assert L.akz_comm_shift_blocks(comms[r], d_descs[r], d_counts[r], NF, CAP, d_rd[r], d_rc[r], None) == 0
        for r in range(world):
            assert L.akz_comm_sync(comms[r]) == 0
        w["after_fail_ok"] = stub.rccl_stub_unfinished() == 0
        # a failing ncclCommInitRank: nothing leaks, the call answers AKZ_E_COMM
        stub.rccl_stub_fail_after(0)
        h = vp()
        w["create_fail_status"] = L.akz_comm_create(idb, 0, 2, 0, C.byref(h))
        w["create_fail_handle"] = h.value
    if world == 1:
        # timing on and never read: the pending event pairs stay bounded (the library resolves the oldest itself)
        ms, calls, nbytes = C.c_double(), C.c_uint64(), C.c_uint64()
        assert L.akz_comm_timing(comms[0], 1, C.byref(ms), C.byref(calls), C.byref(nbytes), 1) == 0
        for _ in range(700):
            assert L.akz_comm_shift_blocks(comms[0], d_descs[0], d_counts[0], NF, CAP, d_rd[0], d_rc[0], None) == 0
        assert L.akz_comm_timing(comms[0], 0, C.byref(ms), C.byref(calls), C.byref(nbytes), 0) == 0
        w["timed_calls"] = calls.value
        w["timed_bytes"] = nbytes.value
        w["timed_ms_positive"] = ms.value > 0.0
    for c in comms:
        assert L.akz_comm_destroy(c) == 0
    res[str(world)] = w
out["worlds"] = res
print(json.dumps(out))
