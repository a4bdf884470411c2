import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cv_amd import build; build.build()
from cv_amd import akaze, _lib
from conftest import synth_frame
kw = {}
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    kw[k] = bool(int(v)) if k in ("det_side_stream", "stream_kernels", "parallel_suppression", "pipeline") else int(v)
img = synth_frame(1920, 1080, 4242, 200, 200)
ak = akaze.Akaze.default()
ctx = ak.context(1920, 1080, 1, options=_lib.make_options(**kw))
for _ in range(3): ctx.extract_batch([img])
t=time.perf_counter(); N=20
for _ in range(N): r = ctx.extract_batch([img])
print("single 1080p frame extract (host in, host out) %s: %.3f ms, %d keypoints" % (kw, (time.perf_counter()-t)/N*1e3, len(r[0][0])))
# the C call alone (preallocated numpy buffers, no per-call Python work beyond the ctypes dispatch)
import ctypes as C
cap = ctx.max_kp
kps = np.empty((1, cap), akaze.KP_DTYPE); descs = np.empty((1, cap, 64), np.uint8); cnt = np.zeros(1, np.uint32)
ptrs = (C.c_void_p * 1)(img.ctypes.data)
L = _lib.lib()
for _ in range(3): L.akz_extract_batch(ctx.handle, ptrs, 0, 1, 1920, 1080, 1920, kps.ctypes.data, descs.ctypes.data, cap, cnt.ctypes.data)
t = time.perf_counter()
for _ in range(N): L.akz_extract_batch(ctx.handle, ptrs, 0, 1, 1920, 1080, 1920, kps.ctypes.data, descs.ctypes.data, cap, cnt.ctypes.data)
print("  akz_extract_batch alone (C ABI, host in, host out): %.3f ms, %d keypoints" % ((time.perf_counter() - t) / N * 1e3, int(cnt[0])))
