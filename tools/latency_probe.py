import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from cv_amd import build; build.build()
from cv_amd import akaze
from conftest import synth_frame
img = synth_frame(1920, 1080, 4242, 200, 200)
ak = akaze.Akaze.default()
ctx = ak.context(1920, 1080, 1)
for _ in range(3): ctx.extract_batch([img])
t=time.perf_counter(); N=10
for _ in range(N): r = ctx.extract_batch([img])
print("single 1080p frame extract (host in, host out): %.2f ms, %d keypoints" % ((time.perf_counter()-t)/N*1e3, len(r[0][0])))
