#!/usr/bin/env python3
"""HBM bandwidth probes on the GPU box: pure write (fill), pure read (reduction), copy (read + write), and mixes
with the read:write ratios of the scale-space kernels.  Prints GB/s of bytes moved."""
import torch
dev = torch.device("cuda", 0)
n = 1 << 28                       # 1 GiB of f32
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
c = torch.empty(n, dtype=torch.float32, device=dev)


def timed(fn, nbytes, reps=5):
    best = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = max(best, nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return round(best, 1)


a.fill_(1.0); b.fill_(2.0)
print("write only (fill)        :", timed(lambda: a.fill_(3.0), 4 * n), "GB/s")
print("read only (sum)          :", timed(lambda: a.sum(), 4 * n), "GB/s")
print("copy (1 read : 1 write)  :", timed(lambda: b.copy_(a), 8 * n), "GB/s")
print("add  (2 reads : 1 write) :", timed(lambda: torch.add(a, b, out=c), 12 * n), "GB/s")
# 1 read : 3 writes (the level front-end's ratio): one f32 in, three f32 planes out
x = torch.empty(n // 4, dtype=torch.float32, device=dev)
o = torch.empty((3, n // 4), dtype=torch.float32, device=dev)
print("1 read : 3 writes        :", timed(lambda: torch.mul(x.unsqueeze(0), 2.0, out=o), 16 * (n // 4)), "GB/s")
