#!/usr/bin/env python3
"""What physically sorting the points would cost: a gather of 125M 12-byte points through a
random permutation into 16-byte records, as a last radix pass writing points instead of indices would do it."""
import time, torch
dev = torch.device("cuda", 0)
n = 125_000_000
pts = torch.rand((n, 3), device=dev)
perm = torch.randperm(n, device=dev)
out = torch.empty((n, 4), device=dev)
def run():
    out[:, :3] = pts[perm]
for f, name in ((run, "gather 12 B -> 16 B records (index_select + strided write)"), (lambda: torch.index_select(pts, 0, perm), "index_select only")):
    f(); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{name}: {best*1e3:.2f} ms")
