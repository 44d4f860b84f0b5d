#!/usr/bin/env python3
"""How much would overlapping one call's sort phase (HBM-bound) with another call's packet kernel
(VALU / latency-bound) buy?  Two handles on the same mesh, each on its own stream, called
alternately, against one handle called back to back."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd
from sdf_amd import meshes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
v, f = meshes.torus(424, 424)
R = meshes.random_rotation(1234)
dev = torch.device("cuda", 0)
hs = []
for k in range(2):
    s = sdf_amd.SDF(v, f, robust=True)
    s.set_raycast_axes(R)
    hs.append(s)
pts = [torch.rand((n, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(7 + k)).mul_(2).sub_(1) for k in range(2)]
out = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(2)]
st = [torch.cuda.Stream(dev) for _ in range(2)]
L = hs[0]._L

def call(k, stream):
    rc = L.sdfhip_sdf_device(hs[k]._h, pts[k].data_ptr(), n, 0, out[k].data_ptr(), stream)
    assert rc == 0

for k in range(2):
    call(k, st[k].cuda_stream)
torch.cuda.synchronize()
reps = 4
t0 = time.perf_counter()
for _ in range(reps):
    call(0, st[0].cuda_stream); call(0, st[0].cuda_stream)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(reps):
    call(0, st[0].cuda_stream); call(1, st[1].cuda_stream)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"one stream : {(t1 - t0) / (2 * reps) * 1e3:.2f} ms per call of {n} points")
print(f"two streams: {(t2 - t1) / (2 * reps) * 1e3:.2f} ms per call")
