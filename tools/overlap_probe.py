#!/usr/bin/env python3
"""How much would overlapping the query sort of one half of a batch with the packet kernel of the other half buy?
Two handles on the same mesh (a workspace each), two streams, half the batch each, against one call on the whole."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd
from sdf_amd import meshes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
v, f = meshes.torus(424, 424)
R = meshes.random_rotation(1234)
hs = []
for _ in range(2):
    s = sdf_amd.SDF(v, f); s.set_raycast_axes(R); hs.append(s)
dev = torch.device("cuda", 0)
pts = torch.rand((n, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(7)).mul_(2).sub_(1)
half = (n // 2 + 63) // 64 * 64
a, b = pts[:half].contiguous(), pts[half:].contiguous()
st = [torch.cuda.Stream(device=dev) for _ in range(2)]
def whole():
    hs[0](pts)
def halves_serial():
    hs[0](a); hs[0](b)
def halves_two_streams():
    with torch.cuda.stream(st[0]): hs[0](a)
    with torch.cuda.stream(st[1]): hs[1](b)
for name, fn in (("whole", whole), ("halves, one stream", halves_serial), ("halves, two streams", halves_two_streams)):
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{name:22s} {best * 1e3:8.2f} ms", flush=True)
