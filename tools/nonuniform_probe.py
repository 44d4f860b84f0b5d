#!/usr/bin/env python3
"""Non-uniform vertex density: the 424^2 torus plus a copy shrunk to 5 % next to it (the same number of vertices
in 1/400 of the surface).  The automatic choice of the approximate pass looks at the average density only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd
from sdf_amd import meshes
v, f = meshes.torus(424, 424)
sc = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
v2 = (v * np.float32(sc) + np.float32([0.5, 0.0, 0.6])).astype(np.float32)
V = np.concatenate([v, v2]); F = np.concatenate([f, f + v.shape[0]]).astype(np.uint32)
dev = torch.device("cuda", 0)
pts = torch.rand((32_000_000, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(7)).mul_(2).sub_(1)
for approx in (-1, 0, 1):
    s = sdf_amd.SDF(V, F, robust=False)
    s.set_option("approx", approx); s.set_option("profile", 1); s.update()
    s.nn(pts); torch.cuda.synchronize(); t0 = time.perf_counter(); s.nn(pts); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = s.stats
    print(f"approx {approx:2d}: nn 32M points {dt*1e3:8.2f} ms  heavy {st['last_heavy_points']} kernel {st['last_kernel']}", flush=True)
