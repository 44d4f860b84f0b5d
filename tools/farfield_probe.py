#!/usr/bin/env python3
"""Queries spread over a box much larger than the mesh: Morton packets as wide as the mesh itself."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd
from sdf_amd import meshes
v, f = meshes.torus(424, 424)
s = sdf_amd.SDF(v, f); s.set_raycast_axes(meshes.random_rotation(1234)); s.set_option("profile", 1)
for kv in sys.argv[3:]:  # key=value handle options
    s.set_option(kv.split("=")[0], int(kv.split("=")[1]))
dev = torch.device("cuda", 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
for half in [float(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "4", "30", "300"])]:
    pts = torch.rand((n, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(7)).mul_(2).sub_(1).mul_(half)
    for name, fn in (("nn", s.nn), ("sdf", s.__call__)):
        fn(pts); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(pts); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"box +-{half:g}: {name:4s} {n/1e6:.0f}M points {dt*1e3:9.2f} ms  heavy {s.stats['last_heavy_points']}", flush=True)
