#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (numpy in, numpy out)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd
from sdf_amd import meshes
v, f = meshes.torus(424, 424)
s = sdf_amd.SDF(v, f); s.set_raycast_axes(meshes.random_rotation(1234))
pts = np.random.default_rng(7).uniform(-1, 1, (32_000_000, 3)).astype(np.float32)
import sys as _s
if len(_s.argv) > 1:
    s.set_option("host_chunk", int(_s.argv[1]))
for name, fn in (("sdf", s.__call__), ("contains", s.contains), ("nn", s.nn)):
    fn(pts[:1_000_000])
    fn(pts)
    t0 = time.perf_counter(); fn(pts); dt = time.perf_counter() - t0
    print(f"host API {name:9s} {pts.shape[0]/dt/1e6:8.1f} Mevals/s ({dt*1e3:.0f} ms for {pts.shape[0]} points incl. H2D/D2H)")
