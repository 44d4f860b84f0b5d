#!/usr/bin/env python3
"""Host entry points through the raw C ABI with a pre-touched output buffer (no Python allocation in the timed region)."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd
from sdf_amd import meshes
v, f = meshes.torus(424, 424)
s = sdf_amd.SDF(v, f); s.set_raycast_axes(meshes.random_rotation(1234))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
pts = np.random.default_rng(7).uniform(-1, 1, (n, 3)).astype(np.float32)
out = np.zeros(n, dtype=np.float32)
L = s._L
for chunk in (0, 4_000_000, 8_000_000, 16_000_000, 32_000_000):
    s.set_option("host_chunk", chunk)
    for _ in range(2):
        t0 = time.perf_counter()
        rc = L.sdfhip_sdf(s._h, pts.ctypes.data_as(C.POINTER(C.c_float)), n, 0, out.ctypes.data_as(C.POINTER(C.c_float)))
        dt = time.perf_counter() - t0
    assert rc == 0
    print(f"raw sdfhip_sdf host_chunk={chunk:9d}: {n/dt/1e6:8.1f} Mevals/s ({dt*1e3:.1f} ms)")
t0 = time.perf_counter(); r = s(pts); dt = time.perf_counter() - t0
print(f"python wrapper: {n/dt/1e6:8.1f} Mevals/s ({dt*1e3:.1f} ms)")
