#!/usr/bin/env python3
"""Small-batch latency of the host entry points (numpy in, numpy out), pybench.py-sized calls, with and without the exact scan."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.getcwd())
import sdf_amd
from sdf_amd import meshes
v, f = meshes.torus(424, 424)
s = sdf_amd.SDF(v, f)
s.set_option("profile", 1)
for n in (1000, 100_000, 1_000_000):
    for dist in ("randn", "uniform"):
        rng = np.random.default_rng(0)
        p = (rng.standard_normal((n, 3)) * 0.5 if dist == "randn" else rng.uniform(-1, 1, (n, 3))).astype(np.float32)
        for _ in range(3): s.nn(p)
        s.set_option("profile", 2)
        t0 = time.perf_counter(); s.nn(p); dt = time.perf_counter() - t0
        st = s.stats
        print(f"n={n} {dist}: {dt*1e6:.0f} us  fallback {st['last_fallback_points']} heavy {st['last_heavy_points']} packet-kernel {st['query_kernel_ms']:.3f} ms")
