#!/usr/bin/env python3
"""Small-batch latency of the host entry points (numpy in, numpy out), pybench.py-sized calls."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd
from sdf_amd import meshes
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "golden_teapot.npz"))
for name, (v, f) in (("teapot", (g["verts"], g["faces"])), ("torus424", meshes.torus(424, 424))):
    s = sdf_amd.SDF(v, f)
    for n in (1, 1000, 100_000, 1_000_000):
        p = np.random.default_rng(0).standard_normal((n, 3)).astype(np.float32) * 0.5
        for fn_name, fn in (("sdf", s.__call__), ("contains", s.contains), ("nn", s.nn)):
            for _ in range(3): fn(p)
            ts = []
            for _ in range(20):
                t0 = time.perf_counter(); fn(p); ts.append(time.perf_counter() - t0)
            print(f"{name:9s} n={n:8d} {fn_name:9s} median {np.median(ts)*1e6:9.1f} us  min {min(ts)*1e6:9.1f} us", flush=True)
