#!/usr/bin/env python3
"""Host entry points: pageable numpy vs arrays in page-locked memory
(sdf_amd.pinned_empty), results written into a reused `out=` array."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd
from sdf_amd import meshes
v, f = meshes.torus(424, 424)
s = sdf_amd.SDF(v, f); s.set_raycast_axes(meshes.random_rotation(1234))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
for kv in sys.argv[2:]:  # key=value handle options
    s.set_option(kv.split("=")[0], int(kv.split("=")[1]))
modes = ("pinned",) if os.environ.get("PINNED_ONLY") else ("pageable", "pinned")
pts = np.random.default_rng(7).uniform(-1, 1, (n, 3)).astype(np.float32)
ppts = sdf_amd.pinned_empty((n, 3), np.float32); ppts[:] = pts
outs = {"sdf": (np.float32, s.__call__), "contains": (np.bool_, s.contains), "nn": (np.int32, s.nn)}
ref = {}
for mode in modes:
    for name, (dt, fn) in outs.items():
        src = ppts if mode == "pinned" else pts
        out = sdf_amd.pinned_empty(n, dt) if mode == "pinned" else np.zeros(n, dt)
        fn(src[:1_000_000], out=out[:1_000_000])
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); fn(src, out=out); best = min(best, time.perf_counter() - t0)
        if name in ref:
            assert (out.view(np.uint8) == ref[name].view(np.uint8)).all(), (mode, name)
        else:
            ref[name] = out.copy()
        print(f"{mode:9s} {name:9s} {n/best/1e6:8.1f} Mevals/s ({best*1e3:.1f} ms incl. H2D/D2H)", flush=True)
