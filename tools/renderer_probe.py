#!/usr/bin/env python3
"""Renderer timings: default 1080 x 1080 camera (sdf.hpp:299-301) on the ~360k-face torus, image
renders and 16M device-resident image points; the CPU oracle beside it."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd
from sdf_amd import meshes
from oracle import pyoracle as po
v, f = meshes.torus(424, 424)
v = v.copy(); v[:, 2] += 6.0
r = sdf_amd.Renderer(v, f)  # 1080 x 1080, f = 2600, c = 540
def best(fn, reps=5):
    fn(); b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b * 1e3
t0 = time.perf_counter(); r.update(); print(f"update {1e3*(time.perf_counter()-t0):.2f} ms")
print(f"render_depth {best(r.render_depth):.2f} ms  render_mask {best(r.render_mask):.2f} ms  "
      f"render_nn {best(r.render_nn):.2f} ms  render_nn(fill) {best(lambda: r.render_nn(True)):.2f} ms (host images, incl. D2H)")
dev = torch.device("cuda", 0)
p = torch.rand((16_000_000, 2), device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 1080
for name, fn in (("depth", lambda: r(p)), ("mask", lambda: r.contains(p)), ("nn", lambda: r.nn(p, False)), ("nn fill", lambda: r.nn(p, True))):
    ms = best(fn, 3)
    print(f"device points {name:8s} {ms:8.2f} ms  {p.shape[0]/ms/1e3:9.1f} Mpoints/s")
o = po.OracleRenderer(v, f)
t0 = time.perf_counter(); d = o.render_depth(); dt = time.perf_counter() - t0
print(f"CPU oracle render_depth {dt*1e3:.1f} ms ({po.default_threads()} threads); covered {float((d>0).mean()):.3f}")
assert (d.view(np.uint32) == r.render_depth().view(np.uint32)).all()
