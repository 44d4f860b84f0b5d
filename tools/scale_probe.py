#!/usr/bin/env python3
"""Large-mesh check: torus res x res (res = 1000 -> 1M vertices / 2M faces): construction, queries,
oracle parity on a sample."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd
from sdf_amd import meshes
from oracle import pyoracle as po
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
v, f = meshes.torus(res, res)
R = meshes.random_rotation(1234)
t0 = time.perf_counter(); s = sdf_amd.SDF(v, f); s.set_raycast_axes(R); print(f"mesh {v.shape[0]} v / {f.shape[0]} f: construct {1e3*(time.perf_counter()-t0):.1f} ms, stats {s.stats}")
for kv in sys.argv[2:]:  # key=value handle options
    s.set_option(kv.split('=')[0], int(kv.split('=')[1]))
t0 = time.perf_counter(); s.update(); print(f"update {1e3*(time.perf_counter()-t0):.1f} ms")
dev = torch.device("cuda", 0)
pts = torch.rand((32_000_000, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(7)).mul_(2).sub_(1)
for name, fn in (("nn", s.nn), ("contains", s.contains), ("sdf", s.__call__)):
    fn(pts); torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(pts); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = s.stats
    print(f"{name:9s} 32M points {dt*1e3:8.2f} ms  {32e6/dt/1e6:8.1f} Mevals/s  fallback {st['last_fallback_points']} heavy {st['last_heavy_points']}")
hp = pts[:60000].cpu().numpy()
o = po.OracleSDF(v, f, True, R)
assert (s.nn(hp) == o.nn(hp)).all() and (s.contains(hp) == o.contains(hp)).all()
a, b = s(hp), o(hp)
assert ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()
print("oracle parity on 60000 points: ok")
