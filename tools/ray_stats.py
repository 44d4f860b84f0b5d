#!/usr/bin/env python3
"""Per-query work of the ray-parity walk (needs `make -C sdf_amd/csrc stats`): faces tested per query in the
mesh box against faces that pass Overlap in (x', y') and faces whose bary2d is evaluated."""
import argparse, os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SDFHIP_LIB"] = os.environ.get("SDFHIP_STATS_LIB", os.path.join(ROOT, "sdf_amd", "libsdfhip_stats.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sdf_amd
from sdf_amd import meshes
ap = argparse.ArgumentParser()
ap.add_argument("--mesh", default="ico7"); ap.add_argument("--noise", type=float, default=0.05)
ap.add_argument("--n", type=int, default=4_000_000); ap.add_argument("--ray-mode", type=int, default=1)
a = ap.parse_args()
v, f = meshes.torus(424, 424) if a.mesh == "torus" else meshes.icosphere(int(a.mesh[3:]), radius=0.8, noise=a.noise, seed=3)
s = sdf_amd.SDF(v, f, robust=True)
s.set_raycast_axes(meshes.random_rotation(1234))
s.set_option("ray_mode", a.ray_mode)
pts = torch.rand((a.n, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)).mul_(2).sub_(1)
st = (C.c_ulonglong * 8)()
s._L.sdfhip_debug_ray_stats(st, 1)
s.contains(pts); torch.cuda.synchronize()
s._L.sdfhip_debug_ray_stats(st, 1)
q = max(1, st[0])
print(f"{a.mesh} noise {a.noise} faces {f.shape[0]}: queries in box {st[0]} ({st[0]/a.n:.2f} of all); per such query: "
      f"faces tested {st[1]/q:.1f}  pass Overlap xy (bary2d evaluated) {st[2]/q:.1f}  inside the triangle (record fetched) {st[3]/q:.1f}\n"
      f"  x+ tie-break rays {st[4]} ({st[4]/q:.4f} per query), nodes per x+ ray {st[5]/max(1,st[4]):.0f}, longest x+ walk {st[6]} nodes, longest cell list {st[7]}")
