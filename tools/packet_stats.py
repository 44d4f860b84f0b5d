#!/usr/bin/env python3
"""Per-wave work of the packet pass (needs `make -C sdf_amd/csrc stats`; SDFHIP_LIB is set here)."""
import argparse, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SDFHIP_LIB"] = os.environ.get("SDFHIP_STATS_LIB", os.path.join(ROOT, "sdf_amd", "libsdfhip_stats.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sdf_amd
from sdf_amd import meshes
ap = argparse.ArgumentParser()
ap.add_argument("--mesh", default="torus"); ap.add_argument("--noise", type=float, default=0.0)
ap.add_argument("--n", type=int, default=16_000_000); ap.add_argument("--grid", type=int, default=0)
ap.add_argument("--fast-nn", type=int, default=0); ap.add_argument("--group", type=int, default=0)
ap.add_argument("--res", type=int, default=424); ap.add_argument("--approx", type=int, default=-1)
ap.add_argument("--box", type=float, default=1.0)
ap.add_argument("--repeat", type=int, default=1, help="every point R times in a row: a wave then holds 64/R distinct queries")
a = ap.parse_args()
v, f = meshes.torus(a.res, a.res) if a.mesh == "torus" else meshes.icosphere(int(a.mesh[3:]), radius=0.8, noise=a.noise, seed=3)
s = sdf_amd.SDF(v, f, robust=False)
if a.fast_nn: s.set_option("fast_nn", 1)
if a.group: s.set_option("packet_group", a.group)
if a.approx >= 0: s.set_option('approx', a.approx)
if a.fast_nn or a.group or a.approx >= 0: s.update()
dev = torch.device("cuda", 0)
if a.grid:
    t = torch.linspace(-1, 1, a.grid, device=dev)
    pts = torch.stack(torch.meshgrid(t, t, t, indexing="ij"), -1).reshape(-1, 3).contiguous()
else:
    pts = torch.rand((a.n, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(7)).mul_(2).sub_(1).mul_(a.box)
if a.repeat > 1:
    pts = pts.repeat_interleave(a.repeat, dim=0).contiguous()
s.set_option('profile', 1)
L = s._L
st = (C.c_ulonglong * 12)()
L.sdfhip_debug_packet_stats(st, 1)
s.nn(pts); torch.cuda.synchronize()
L.sdfhip_debug_packet_stats(st, 1)
w = max(1, st[0])
print("heavy", s.stats["last_heavy_points"], "fallback", s.stats["last_fallback_points"], "fourth-candidate lanes (approx)", st[7])
print(f"{a.mesh} noise {a.noise} {'grid%d' % a.grid if a.grid else 'random'} n={pts.shape[0]} fast={a.fast_nn}: waves {st[0]}  per wave: quad nodes {st[1]/w:.1f}  groups {st[2]/w:.1f}  chunks {st[3]/w:.1f} (= {4*st[3]/w:.0f} vertex slots)  tie redos {st[4]/w:.2f}  pops {st[5]/w:.1f}  near-tie lanes (path check) {st[6]/w:.2f}  deferred lanes {st[7]/w:.3f}  path checks: leaf-box passes {st[8]/w:.2f} (failing the walk: {st[9]}, must be 0)  walks {st[10]/w:.3f}")
