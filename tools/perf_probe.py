#!/usr/bin/env python3
"""Times nn / contains / sdf separately (device-resident points) for kernel tuning."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd
from sdf_amd import meshes

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=424)
ap.add_argument("--n", type=int, default=16_000_000)
ap.add_argument("--grid", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--robust", type=int, default=1)
ap.add_argument("--only", default="")
ap.add_argument("--mesh", default="torus", help="torus | ico<subdiv> (radius 0.8, 5%% radial noise) | teapot")
ap.add_argument("--nosort", type=int, default=0)
ap.add_argument("--ray-mode", type=int, default=0)
ap.add_argument("--noise", type=float, default=0.05)
ap.add_argument("--group", type=int, default=0, help="packet_group option (0 = library default)")
ap.add_argument("--heavy-budget", type=int, default=0)
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--opt", action="append", default=[], help="key=value handle option, set before the rotation is pinned")
ap.add_argument("--rmin", type=float, default=0.0, help="drop points closer than this to the origin")
ap.add_argument("--offset", type=float, default=0.0, help="translate mesh and points by (offset, -offset/2, offset/4)")
ap.add_argument("--presort", type=int, default=0, help="1: the points already lie in Morton order in memory (the query sort then yields ~identity: coalesced input gather)")
a = ap.parse_args()
if a.mesh.startswith("ico"):
    v, f = meshes.icosphere(int(a.mesh[3:]), radius=0.8, noise=a.noise, seed=3)
elif a.mesh == "teapot":
    import numpy as _np
    g = _np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "golden_teapot.npz"))
    v, f = g["verts"], g["faces"]
else:
    v, f = meshes.torus(a.res, a.res)
if a.offset:
    import numpy as _np2
    v = (v + _np2.array([a.offset, -a.offset / 2, a.offset / 4], _np2.float32)).astype(_np2.float32)
print("mesh", a.mesh, v.shape, f.shape)
s = sdf_amd.SDF(v, f, robust=bool(a.robust))
if a.group:
    s.set_option("packet_group", a.group)
    s.update()
for kv in a.opt:
    k, v_ = kv.split("=")
    s.set_option(k, int(v_))
s.set_raycast_axes(meshes.random_rotation(1234))
dev = torch.device("cuda", 0)
if a.grid:
    t = torch.linspace(-1, 1, a.grid, device=dev)
    pts = torch.stack(torch.meshgrid(t, t, t, indexing="ij"), -1).reshape(-1, 3).contiguous()
else:
    g = torch.Generator(device=dev).manual_seed(7)
    pts = torch.rand((a.n, 3), device=dev, generator=g).mul_(2).sub_(1)
if a.offset:
    pts = (pts + torch.tensor([a.offset, -a.offset / 2, a.offset / 4], device=dev)).contiguous()
if a.rmin > 0:
    pts = pts[pts.norm(dim=1) >= a.rmin].contiguous()
if a.presort:
    q = ((pts + 1) * 128).clamp_(0, 255).to(torch.int64)
    def spread(x):
        x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F; x = (x | (x << 4)) & 0x030C30C3; x = (x | (x << 2)) & 0x09249249; return x
    key = (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])
    pts = pts[torch.argsort(key)].contiguous(); del q, key
n = pts.shape[0]
if a.ray_mode:
    s.set_option("ray_mode", a.ray_mode)
if a.chunk:
    s.set_option("chunk", a.chunk)
if a.heavy_budget:
    s.set_option("heavy_budget", a.heavy_budget)
if a.nosort:
    s.set_option("sort_queries", 0)
print("stats", s.stats)
s.set_option("profile", 1)
for name, fn in (("nn", s.nn), ("contains", s.contains), ("sdf", s.__call__)):
    if a.only and name != a.only:
        continue
    fn(pts); torch.cuda.synchronize()
    best = 1e9
    s.set_option("profile", 2)
    for _ in range(a.reps):
        t0 = time.perf_counter(); fn(pts); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{name:9s} {best*1e3:9.2f} ms  {n/best/1e6:9.1f} Mevals/s  fallback {s.stats['last_fallback_points']} heavy {s.stats['last_heavy_points']}  kernel {s.stats['query_kernel_ms']/max(1,s.stats['query_kernel_launches']):.2f} ms", flush=True)
