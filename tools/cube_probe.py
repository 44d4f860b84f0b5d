#!/usr/bin/env python3
"""Cube lists of the packet pass (option cube_lists) against the plain walk: same bits, time, per-wave work.
usage: cube_probe.py [--n 125000000] [--res 424] [--grid 0] [--level 0] [--stats 1]"""
import argparse, os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=125_000_000); ap.add_argument("--res", type=int, default=424)
ap.add_argument("--grid", type=int, default=0); ap.add_argument("--level", type=int, default=0)
ap.add_argument("--stats", type=int, default=0); ap.add_argument("--mesh", default="torus")
ap.add_argument("--box", type=float, default=1.0); ap.add_argument("--approx", type=int, default=-1)
ap.add_argument("--kinds", default="nn,sdf")
a = ap.parse_args()
if a.stats:
    os.environ["SDFHIP_LIB"] = os.path.join(ROOT, "sdf_amd", "libsdfhip_stats.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
import sdf_amd
from sdf_amd import meshes
v, f = meshes.torus(a.res, a.res) if a.mesh == "torus" else meshes.icosphere(int(a.mesh[3:]), radius=0.8, noise=0.05 if a.mesh.endswith("n") else 0.0, seed=3)
s = sdf_amd.SDF(v, f, robust=True); s.set_raycast_axes(meshes.random_rotation(1234))
s.set_option("approx", a.approx)
if a.level: s.set_option("cube_level", a.level)
dev = torch.device("cuda", 0)
if a.grid:
    t = torch.linspace(-a.box, a.box, a.grid, device=dev)
    pts = torch.stack(torch.meshgrid(t, t, t, indexing="ij"), -1).reshape(-1, 3).contiguous()
else:
    pts = torch.rand((a.n, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(7)).mul_(2 * a.box).sub_(a.box)
L = s._L
st = (C.c_ulonglong * 12)()
res = {}
for lists in (0, 1):
    s.set_option("cube_lists", lists)
    for kind in a.kinds.split(","):
        fn = {"nn": s.nn, "sdf": s.__call__}[kind]
        out = fn(pts); torch.cuda.synchronize()
        if a.stats: L.sdfhip_debug_packet_stats(st, 1)
        best = 1e9
        s.set_option("profile", 2)
        for _ in range(3 if not a.stats else 1):
            t0 = time.perf_counter(); out = fn(pts); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        sst = s.stats
        line = f"lists={lists} {kind:4s} {best*1e3:8.2f} ms  kernel {sst['query_kernel_ms']/max(1,sst['query_kernel_launches']):7.2f} ms  heavy {sst['last_heavy_points']} fallback {sst['last_fallback_points']}"
        if a.stats:
            L.sdfhip_debug_packet_stats(st, 1); w = max(1, st[0])
            line += f" | waves {st[0]} list-waves {st[11]/w:.3f} wide {st[1]/w:.1f} cands {st[5]/w:.1f} groups {st[2]/w:.1f} chunks {st[3]/w:.1f}"
        print(line, flush=True)
        key = kind
        if lists == 0: res[key] = out
        else:
            same = bool((out.view(torch.int32) == res[key].view(torch.int32)).all()) if kind == "sdf" else bool((out == res[key]).all())
            print(f"   {kind}: lists == walk: {same}", flush=True)
            if not same:
                bad = (out != res[key]).nonzero().flatten()[:5].tolist(); print("   first differing:", bad)

if a.stats:
    hdr, lst, cap = C.c_void_p(), C.c_void_p(), C.c_ulonglong()
    L.sdfhip_debug_cube_tables.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_ulonglong)]
    L.sdfhip_debug_cube_tables(s._h, C.byref(hdr), C.byref(lst), C.byref(cap))
    nc3 = cap.value
    buf = (C.c_float * (12 * nc3))()
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(buf, hdr, 48 * nc3, 2)
    H = np.frombuffer(buf, dtype=np.float32).reshape(nc3, 3, 4)
    cnt = H[:, 0, 3].copy().view(np.uint32)
    bad = cnt == 0xFFFFFFFF
    print(f"cubes {nc3}: unusable {int(bad.sum())} ({bad.mean()*100:.1f} %), list length of the rest: mean {cnt[~bad].mean():.1f} median {np.median(cnt[~bad]):.0f} p90 {np.percentile(cnt[~bad], 90):.0f} max {cnt[~bad].max()}")
    c = H[:, 0, :3]
    rho_t = np.sqrt(c[:, 0] ** 2 + c[:, 1] ** 2) - 0.7
    d = np.abs(np.sqrt(rho_t ** 2 + c[:, 2] ** 2) - 0.24)
    for lo_, hi_ in ((0, .05), (.05, .1), (.1, .2), (.2, .4), (.4, .6), (.6, .8), (.8, 2)):
        m = (d >= lo_) & (d < hi_)
        if m.sum():
            ok = m & ~bad
            print(f"  dist to the torus {lo_:.2f}-{hi_:.2f}: cubes {int(m.sum()):6d} unusable {bad[m].mean()*100:5.1f} %  mean list {cnt[ok].mean() if ok.sum() else 0:.0f}")
