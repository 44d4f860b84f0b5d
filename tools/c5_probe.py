#!/usr/bin/env python3
"""BASELINE.json config 5: deforming mesh, frames of vertices_mutable + update() with
contains() + nn() on 4M device-resident points per frame.  Prints the per-frame split."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd
from sdf_amd import meshes
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=20); ap.add_argument("--res", type=int, default=424)
ap.add_argument("--device-verts", type=int, default=0); ap.add_argument("--fast-nn", type=int, default=0)
ap.add_argument("--pipeline", type=int, default=0, help="1: frames issued back to back (no sync inside a frame), results stay on the device")
a = ap.parse_args()
v0, f = meshes.torus(a.res, a.res)
s = sdf_amd.SDF(v0, f, robust=True, copy=True)
if a.fast_nn:
    s.set_option("fast_nn", 1)
dev = torch.device("cuda", 0)
pts = torch.rand((4_000_000, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(7)).mul_(2).sub_(1)
tv0 = torch.from_numpy(v0).to(dev)
if a.pipeline:
    keep = []
    for t in range(a.frames + 2):
        if t == 2:
            torch.cuda.synchronize(); t00 = time.perf_counter()
        v = (v0 * (1 + 0.05 * np.sin(2 * np.pi * t / 100 + 4 * v0[:, 1:2]))).astype(np.float32) if not a.device_verts else None
        if a.device_verts:
            s.update_from_device(tv0 * (1 + 0.05 * torch.sin(2 * np.pi * t / 100 + 4 * tv0[:, 1:2])))
        else:
            s.vertices_mutable = v; s.update()
        keep.append((s.contains(pts), s.nn(pts)))
        keep = keep[-2:]
    torch.cuda.synchronize(); t11 = time.perf_counter()
    print(f"C5 pipelined: {(t11 - t00) / a.frames * 1e3:.2f} ms per frame (update + contains(4M) + nn(4M), frames back to back) inside {int(keep[-1][0].sum())}")
    sys.exit(0)
tu = tc = tn = 0.0
for t in range(a.frames + 2):
    if a.device_verts:
        tv = tv0 * (1 + 0.05 * torch.sin(2 * np.pi * t / 100 + 4 * tv0[:, 1:2]))
        torch.cuda.synchronize(); t0 = time.perf_counter(); s.update_from_device(tv)
    else:
        v = (v0 * (1 + 0.05 * np.sin(2 * np.pi * t / 100 + 4 * v0[:, 1:2]))).astype(np.float32)
        t0 = time.perf_counter(); s.vertices_mutable = v; s.update()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    c = s.contains(pts); torch.cuda.synchronize(); t2 = time.perf_counter()
    n = s.nn(pts); torch.cuda.synchronize(); t3 = time.perf_counter()
    if t >= 2:
        tu += t1 - t0; tc += t2 - t1; tn += t3 - t2
k = a.frames
st = s.stats
print(f"C5 per frame: update {tu/k*1e3:.2f} ms (kd host build {st['last_kd_build_ms']:.2f}) contains(4M) {tc/k*1e3:.2f} ms nn(4M) {tn/k*1e3:.2f} ms total {(tu+tc+tn)/k*1e3:.2f} ms inside {int(c.sum())}")
