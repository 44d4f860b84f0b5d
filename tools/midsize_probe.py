#!/usr/bin/env python3
"""Packet pass vs strict per-lane kernel for mid-size (sparse) batches on the 424^2 torus."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd
from sdf_amd import meshes
v, f = meshes.torus(424, 424)
s = sdf_amd.SDF(v, f, robust=True)
s.set_raycast_axes(meshes.random_rotation(1234))
dev = torch.device("cuda", 0)
for n in (250_000, 1_000_000, 4_000_000, 16_000_000):
    pts = torch.rand((n, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(7)).mul_(2).sub_(1)
    for packet in (1, 0):
        s.set_option("packet", packet)
        row = []
        for name, fn in (("nn", s.nn), ("sdf", s.__call__)):
            fn(pts); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); fn(pts); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            row.append(f"{name} {best*1e3:7.2f} ms")
        print(f"n={n:9d} packet={packet}  " + "  ".join(row), flush=True)
