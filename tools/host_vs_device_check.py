#!/usr/bin/env python3
"""One-off: the chunked host entry points (pageable numpy arrays, > 8M points: several chunks, bounce buffers,
download thread) return exactly what one device-resident call returns."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sdf_amd
from sdf_amd import meshes
v, f = meshes.torus(300, 300)
s = sdf_amd.SDF(v, f); s.set_raycast_axes(meshes.random_rotation(3))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
pts = np.random.default_rng(1).uniform(-1, 1, (n, 3)).astype(np.float32)
d = torch.from_numpy(pts).cuda()
ok = True
for name, fn in (("sdf", lambda x: s(x)), ("nn", lambda x: s.nn(x)), ("contains", lambda x: s.contains(x))):
    for rep in range(2):
        h = fn(pts)
        g = fn(d).cpu().numpy()
        same = np.array_equal(h.view(np.uint8).reshape(-1), g.view(np.uint8).reshape(-1))
        ok &= same
        print(name, rep, "identical" if same else "DIFFERENT", flush=True)
sys.exit(0 if ok else 1)
