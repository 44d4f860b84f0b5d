#!/usr/bin/env python3
"""Odd meshes under mixed near / far batches, against the oracle: one triangle, all vertices equal, a mesh that is a line, huge and tiny scales."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po
from sdf_amd import SDF, meshes
rng = np.random.default_rng(1)
def queries(c, r, n=60000):
    r = max(r, 1e-3)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([c + rng.uniform(-2 * r, 2 * r, (n, 3)), c + d * r * 10.0 ** rng.uniform(0, 4, (n, 1))]).astype(np.float32)
cases = {}
cases["one triangle"] = (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.uint32))
cases["all equal"] = (np.tile(np.float32([[0.3, -0.2, 0.1]]), (300, 1)), rng.integers(0, 300, (100, 3)).astype(np.uint32))
t = np.linspace(0, 1, 2000, dtype=np.float32)
cases["a line"] = (np.stack([t, 2 * t, -t], 1), np.stack([np.arange(0, 1998), np.arange(1, 1999), np.arange(2, 2000)], 1).astype(np.uint32))
v, f = meshes.torus(60, 60)
cases["huge"] = ((v * np.float32(1e6)).astype(np.float32), f)
cases["tiny"] = ((v * np.float32(1e-6) + np.float32([1, 1, 1])).astype(np.float32), f)
bad = 0
for name, (v, f) in cases.items():
    fin = v[np.isfinite(v).all(1)]
    c, r = fin.mean(0).astype(np.float64), float(np.abs(fin - fin.mean(0)).max())
    q = queries(c, r)
    o = po.OracleSDF(v, f, False)
    for approx in (0, 1):
        s = SDF(v, f, robust=False)
        s.set_option("approx", approx); s.set_option("scan_max", 0); s.set_option("profile", 1)
        got = s.nn(q); want = o.nn(q)
        a, b = s(q), o(q)
        ok_nn = bool((got == want).all())
        ok_sdf = bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
        print(f"{name:14s} approx {approx}: nn {'ok' if ok_nn else 'MISMATCH'} sdf {'ok' if ok_sdf else 'MISMATCH'}  kernel {s.stats['last_kernel']} heavy {s.stats['last_heavy_points']}", flush=True)
        bad += (not ok_nn) + (not ok_sdf)
print("mismatching:", bad)
sys.exit(1 if bad else 0)
