#!/usr/bin/env python3
"""One-off wide parity sweep (GPU box): random meshes x query clouds x both packet walks, GPU results against the
CPU oracle, bit for bit.  Not part of the test suite (minutes of oracle time); the suite's cases came out of runs
like this one.  usage: python tools/fuzz_parity.py [seed0 [nseeds [points [big]]]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import pyoracle as po
from sdf_amd import SDF, meshes

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
npts = int(sys.argv[3]) if len(sys.argv) > 3 else 400_000


def same(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


BIG = len(sys.argv) > 4 and sys.argv[4] == "big"  # meshes of 40k-180k vertices: cooperative continuation, exact scan


def mesh(rng, kind):
    if BIG and kind in (0, 2):
        v, f = meshes.torus(int(rng.integers(200, 425)), int(rng.integers(200, 425)))
    elif BIG:
        v, f = meshes.icosphere(int(rng.integers(6, 8)), radius=float(rng.uniform(0.2, 0.9)), noise=float(rng.choice([0, 0.02, 0.05])), seed=int(rng.integers(1 << 30)))
    elif kind == 0:
        v, f = meshes.torus(int(rng.integers(20, 160)), int(rng.integers(20, 160)))
    elif kind == 1:
        v, f = meshes.icosphere(int(rng.integers(2, 6)), radius=float(rng.uniform(0.2, 0.9)), noise=float(rng.choice([0, 0.02, 0.1])), seed=int(rng.integers(1 << 30)))
    elif kind == 2:  # triangle soup: open, self-intersecting, degenerate faces among them
        nv = int(rng.integers(50, 5000))
        v = rng.uniform(-1, 1, (nv, 3)).astype(np.float32)
        f = rng.integers(0, nv, (2 * nv, 3)).astype(np.uint32)
    else:  # lattice-aligned vertices: exact ties
        g = np.stack(np.meshgrid(*[np.arange(-5, 6)] * 3, indexing="ij"), -1).reshape(-1, 3)
        v = (g[rng.permutation(g.shape[0])[: int(rng.integers(200, 1300))]] * 0.125).astype(np.float32)
        f = rng.integers(0, v.shape[0], (2 * v.shape[0], 3)).astype(np.uint32)
    s = np.float32(rng.choice([1.0, 1.0, 0.01, 37.0]))
    t = (rng.choice([0.0, 0.0, 1.0]) * rng.uniform(-100, 100, 3)).astype(np.float32)
    return (v * s + t).astype(np.float32), f, s, t


bad = 0
t0 = time.time()
for seed in range(seed0, seed0 + nseeds):
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 4))
    v, f, sc, tr = mesh(rng, kind)
    robust = bool(rng.integers(0, 2))
    R = meshes.random_rotation(int(rng.integers(1 << 30)))
    lo, hi = v.min(0), v.max(0)
    c, e = 0.5 * (lo + hi), 0.5 * (hi - lo).max()
    q = np.concatenate([
        rng.uniform(-1.3, 1.3, (npts // 2, 3)) * e + c,                      # box around the mesh
        rng.normal(0, 1, (npts // 4, 3)) * e * float(rng.choice([0.01, 1, 30])) + c,  # near the centre / far field
        v[rng.integers(0, v.shape[0], npts // 8)] + rng.normal(0, 1e-4, (npts // 8, 3)) * e,  # next to vertices
        (np.round(rng.uniform(-1, 1, (npts // 8, 3)) * 16) / 16) * e + c,    # lattice: ties
    ]).astype(np.float32)
    o = po.OracleSDF(v, f, robust, R)
    ref_nn, ref_sdf, ref_in = o.nn(q), o(q), o.contains(q)
    for approx in (1, 0):
        s = SDF(v, f, robust=robust)
        s.set_raycast_axes(R)
        s.set_option("approx", approx)
        s.set_option("scan_max", 0)
        ok = [(s.nn(q) == ref_nn).all(), same(s(q), ref_sdf).all(), (s.contains(q) == ref_in).all(),
              (s.nn(q[:3000]) == ref_nn[:3000]).all(), same(s(q[:70000]), ref_sdf[:70000]).all()]
        if not all(ok):
            bad += 1
        print(f"seed {seed} kind {kind} verts {v.shape[0]} faces {f.shape[0]} scale {sc} shift {tr} robust {robust} approx {approx}: "
              f"{'ok' if all(ok) else 'MISMATCH ' + str(ok)}  [{time.time() - t0:.0f} s]", flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
