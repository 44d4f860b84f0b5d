"""Mesh utilities: a minimal OBJ reader and seeded synthetic generators.

The reference has no OBJ loader (its demo uses meshview, demo.cpp:52; pybench uses trimesh,
tools/pybench.py:6), so this module supplies the few things the benchmark configurations of
BASELINE.json need.  All generators return (verts float32 [V,3], faces uint32 [F,3]) with
outward counter-clockwise winding (required by the non-robust sign rule, sdf.cpp:149-153).
"""
from __future__ import annotations

import numpy as np


def load_obj(path: str):
    """Read `v x y z` and `f a/b/c ...` lines (1-based, triangles only)."""
    vs, fs = [], []
    with open(path, "r") as fh:
        for line in fh:
            if line.startswith("v "):
                p = line.split()
                vs.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                p = line.split()[1:]
                idx = [int(t.split("/")[0]) for t in p]
                if len(idx) != 3:
                    raise ValueError("only triangle meshes are supported")
                fs.append(tuple(i - 1 if i > 0 else len(vs) + i for i in idx))
    return np.asarray(vs, dtype=np.float32), np.asarray(fs, dtype=np.uint32)


def torus(nu: int, nv: int, R: float = 0.7, r: float = 0.24):
    """Torus around the z axis, nu x nv quads split into 2 triangles each.

    nu=nv=83 -> 6,889 v / 13,778 f (SMPL scale, README.md:153);
    nu=nv=424 -> 179,776 v / 359,552 f (the ~360k-face configuration)."""
    u = np.arange(nu, dtype=np.float64) * (2 * np.pi / nu)
    v = np.arange(nv, dtype=np.float64) * (2 * np.pi / nv)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    x = (R + r * np.cos(vv)) * np.cos(uu)
    y = (R + r * np.cos(vv)) * np.sin(uu)
    z = r * np.sin(vv)
    verts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    i1 = (i + 1) % nu
    j1 = (j + 1) % nv
    a = (i * nv + j).ravel()
    b = (i1 * nv + j).ravel()
    c = (i1 * nv + j1).ravel()
    d = (i * nv + j1).ravel()
    faces = np.concatenate([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], 0)
    # interleave the two triangles of each quad so face ids stay spatially coherent
    faces = faces.reshape(2, -1, 3).transpose(1, 0, 2).reshape(-1, 3)
    return verts, faces.astype(np.uint32)


def icosphere(subdiv: int, radius: float = 1.0, noise: float = 0.0, seed: int = 0):
    """Subdivided icosahedron: 10*4^s+2 verts, 20*4^s faces (s=5: 10,242/20,480;
    s=7: 163,842/327,680).  `noise` adds seeded relative radial noise."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array(
        [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t],
         [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array(
        [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4],
         [11, 10, 2], [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8],
         [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(subdiv):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
        es = np.sort(e, axis=1)
        key = es[:, 0] * (v.shape[0] + 1) + es[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        first = np.zeros(uniq.shape[0], dtype=np.int64)
        first[inv[::-1]] = np.arange(key.shape[0])[::-1]
        mid = v[es[first, 0]] + v[es[first, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = v.shape[0]
        v = np.concatenate([v, mid], 0)
        nf = f.shape[0]
        m01 = base + inv[:nf]
        m12 = base + inv[nf:2 * nf]
        m20 = base + inv[2 * nf:]
        f = np.concatenate([
            np.stack([f[:, 0], m01, m20], -1), np.stack([f[:, 1], m12, m01], -1),
            np.stack([f[:, 2], m20, m12], -1), np.stack([m01, m12, m20], -1)], 0)
    if noise > 0:
        rng = np.random.default_rng(seed)
        v = v * (1.0 + noise * rng.standard_normal((v.shape[0], 1)))
    return (v * radius).astype(np.float32), f.astype(np.uint32)


def cube(half: float = 0.5):
    """Axis-aligned cube, 8 verts / 12 faces, outward CCW."""
    h = half
    v = np.array([[-h, -h, -h], [h, -h, -h], [h, h, -h], [-h, h, -h], [-h, -h, h], [h, -h, h],
                  [h, h, h], [-h, h, h]], dtype=np.float32)
    f = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [2, 3, 7],
                  [2, 7, 6], [1, 2, 6], [1, 6, 5], [3, 0, 4], [3, 4, 7]], dtype=np.uint32)
    return v, f


def grid_points(res: int, lo: float = -1.0, hi: float = 1.0):
    """res^3 grid, p[i,j,k] = lo + (hi-lo)*(i,j,k)/(res-1), i-major."""
    t = (lo + (hi - lo) * np.arange(res, dtype=np.float64) / (res - 1)).astype(np.float32)
    g = np.stack(np.meshgrid(t, t, t, indexing="ij"), -1).reshape(-1, 3)
    return np.ascontiguousarray(g, dtype=np.float32)


def random_rotation(seed: int):
    """Unit quaternion (w,x,y,z) from 4 N(0,1) draws -> row-major 3x3 float32, following
    sdf.cpp:43-48 (Eigen toRotationMatrix term order)."""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal(4).astype(np.float32)
    q = q / np.sqrt(np.float32((q * q).sum(dtype=np.float32)))
    w, x, y, z = [np.float32(t) for t in q]
    two = np.float32(2)
    one = np.float32(1)
    tx, ty, tz = two * x, two * y, two * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    R = np.array([[one - (tyy + tzz), txy - twz, txz + twy],
                  [txy + twz, one - (txx + tzz), tyz - twx],
                  [txz - twy, tyz + twx, one - (txx + tyy)]], dtype=np.float32)
    return np.ascontiguousarray(R)
