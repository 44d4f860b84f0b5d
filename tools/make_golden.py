#!/usr/bin/env python3
"""Generate tests/golden/*.npz in THIS container (needs /root/reference).

What is pinned, and by what:
  golden_<mesh>.npz    for each of the reference's four sample meshes (sample-obj/*.obj, BSD-2,
                       LICENSE.txt; stored as arrays — data, not source) + two seeded synthetic
                       meshes:
      verts, faces                      the mesh
      points, R                         inputs (grid + random + vertex-snapped points)
      nn, nn_dist                       from the reference's REAL nanoflann (oracle/_ref)
      vind                              nanoflann's point permutation after buildIndex
      ray                               [N,3] parities {z+, z-, x+|-1}: candidates from the
                                        reference's REAL RTree::Search, leaf predicate
                                        (sdf.cpp:297-313) restated by the oracle
      ray_ncand                         [N,3] candidate counts returned by the real R-tree
      contains_robust                   majority vote of `ray` (sdf.cpp:325-327)
      sdf_robust, sdf_robust_trunc,     restated-oracle outputs (regression vectors; the
      sdf_plain, sdf_plain_trunc,       Eigen-dependent arithmetic is "parity unpinned",
      contains_plain                    see oracle/sdf_oracle.cpp header)
      face_normals, face_areas, aabb    restated-oracle precompute (sdf.cpp:31-73)

Run:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from sdf_amd import meshes  # noqa: E402

REF_OBJ = "/root/reference/sample-obj"
OUT = os.path.join(ROOT, "tests", "golden")


def make_points(verts, seed, n_rand=5000, n_snap=2000, grid=17):
    rng = np.random.default_rng(seed)
    g = meshes.grid_points(grid)
    r = rng.uniform(-1, 1, (n_rand, 3)).astype(np.float32)
    # inside the mesh AABB so that rays are actually cast
    lo, hi = verts.min(0), verts.max(0)
    r2 = (lo + (hi - lo) * rng.uniform(0, 1, (n_rand // 2, 3))).astype(np.float32)
    # vertex-snapped coordinates provoke edge / vertex hits and exact distance ties
    idx = rng.integers(0, verts.shape[0], n_snap)
    sn = verts[idx].copy()
    mask = rng.integers(0, 2, (n_snap, 3)).astype(bool)
    other = verts[rng.integers(0, verts.shape[0], n_snap)]
    sn[mask] = other[mask]
    return np.ascontiguousarray(np.concatenate([g, r, r2, sn]), dtype=np.float32)


def one(name, v, f, seed):
    R = meshes.random_rotation(1234 + seed)
    pts = make_points(v, seed)
    o = po.OracleSDF(v, f, True, R)
    on = po.OracleSDF(v, f, False)
    rk = po.RefKd(v)
    nn, nnd = rk.nn(pts)
    assert (o.nn(pts) == nn).all(), "oracle kd-tree disagrees with real nanoflann"
    mn, mx = o.face_boxes()
    rt = po.RefRTree(mn, mx)
    det = o.raycast_detail(pts)
    ray = np.full((pts.shape[0], 3), -1, np.int32)
    ncand = np.zeros((pts.shape[0], 3), np.int32)
    aabb = o.aabb
    inside_box = np.all((pts >= aabb[:3]) & (pts <= aabb[3:]), axis=1)
    for i in np.nonzero(inside_box)[0]:
        q = o.rotate_point(pts[i])
        c = rt.ray(q, 2, 0)
        ray[i, 0] = o.ray_parity_candidates(q, 2, 0, c)
        ncand[i, 0] = len(c)
        c = rt.ray(q, 2, 1)
        ray[i, 1] = o.ray_parity_candidates(q, 2, 1, c)
        ncand[i, 1] = len(c)
        if ray[i, 0] + ray[i, 1] == 1:
            c = rt.ray(q, 0, 0)
            ray[i, 2] = o.ray_parity_candidates(q, 0, 0, c)
            ncand[i, 2] = len(c)
    assert (ray == det).all(), "oracle ray filter disagrees with real RTree candidates"
    tot = np.where(ray[:, 0] >= 0, ray[:, 0] + ray[:, 1] + np.maximum(ray[:, 2], 0), 0)
    contains_robust = tot > 1
    assert (contains_robust == o.contains(pts)).all()
    d = dict(
        verts=v, faces=f, points=pts, R=R, nn=nn, nn_dist=nnd, vind=rk.vind(), ray=ray,
        ray_ncand=ncand, contains_robust=contains_robust,
        sdf_robust=o(pts), sdf_robust_trunc=o(pts, True), sdf_plain=on(pts),
        sdf_plain_trunc=on(pts, True), contains_plain=on.contains(pts),
        face_normals=o.face_normals, face_areas=o.face_areas, aabb=aabb)
    np.savez_compressed(os.path.join(OUT, f"golden_{name}.npz"), **d)
    print(f"{name}: V={v.shape[0]} F={f.shape[0]} N={pts.shape[0]} inside_box={inside_box.sum()} "
          f"contains={contains_robust.sum()} x-tiebreaks={(ray[:, 2] >= 0).sum()}")


def main():
    os.makedirs(OUT, exist_ok=True)
    po.build()
    for k, name in enumerate(["teapot", "torus", "boxframe", "cubegrid"]):
        v, f = meshes.load_obj(os.path.join(REF_OBJ, name + ".obj"))
        one(name, v, f, k)
    v, f = meshes.torus(83, 83)
    one("syntorus83", v, f, 7)
    v, f = meshes.icosphere(3, 0.8, 0.05, 5)
    one("icosphere3", v, f, 9)


if __name__ == "__main__":
    main()
