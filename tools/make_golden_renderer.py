#!/usr/bin/env python3
"""Generate tests/golden/golden_renderer_<mesh>.npz in THIS container (needs /root/reference).

sdf::Renderer (src/renderer.cpp) needs Eigen, so it cannot be built here; what CAN be run are the
two vendored indices it is made of, as it configures them (oracle/ref_shim.cpp):
  RTree<int, float, 2>   over the per-face image-space boxes, inserted in face order
  nanoflann DIM = 2      over the projected vertices (fill_outside)
Per mesh (the golden meshes of tools/make_golden.py, moved in front of the camera):
  cam [8]                width, height, fx, fy, cx, cy, z offset, 0
  points2                image-space points: random + pixel centres + points on projected vertices
  depth, mask, vertex    the reference's handlers (renderer.cpp:63-87, restated by the oracle) over
                         the REAL R-tree's candidates in the REAL search order
  vertex_fill            vertex, with -1 replaced by the REAL nanoflann 2-D nearest vertex
  ncand                  candidate count the real R-tree returned per point

Run:  python tools/make_golden_renderer.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CAM = {  # name: (width, height, fx, fy, cx, cy, z offset)
    "teapot": (160, 120, 260.0, 250.0, 70.0, 40.0, 3.0),
    "torus": (128, 128, 150.0, 150.0, 64.0, 64.0, 2.5),
    "boxframe": (96, 160, 120.0, 200.0, 48.0, 80.0, 3.5),
    "cubegrid": (128, 96, 110.0, 110.0, 64.0, 48.0, 4.0),
    "icosphere3": (100, 100, 90.0, 90.0, 50.0, 50.0, 2.0),
}


def camera_mesh(name):
    with np.load(os.path.join(OUT, f"golden_{name}.npz")) as z:
        v, f = z["verts"].copy(), z["faces"].copy()
    w, h, fx, fy, cx, cy, zo = CAM[name]
    v[:, 2] += np.float32(zo)
    return v, f, (w, h, fx, fy, cx, cy)


def main():
    for name in CAM:
        v, f, cam = camera_mesh(name)
        w, h = cam[0], cam[1]
        o = po.OracleRenderer(v, f, *cam)
        rng = np.random.default_rng(5)
        vc = o.verts_cam
        px = np.stack(np.meshgrid(np.arange(w), np.arange(h)), -1).reshape(-1, 2)
        pts = np.concatenate([
            rng.uniform(-0.2, 1.2, (6000, 2)) * [w, h],
            px[rng.choice(px.shape[0], 4000, replace=False)],
            vc[rng.integers(0, vc.shape[0], 1500)],          # on projected vertices: edge hits
        ]).astype(np.float32)
        mn, mx, valid = o.face_boxes()
        rt = po.RefRTree2(mn, mx, valid)
        kd = po.RefKd2(vc)
        n = pts.shape[0]
        depth = np.zeros(n, np.float32)
        mask = np.zeros(n, np.bool_)
        vert = np.zeros(n, np.int32)
        ncand = np.zeros(n, np.int32)
        for i in range(n):
            cand = rt.point(pts[i])
            ncand[i] = cand.shape[0]
            depth[i], mask[i], vert[i] = o.eval_candidates(pts[i, 0], pts[i, 1], cand)
        fill = vert.copy()
        miss = vert < 0
        fill[miss] = kd.nn(pts[miss])
        np.savez_compressed(os.path.join(OUT, f"golden_renderer_{name}.npz"),
                            cam=np.array(list(cam) + [CAM[name][6], 0], np.float32), points2=pts,
                            depth=depth, mask=mask, vertex=vert, vertex_fill=fill, ncand=ncand)
        print(name, "points", n, "hit", int(mask.sum()), "mean candidates", ncand.mean())


if __name__ == "__main__":
    main()
