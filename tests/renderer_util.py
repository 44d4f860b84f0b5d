"""shared by the renderer tests: golden meshes moved in front of the camera (tools/make_golden_renderer.py)"""
import os

import numpy as np

from conftest import GOLDEN_DIR, load_golden

RENDERER_NAMES = ["teapot", "torus", "boxframe", "cubegrid", "icosphere3"]


def load_renderer_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, f"golden_renderer_{name}.npz")) as z:
        g = {k: z[k] for k in z.files}
    m = load_golden(name)
    v = m["verts"].copy()
    v[:, 2] += g["cam"][6]
    g["verts"], g["faces"] = v, m["faces"]
    c = g["cam"]
    g["camera"] = (int(c[0]), int(c[1]), float(c[2]), float(c[3]), float(c[4]), float(c[5]))
    return g


def z_tie(g, a, b):
    """True where vertex ids a, b differ but the two vertices have exactly equal z (the one case in
    which the reference's answer depends on its R-tree's visiting order, renderer.cpp:84)"""
    z = g["verts"][:, 2]
    ok = (a >= 0) & (b >= 0)
    out = np.zeros(a.shape, bool)
    out[ok] = (a[ok] != b[ok]) & (z[a[ok]] == z[b[ok]])
    return out
