"""CPU: the Renderer restatement (oracle/sdf_oracle.cpp, orc_rend_*) against golden vectors made
from the reference's REAL vendored RTree<int,float,2> / nanoflann DIM=2 (tools/
make_golden_renderer.py), and — when oracle/_ref is built — against those libraries directly."""
import numpy as np
import pytest

from conftest import same_float
from oracle import pyoracle as po
from renderer_util import RENDERER_NAMES, load_renderer_golden, z_tie


@pytest.fixture(scope="module", params=RENDERER_NAMES)
def rg(request):
    g = load_renderer_golden(request.param)
    g["name"] = request.param
    return g


def test_depth_mask_vertex_match_real_rtree_golden(rg):
    o = po.OracleRenderer(rg["verts"], rg["faces"], *rg["camera"])
    p = rg["points2"]
    assert same_float(o(p), rg["depth"]).all()
    assert (o.contains(p) == rg["mask"]).all()
    v = o.nn(p, fill_outside=False)
    diff = v != rg["vertex"]
    # the only admissible difference: two different vertices of exactly equal z (documented rule)
    assert (diff == z_tie(rg, v, rg["vertex"])).all()
    assert ((v < 0) == (rg["vertex"] < 0)).all()
    if rg["name"] != "cubegrid":  # axis-aligned cubes facing the camera: every face is a z tie
        assert diff.mean() < 1e-3


def test_fill_outside_matches_real_nanoflann_2d_golden(rg):
    o = po.OracleRenderer(rg["verts"], rg["faces"], *rg["camera"])
    p = rg["points2"]
    miss = rg["vertex"] < 0
    assert (o.nn(p, fill_outside=True)[miss] == rg["vertex_fill"][miss]).all()


def test_images_equal_point_queries_at_pixel_centres(rg):
    o = po.OracleRenderer(rg["verts"], rg["faces"], *rg["camera"])
    w, h = rg["camera"][:2]
    px = np.stack(np.meshgrid(np.arange(w), np.arange(h)), -1).reshape(-1, 2).astype(np.float32)
    assert same_float(o.render_depth().ravel(), o(px)).all()
    assert (o.render_mask().ravel() == o.contains(px)).all()
    assert (o.render_nn(True).ravel() == o.nn(px, True)).all()
    assert o.render_depth().shape == (h, w)
    assert ((o.render_depth() > 0) == o.render_mask()).all()


@pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_against_real_indices_directly():
    g = load_renderer_golden("torus")
    v = g["verts"].copy()
    v[:, 2] += np.float32(0.5)
    o = po.OracleRenderer(v, g["faces"], 96, 96, 130.0, 140.0, 50.0, 44.0)
    mn, mx, valid = o.face_boxes()
    rt, kd = po.RefRTree2(mn, mx, valid), po.RefKd2(o.verts_cam)
    rng = np.random.default_rng(3)
    p = rng.uniform(-10, 106, (4000, 2)).astype(np.float32)
    d, m, n = o(p), o.contains(p), o.nn(p, True)
    for i in range(p.shape[0]):
        dd, mm, vv = o.eval_candidates(p[i, 0], p[i, 1], rt.point(p[i]))
        assert np.float32(dd).view(np.uint32) == d[i].view(np.uint32) and mm == m[i]
        if vv >= 0:
            assert vv == n[i] or v[vv, 2] == v[n[i], 2]
    assert (kd.nn(p[~m]) == n[~m]).all()


def test_nan_faces_are_skipped_and_update_rereads_vertices():
    v = np.array([[0, 0, 2], [1, 0, 2], [0, 1, 2], [0, 0, 0], [1, 1, 3]], np.float32)
    f = np.array([[0, 1, 2], [3, 0, 4]], np.uint32)  # second face starts at a z = 0 vertex: 0/0 = NaN
    # (std::min / max keep a NaN only when it comes first, renderer.cpp:46-51)
    o = po.OracleRenderer(v, f, 8, 8, 4.0, 4.0, 1.0, 1.0)
    assert o.face_boxes()[2].tolist() == [1, 0]
    assert o.contains(np.array([[1.5, 1.5], [5.0, 5.0]], np.float32)).tolist() == [True, False]
    assert o(np.array([[1.5, 1.5]], np.float32))[0] == 2.0
    v[:3, 2] = 4.0
    o.update()
    assert o(np.array([[1.2, 1.2]], np.float32))[0] == 4.0
