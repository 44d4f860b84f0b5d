"""GPU: sdf_amd.Renderer (HIP, through the C ABI) against the golden vectors made from the
reference's real vendored RTree / nanoflann (tools/make_golden_renderer.py) and against the oracle.
Bar: depth bit-exact, mask and fill_outside vertex ids exact; hit vertex ids exact except when two
different vertices of exactly equal z compete (documented rule, rate printed)."""
import numpy as np
import pytest

from conftest import same_float
from oracle import pyoracle as po
from renderer_util import RENDERER_NAMES, load_renderer_golden, z_tie
from sdf_amd import Renderer, SdfHipError, meshes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=RENDERER_NAMES)
def rg(request):
    g = load_renderer_golden(request.param)
    g["name"] = request.param
    return g


def test_point_queries_golden(rg):
    r = Renderer(rg["verts"], rg["faces"], *rg["camera"])
    p = rg["points2"]
    assert same_float(r(p), rg["depth"]).all()
    assert (r.contains(p) == rg["mask"]).all()
    v = r.nn(p, fill_outside=False)
    diff = v != rg["vertex"]
    assert (diff == z_tie(rg, v, rg["vertex"])).all()
    print(f"\nrenderer {rg['name']}: vertex id differs from the reference's R-tree order on "
          f"{diff.mean() * 100:.3f} % of points (all exact z ties)")
    vf = r.nn(p, fill_outside=True)
    miss = rg["vertex"] < 0
    assert (vf[miss] == rg["vertex_fill"][miss]).all() and (vf[~miss] == v[~miss]).all()


def test_images_match_oracle(rg):
    r = Renderer(rg["verts"], rg["faces"], *rg["camera"])
    o = po.OracleRenderer(rg["verts"], rg["faces"], *rg["camera"])
    assert same_float(r.render_depth(), o.render_depth()).all()
    assert (r.render_mask() == o.render_mask()).all()
    assert (r.render_nn() == o.render_nn()).all()          # same order-independent rule
    assert (r.render_nn(True) == o.render_nn(True)).all()
    assert r.render_depth().shape == (rg["camera"][1], rg["camera"][0])


def test_large_mesh_image_and_update():
    v0, f = meshes.torus(300, 300)  # 90,000 v / 180,000 f
    v = v0.copy()
    v[:, 2] += 2.0
    cam = (640, 480, 500.0, 500.0, 320.0, 240.0)
    r = Renderer(v, f, *cam, copy=True)
    o = po.OracleRenderer(v, f, *cam)
    assert same_float(r.render_depth(), o.render_depth()).all()
    assert (r.render_mask() == o.render_mask()).all()
    assert (r.render_nn(True) == o.render_nn(True)).all()
    rng = np.random.default_rng(2)
    p = (rng.uniform(-0.1, 1.1, (400_000, 2)) * [640, 480]).astype(np.float32)
    assert same_float(r(p), o(p)).all() and (r.nn(p) == o.nn(p)).all()
    # deformation + update(): vertices_mutable, then the lazily rebuilt 2-D index
    v2 = (v * (1 + 0.05 * np.sin(1.0 + 4 * v[:, 1:2]))).astype(np.float32)
    r.vertices_mutable = v2
    r.update()
    o2 = po.OracleRenderer(v2, f, *cam)
    assert same_float(r(p), o2(p)).all() and (r.contains(p) == o2.contains(p)).all()
    assert (r.nn(p, True) == o2.nn(p, True)).all()
    # analytic sanity: the depth of a hit pixel lies inside the torus' z range
    d = r.render_depth()
    assert d[d > 0].min() >= v2[:, 2].min() - 1e-4 and d.max() <= v2[:, 2].max() + 1e-4


def test_device_tensors_and_copy_false():
    import torch
    g = load_renderer_golden("teapot")
    v = np.ascontiguousarray(g["verts"])
    f = np.ascontiguousarray(g["faces"])
    r = Renderer(v, f, *g["camera"], copy=False)
    o = po.OracleRenderer(v, f, *g["camera"])
    p = g["points2"]
    t = torch.from_numpy(p).cuda()
    d = r(t)
    assert d.is_cuda and same_float(d.cpu().numpy(), o(p)).all()
    assert (r.contains(t).cpu().numpy() == o.contains(p)).all()
    assert (r.nn(t, True).cpu().numpy() == o.nn(p, True)).all()
    assert not r.own_data and r.verts_mutable.shape == (0, 3)
    v[:, 2] += np.float32(0.25)  # copy=False: update() re-reads the caller's memory
    r.update()
    o.update()
    assert same_float(r(p), o(p)).all()
    assert repr(r) == f"<sdf.Renderer(verts.shape=({v.shape[0]}, 3), faces.shape=({f.shape[0]}, 3)>"


def test_edge_cases():
    v = np.array([[0, 0, 2], [1, 0, 2], [0, 1, 2], [0, 0, 0], [1, 1, 3]], np.float32)
    f = np.array([[0, 1, 2], [3, 0, 4]], np.uint32)  # second face has a NaN box: never inserted
    r = Renderer(v, f, 8, 8, 4.0, 4.0, 1.0, 1.0)
    o = po.OracleRenderer(v, f, 8, 8, 4.0, 4.0, 1.0, 1.0)
    assert same_float(r.render_depth(), o.render_depth()).all()
    assert (r.render_mask() == o.render_mask()).all() and (r.render_nn() == o.render_nn()).all()
    assert r(np.zeros((0, 2), np.float32)).shape == (0,)
    assert r([1.5, 1.5])[0] == 2.0
    with pytest.raises(SdfHipError):  # a vertex at z = 0 has no finite projection
        r.render_nn(True)
    # no faces at all: empty image, fill_outside still answers with the nearest projected vertex
    r0 = Renderer(v[[0, 1, 2, 4]], np.zeros((0, 3), np.uint32), 8, 8, 4.0, 4.0, 1.0, 1.0)
    o0 = po.OracleRenderer(v[[0, 1, 2, 4]], np.zeros((0, 3), np.uint32), 8, 8, 4.0, 4.0, 1.0, 1.0)
    assert not r0.render_mask().any() and (r0.render_depth() == 0).all()
    assert (r0.render_nn(True) == o0.render_nn(True)).all()
    # NaN / inf / huge query points terminate and hit nothing
    q = np.array([[np.nan, 1], [np.inf, 1], [1e30, -1e30], [1.5, 1.5]], np.float32)
    assert r.contains(q).tolist() == [False, False, False, True]


def test_vertex_close_to_the_camera_plane_does_not_degrade_the_grid():
    """one face reaching almost to z = 0 projects ~1e6 pixels wide: the grid keeps covering the
    image (renderer.h, rend_update) and results stay exact"""
    import time
    v0, f = meshes.torus(200, 200)
    v = v0.copy()
    v[:, 2] += 2.0
    v = np.concatenate([v, np.array([[0.3, 0.2, 1e-6], [0.31, 0.2, 2.0], [0.3, 0.21, 2.0]], np.float32)])
    f = np.concatenate([f, np.array([[v.shape[0] - 3, v.shape[0] - 2, v.shape[0] - 1]], np.uint32)])
    cam = (320, 240, 250.0, 250.0, 160.0, 120.0)
    r = Renderer(v, f, *cam)
    o = po.OracleRenderer(v, f, *cam)
    assert same_float(r.render_depth(), o.render_depth()).all()
    assert (r.render_nn(True) == o.render_nn(True)).all()
    rng = np.random.default_rng(3)
    p = np.concatenate([rng.uniform(-1, 2, (50000, 2)) * [320, 240], rng.uniform(-1e6, 1e6, (5000, 2))]).astype(np.float32)
    assert same_float(r(p), o(p)).all() and (r.contains(p) == o.contains(p)).all()
    t0 = time.perf_counter()
    r.render_depth()
    assert time.perf_counter() - t0 < 0.05  # one cell for the whole image would take far longer
