"""CPU tests: the oracle (oracle/sdf_oracle.cpp) against the golden vectors generated from the
reference's real vendored nanoflann / RTree (tools/make_golden.py), against those libraries
directly when oracle/_ref is built, and against analytic known answers."""
import numpy as np
import pytest

from conftest import same_float
from oracle import pyoracle as po
from sdf_amd import meshes


def test_nn_matches_real_nanoflann_golden(golden):
    o = po.OracleSDF(golden["verts"], golden["faces"], True, golden["R"])
    assert (o.nn(golden["points"]) == golden["nn"]).all()
    vind, _, _ = o.kd_export()
    assert (vind == golden["vind"]).all()


def test_ray_parity_matches_real_rtree_golden(golden):
    o = po.OracleSDF(golden["verts"], golden["faces"], True, golden["R"])
    det = o.raycast_detail(golden["points"])
    assert (det == golden["ray"]).all()
    assert (o.contains(golden["points"]) == golden["contains_robust"]).all()


def test_brute_force_filter_equals_grid_filter(golden):
    o = po.OracleSDF(golden["verts"], golden["faces"], True, golden["R"])
    pts = golden["points"][::7]
    a = o.raycast_detail(pts)
    o.set_brute(True)
    b = o.raycast_detail(pts)
    assert (a == b).all()


def test_calc_regression_vectors(golden):
    pts = golden["points"]
    o = po.OracleSDF(golden["verts"], golden["faces"], True, golden["R"])
    on = po.OracleSDF(golden["verts"], golden["faces"], False)
    assert same_float(o(pts), golden["sdf_robust"]).all()
    assert same_float(o(pts, True), golden["sdf_robust_trunc"]).all()
    assert same_float(on(pts), golden["sdf_plain"]).all()
    assert same_float(on(pts, True), golden["sdf_plain_trunc"]).all()
    assert (on.contains(pts) == golden["contains_plain"]).all()
    assert same_float(o.face_normals, golden["face_normals"]).all()
    assert same_float(o.face_areas, golden["face_areas"]).all()
    assert (o.aabb == golden["aabb"]).all()


def test_thread_count_does_not_change_results(golden):
    o = po.OracleSDF(golden["verts"], golden["faces"], True, golden["R"])
    pts = golden["points"][:3000]
    assert same_float(o(pts, n_threads=1), o(pts, n_threads=4)).all()


@pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref not built (no /root/reference)")
def test_kd_tree_identical_to_real_nanoflann():
    rng = np.random.default_rng(3)
    for nu in (9, 40, 120):
        v, f = meshes.torus(nu, nu)
        o = po.OracleSDF(v, f, False)
        rk = po.RefKd(v)
        vind, ni, nf = o.kd_export()
        assert (vind == rk.vind()).all()
        rni, rnf = rk.export()
        leaf = ni[:, 0] < 0
        assert (rni[:, 0] == leaf).all()
        assert (rni[leaf][:, 1:] == ni[leaf][:, 2:4]).all()
        assert (rni[~leaf][:, 1] == ni[~leaf][:, 4]).all()
        assert (rnf[~leaf] == nf[~leaf]).all()
        pts = rng.uniform(-1, 1, (20000, 3)).astype(np.float32)
        assert (o.nn(pts) == rk.nn(pts)[0]).all()


@pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref not built (no /root/reference)")
def test_duplicate_and_degenerate_point_sets_match_real_nanoflann():
    rng = np.random.default_rng(5)
    cases = [
        np.zeros((57, 3), np.float32),                                  # all identical
        np.repeat(rng.uniform(-1, 1, (9, 3)).astype(np.float32), 13, 0),  # heavy duplicates
        np.stack([np.linspace(-1, 1, 300), np.zeros(300), np.zeros(300)], 1).astype(np.float32),
        (2.0 ** -np.arange(120))[:, None].astype(np.float32) * np.ones((1, 3), np.float32),
        rng.integers(-3, 4, (500, 3)).astype(np.float32),              # lattice: many ties
        rng.uniform(-1, 1, (7, 3)).astype(np.float32),                  # single leaf
        rng.uniform(-1, 1, (11, 3)).astype(np.float32),                 # smallest split
    ]
    for v in cases:
        f = np.array([[0, 1, 2]], np.uint32)
        o = po.OracleSDF(v, f, False)
        rk = po.RefKd(v)
        assert (o.kd_export()[0] == rk.vind()).all()
        pts = np.concatenate([rng.uniform(-2, 2, (4000, 3)), rng.integers(-3, 4, (4000, 3))]).astype(np.float32)
        assert (o.nn(pts) == rk.nn(pts)[0]).all()


def test_cube_known_answers():
    v, f = meshes.cube(0.5)
    R = meshes.random_rotation(11)
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1, 1, (20000, 3)).astype(np.float32)
    inside = np.all(np.abs(pts) < 0.5, axis=1)
    for robust in (True, False):
        o = po.OracleSDF(v, f, robust, R)
        assert (o.contains(pts) == inside).all()
        s = o(pts)
        assert ((s > 0) == inside).all()
    # the incident-faces-of-nearest-vertex rule is exact for points nearest to a face whose
    # nearest vertex is one of its corners: always true for a cube
    o = po.OracleSDF(v, f, True, R)
    s = o(pts)
    q = np.abs(pts) - 0.5
    exact_out = np.linalg.norm(np.maximum(q, 0), axis=1)
    exact_in = -np.minimum(q.max(1), 0)
    exact = np.where(inside, exact_in, -exact_out)
    assert np.allclose(s, exact, atol=2e-6)


def test_sphere_distance_within_discretisation_error():
    v, f = meshes.icosphere(4, 0.8)
    R = meshes.random_rotation(2)
    o = po.OracleSDF(v, f, True, R)
    rng = np.random.default_rng(1)
    pts = rng.uniform(-1, 1, (20000, 3)).astype(np.float32)
    s = o(pts)
    exact = 0.8 - np.linalg.norm(pts, axis=1)
    assert np.abs(s - exact).max() < 0.01
    far = np.abs(exact) > 0.01
    assert ((s > 0) == (exact > 0))[far].all()


def test_trunc_aabb_quirk_returns_nan_outside_box():
    """sdf.cpp:113-123: the `continue` only continues the t-loop, so an outside point keeps
    min_dist = -FLT_MAX and sqrt() turns it into NaN (SURVEY §8 a1 quirk i)."""
    v, f = meshes.cube(0.5)
    o = po.OracleSDF(v, f, True, meshes.random_rotation(3))
    pts = np.array([[0.9, 0, 0], [0.1, 0.2, 0.05], [0.5, 0.5, 0.5], [0.50001, 0, 0]], np.float32)
    s = o(pts, True)
    assert np.isnan(s[0]) and np.isnan(s[3]) and s[1] > 0 and not np.isnan(s[2])
    on = po.OracleSDF(v, f, False)
    assert (on.contains(pts) == np.array([False, True, True, False])).all()


def test_isolated_vertex_gives_sqrt_fltmax():
    """vertex with no incident face (quirk ii): sqrt(FLT_MAX), sign from the ray parity."""
    v, f = meshes.cube(0.5)
    v = np.concatenate([v, np.array([[3, 3, 3]], np.float32)])
    o = po.OracleSDF(v, f, True, np.eye(3, dtype=np.float32))
    s = o(np.array([[2.9, 3, 3]], np.float32))
    assert s[0] == -np.sqrt(np.float32(np.finfo(np.float32).max))
    on = po.OracleSDF(v, f, False)
    assert on(np.array([[2.9, 3, 3]], np.float32))[0] == np.sqrt(np.float32(np.finfo(np.float32).max))


def test_geometry_kats():
    L = po.lib()
    f = lambda *a: [np.asarray(x, np.float32).ctypes.data_as(po._fp) for x in a]
    # point above the interior of a right triangle: squared height
    assert L.orc_dist_point2tri(*f([0.25, 0.25, 2], [0, 0, 0], [1, 0, 0], [0, 1, 0])) == 4.0
    # beyond vertex b along x: squared distance to b
    assert L.orc_dist_point2tri(*f([3, 0, 0], [0, 0, 0], [1, 0, 0], [0, 1, 0])) == 4.0
    # segment clamp
    assert L.orc_dist_point2lineseg(*f([-2, 0, 0], [0, 0, 0], [1, 0, 0])) == 4.0
    assert L.orc_dist_point2lineseg(*f([0.5, 3, 0], [0, 0, 0], [1, 0, 0])) == 9.0
    # bary2d: edges and vertices are inside (>= 0, sdf.cpp:310)
    assert L.orc_bary2d_inside(*f([0.5, 0], [0, 0], [1, 0], [0, 1])) == 1
    assert L.orc_bary2d_inside(*f([0, 0], [0, 0], [1, 0], [0, 1])) == 1
    assert L.orc_bary2d_inside(*f([0.6, 0.6], [0, 0], [1, 0], [0, 1])) == 0


def test_quaternion_rotation_is_orthonormal():
    R = po.quat_to_R([0.3, -1.2, 0.5, 2.0])
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and np.isclose(np.linalg.det(R), 1, atol=1e-6)
    assert np.allclose(meshes.random_rotation(5) @ meshes.random_rotation(5).T, np.eye(3), atol=1e-6)


def test_generators_are_watertight_and_outward():
    for v, f in (meshes.torus(20, 16), meshes.icosphere(2), meshes.cube()):
        a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
        vol = (a * np.cross(b, c)).sum() / 6
        assert vol > 0
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).astype(np.int64)
        fwd = set(map(tuple, e))
        assert all((y, x) in fwd for x, y in fwd) and len(fwd) == len(e)
    v, f = meshes.torus(424, 424)
    assert v.shape == (179776, 3) and f.shape == (359552, 3)


def test_python_util_helpers_match_the_oracle_geometry():
    """sdf_amd.util (= pysdf.util, pybind.cpp:163-194) against the oracle's restatement of
    sdf.hpp:64-142 on random and degenerate inputs, bit for bit."""
    import ctypes as C
    from sdf_amd import util
    L = po.lib()
    fp = C.POINTER(C.c_float)
    rng = np.random.default_rng(5)
    for k in range(3000):
        p, a, b, c = rng.uniform(-1, 1, (4, 3)).astype(np.float32)
        if k % 7 == 0:
            p = (a + b) / np.float32(2)          # on an edge
        if k % 11 == 0:
            c = a.copy()                         # degenerate triangle -> NaN
        args = [x.ctypes.data_as(fp) for x in (p, a, b, c)]
        got, want = np.float32(util.dist_point2tri(p, a, b, c)), np.float32(L.orc_dist_point2tri(*args))
        assert got.view(np.uint32) == want.view(np.uint32) or (np.isnan(got) and np.isnan(want))
        got, want = np.float32(util.dist_point2lineseg(p, a, b)), np.float32(L.orc_dist_point2lineseg(*args[:3]))
        assert got.view(np.uint32) == want.view(np.uint32) or (np.isnan(got) and np.isnan(want))
        bc = util.bary2d(p[:2], a[:2], b[:2], c[:2])
        p2, a2, b2, c2 = [np.ascontiguousarray(x[:2]) for x in (p, a, b, c)]
        inside = bool(L.orc_bary2d_inside(*[x.ctypes.data_as(fp) for x in (p2, a2, b2, c2)]))
        assert inside == bool((bc >= 0).all())
    # known answers
    assert util.normal([0, 0, 0], [1, 0, 0], [0, 1, 0]).tolist() == [0, 0, 1]
    assert util.dist_point2line([0, 1, 0], [0, 0, 0], [1, 0, 0]) == 1.0
    assert util.dist_point2lineseg([3, 0, 0], [0, 0, 0], [1, 0, 0]) == 4.0
    assert np.allclose(util.bary([0.25, 0.25, 5], [0, 0, 0], [1, 0, 0], [0, 1, 0]), [0.5, 0.25, 0.25])
    assert util.dist_point2tri([0.25, 0.25, 2], [0, 0, 0], [1, 0, 0], [0, 1, 0]) == 4.0
