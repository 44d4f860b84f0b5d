"""GPU: small batches take the exact scan (k_query_heavy in scan mode, option "scan_max", the
library default) instead of a tree walk; same bits as the oracle, ties and near ties included."""
import numpy as np
import pytest

from conftest import GOLDEN_NAMES, SCAN_MAX_DEFAULT, load_golden, same_float
from oracle import pyoracle as po
from sdf_amd import SDF, meshes

pytestmark = pytest.mark.gpu


def _scan(s):
    s.set_option("scan_max", SCAN_MAX_DEFAULT)
    s.set_option("profile", 1)
    return s


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_golden_through_the_scan(name):
    g = load_golden(name)
    pts = g["points"]
    s = _scan(SDF(g["verts"], g["faces"], robust=True))
    s.set_raycast_axes(g["R"])
    assert (s.nn(pts) == g["nn"]).all()
    assert s.stats["last_heavy_points"] == pts.shape[0]  # the scan answered the whole batch
    assert same_float(s(pts), g["sdf_robust"]).all()
    assert same_float(s(pts, True), g["sdf_robust_trunc"]).all()
    assert (s.contains(pts) == g["contains_robust"]).all()
    sn = _scan(SDF(g["verts"], g["faces"], robust=False))
    assert same_float(sn(pts), g["sdf_plain"]).all()
    assert (sn.contains(pts) == g["contains_plain"]).all()
    assert (sn.nn(pts[:1]) == g["nn"][:1]).all() and (sn.nn(pts[:5]) == g["nn"][:5]).all()


def test_ties_near_ties_and_deep_trees_through_the_scan():
    rng = np.random.default_rng(11)
    g = np.stack(np.meshgrid(*[np.arange(-6, 7)] * 3, indexing="ij"), -1).reshape(-1, 3)
    v = (g[rng.permutation(g.shape[0])[:1500]] * 0.125).astype(np.float32)
    v = np.concatenate([v, v[:40]])
    f = np.array([[0, 1, 2], [3, 4, 5]], np.uint32)
    q = np.concatenate([(rng.integers(-14, 15, (40000, 3)) * 0.0625), rng.uniform(-1, 1, (20000, 3)),
                        v[rng.integers(0, v.shape[0], 5000)]]).astype(np.float32)
    o = po.OracleSDF(v, f, False)
    s = _scan(SDF(v, f, robust=False))
    assert (s.nn(q) == o.nn(q)).all() and same_float(s(q), o(q)).all()
    # geometric vertex sequence: very deep nanoflann tree (path walks of depth > 40)
    v = np.concatenate([(2.0 ** -np.arange(0, 120, 0.25))[:, None] * np.ones((1, 3)),
                        rng.uniform(-1, 1, (200, 3))]).astype(np.float32)
    s = _scan(SDF(v, np.array([[0, 1, 2]], np.uint32), robust=False))
    o = po.OracleSDF(v, np.array([[0, 1, 2]], np.uint32), False)
    pts = np.concatenate([rng.uniform(-1, 1, (20000, 3)), rng.uniform(0, 1e-3, (20000, 3))]).astype(np.float32)
    assert (s.nn(pts) == o.nn(pts)).all()
    # torus symmetry axis / tube centre circle: hundreds of near-equidistant vertices
    v, f = meshes.torus(200, 200)
    R = meshes.random_rotation(8)
    o = po.OracleSDF(v, f, True, R)
    s = _scan(SDF(v, f))
    s.set_raycast_axes(R)
    n = 20000
    axis = np.stack([rng.normal(0, 1e-6, n), rng.normal(0, 1e-6, n), rng.uniform(-1, 1, n)], 1)
    axis[: n // 4, :2] = 0
    ang = rng.uniform(0, 2 * np.pi, n)
    circ = np.stack([0.7 * np.cos(ang), 0.7 * np.sin(ang), np.zeros(n)], 1) + rng.normal(0, 1e-6, (n, 3))
    q = np.concatenate([axis, circ]).astype(np.float32)
    assert (s.nn(q) == o.nn(q)).all() and same_float(s(q), o(q)).all()
    assert s.stats["last_heavy_points"] == q.shape[0]


def test_scan_threshold_and_fast_mode():
    v, f = meshes.torus(150, 150)
    R = meshes.random_rotation(5)
    o = po.OracleSDF(v, f, True, R)
    s = _scan(SDF(v, f, copy=True))
    s.set_raycast_axes(R)
    rng = np.random.default_rng(1)
    small = rng.uniform(-1, 1, (1000, 3)).astype(np.float32)
    big = rng.uniform(-1, 1, (300_000, 3)).astype(np.float32)   # 300k x 22.5k is past the threshold: tree walk
    assert same_float(s(small), o(small)).all() and s.stats["last_heavy_points"] == 1000
    assert same_float(s(big), o(big)).all() and s.stats["last_heavy_points"] < 300_000
    s.set_option("fast_nn", 1)
    s.update()
    n1 = s.nn(small)
    d = ((small[:, None, :] - v[None]) ** 2).sum(-1)
    assert (np.take_along_axis(d, n1[:, None].astype(np.int64), 1)[:, 0] <= d.min(1) * (1 + 1e-6)).all()
    assert s.stats["last_heavy_points"] == 1000
