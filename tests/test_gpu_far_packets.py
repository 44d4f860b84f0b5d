"""Far packets (kernels_packet.h: k_query_packet_far, option "far_packets"): queries farther from the mesh than twelve
half diagonals of its box (two in a sparse batch) are sorted by distance shell and direction and walked with bounds made for them.  A walk
may only skip what is farther than a lane's threshold, so results must not depend on it: bit-equal to the oracle
(oracle/pyoracle.py <- reference sdf.cpp:110-160 / nanoflann.hpp) and to the same handle with far_packets=0, for
batches that mix near and far queries, shells, directions along the axes, meshes off the origin, lists forced on."""
import numpy as np
import pytest

from conftest import same_float
from oracle import pyoracle as po
from sdf_amd import SDF, meshes

pytestmark = pytest.mark.gpu


def _mixed_queries(rng, c, r, n):
    """near box, shells out to 2^7 far radii and beyond, axis directions, points on shell boundaries"""
    far = 2.0 * r  # (the sparse far radius; the dense one is 12 r: the shells below reach past both)
    parts = [c + rng.uniform(-1.2 * r, 1.2 * r, (n, 3))]
    for k in range(0, 9):
        d = rng.normal(size=(n // 4, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        parts.append(c + d * (far * 2.0 ** k * rng.uniform(1.0, 2.0, (n // 4, 1))))
    axes = np.concatenate([np.eye(3), -np.eye(3)])
    parts.append(c + axes[rng.integers(0, 6, n // 8)] * far * rng.uniform(1, 50, (n // 8, 1)))
    d = rng.normal(size=(n // 8, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    parts.append(c + d * far * (2.0 ** rng.integers(0, 8, (n // 8, 1))))  # on the shell boundaries (up to rounding)
    q = np.concatenate(parts).astype(np.float32)
    return q[rng.permutation(q.shape[0])]


@pytest.mark.parametrize("robust", [True, False])
@pytest.mark.parametrize("scale, shift", [(1.0, (0.0, 0.0, 0.0)), (0.01, (30.0, -20.0, 10.0)), (40.0, (0.0, 500.0, 0.0))])
def test_mixed_near_and_far_batches_equal_the_oracle(scale, shift, robust):
    v, f = meshes.torus(140, 140)
    v = (v * np.float32(scale) + np.asarray(shift, np.float32)).astype(np.float32)
    R = meshes.random_rotation(3)
    lo, hi = v.min(0).astype(np.float64), v.max(0).astype(np.float64)
    c, r = 0.5 * (lo + hi), 0.5 * float(np.linalg.norm(hi - lo))
    q = _mixed_queries(np.random.default_rng(12), c, r, 160_000)
    o = po.OracleSDF(v, f, robust, R if robust else None)
    want_nn, want = o.nn(q), o(q)
    s = SDF(v, f, robust=robust)
    if robust:
        s.set_raycast_axes(R)
    s.set_option("profile", 1)
    s.set_option("scan_max", 0)
    for lists in (0, 1):
        s.set_option("cube_lists", lists)
        assert (s.nn(q) == want_nn).all()
        assert same_float(s(q), want).all()
    assert s.stats["last_heavy_points"] < q.shape[0] // 20  # the far packets are really walked
    s.set_option("far_packets", 0)
    assert (s.nn(q) == want_nn).all()
    assert s.stats["last_heavy_points"] > q.shape[0] // 4     # ... which these lists answered before


def test_far_only_batches_ties_and_small_meshes():
    """a lattice mesh (exact ties among far-field distances are the rule), a 70-vertex mesh (one L1 node, groups
    of a few vertices), queries in one direction only (a packet's axis is its only direction)"""
    rng = np.random.default_rng(5)
    g = np.stack(np.meshgrid(*[np.arange(-5, 6)] * 3, indexing="ij"), -1).reshape(-1, 3)
    v = (g[rng.permutation(g.shape[0])[:1200]] * 0.125).astype(np.float32)
    f = rng.integers(0, v.shape[0], (2400, 3)).astype(np.uint32)
    d = rng.normal(size=(200_000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    q = np.concatenate([np.round(d * 64) * 0.5,                               # lattice points, 32 units out
                        d * rng.uniform(8, 4000, (d.shape[0], 1)),
                        np.outer(rng.uniform(10, 1e4, 30_000), [0.6, 0.0, 0.8])]).astype(np.float32)
    o = po.OracleSDF(v, f, False)
    s = SDF(v, f, robust=False)
    s.set_option("approx", 1)
    s.set_option("scan_max", 0)
    assert (s.nn(q) == o.nn(q)).all()
    assert same_float(s(q), o(q)).all()
    v2, f2 = meshes.icosphere(1, radius=0.5)
    v2 = np.concatenate([v2, v2[:28] * np.float32(0.5)]).astype(np.float32)
    o2 = po.OracleSDF(v2, f2, False)
    s2 = SDF(v2, f2, robust=False)
    s2.set_option("approx", 1)
    s2.set_option("scan_max", 0)
    assert (s2.nn(q) == o2.nn(q)).all()


def test_far_queries_at_absurd_distances_and_non_finite_points():
    """|q - c|^2 overflows from 1.8e19 on: such queries are not far packets (nothing finite to compare), the old
    way answers them; NaN / inf points in between"""
    v, f = meshes.torus(100, 100)
    rng = np.random.default_rng(8)
    d = rng.normal(size=(120_000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    q = (d * 10.0 ** rng.uniform(1.2, 30, (d.shape[0], 1))).astype(np.float32)
    q[::97, 1] = np.nan
    q[5::89, 0] = np.inf
    o = po.OracleSDF(v, f, False)
    s = SDF(v, f, robust=False)
    s.set_option("scan_max", 0)
    assert (s.nn(q) == o.nn(q)).all()
    a, b = s(q), o(q)
    assert (same_float(a, b) | (np.isnan(a) & np.isnan(b))).all()


def test_far_packets_in_fast_nn_mode_follow_its_tie_rule():
    """fast_nn (GPU-built tree; ties go to the lowest tree position): far batches take the far packets too and must
    give what the handle gives without them: the exact fp32 minimum (never farther than the reference's answer;
    nanoflann itself prunes the true minimiser of a far query now and then, by the rounding of its bounds)"""
    import torch
    v, f = meshes.torus(160, 160)
    rng = np.random.default_rng(3)
    lo, hi = v.min(0).astype(np.float64), v.max(0).astype(np.float64)
    q = _mixed_queries(rng, 0.5 * (lo + hi), 0.5 * float(np.linalg.norm(hi - lo)), 200_000)
    s = SDF(v, f, robust=False)
    s.set_option("fast_nn", 1)
    s.set_option("scan_max", 0)
    s.set_option("profile", 1)
    s.update()
    got, d = s.nn(q), s(q)
    assert s.stats["last_heavy_points"] < q.shape[0] // 20
    s.set_option("far_packets", 0)
    assert (s.nn(q) == got).all()
    assert same_float(s(q), d).all()
    o = po.OracleSDF(v, f, False)
    ref = o.nn(q)
    d2 = lambda a, b: ((a[:, 0] - b[:, 0]) ** 2 + (a[:, 1] - b[:, 1]) ** 2) + (a[:, 2] - b[:, 2]) ** 2  # fp32, reference order
    assert (d2(q, v[got]) <= d2(q, v[ref])).all()
    assert np.mean(d2(q, v[got]) == d2(q, v[ref])) > 0.999


def test_far_packets_in_chunked_launches_and_in_a_group():
    """several chunks per call (every chunk sorts, counts its far queries and launches its far kernel anew), with
    30-bit sort keys; and the same batch through a two-member group on one device"""
    import torch
    from sdf_amd import GroupSDF
    v, f = meshes.torus(120, 120)
    R = meshes.random_rotation(9)
    lo, hi = v.min(0).astype(np.float64), v.max(0).astype(np.float64)
    q = _mixed_queries(np.random.default_rng(4), 0.5 * (lo + hi), 0.5 * float(np.linalg.norm(hi - lo)), 100_000)
    o = po.OracleSDF(v, f, True, R)
    want_nn, want = o.nn(q), o(q)
    s = SDF(v, f, robust=True)
    s.set_raycast_axes(R)
    s.set_option("scan_max", 0)
    s.set_option("chunk", 1 << 16)
    s.set_option("sort_bits", 30)
    qt = torch.from_numpy(q).cuda()
    assert (s.nn(qt).cpu().numpy() == want_nn).all()
    assert same_float(s(qt).cpu().numpy(), want).all()
    g = GroupSDF(v, f, devices=[0, 0], robust=True)
    g.set_raycast_axes(R)
    assert (g.nn(q) == want_nn).all()
    assert same_float(g(q), want).all()
