"""GPU tests of the optional "fast_nn" mode (the vertex tree built on the GPU,
no host kd build in update()) and of query points no finite arithmetic covers.

fast_nn keeps distances and the minimum exact; what it changes is the DOCUMENTED tie rule: among
the vertices at the (exact fp32) minimum distance the winner is the lowest (30-bit Morton code of
the vertex in the mesh AABB, vertex index) instead of nanoflann's first-visited.  So:
  * the distance of the returned vertex equals the oracle's nearest distance bit for bit;
  * where the minimiser is unique the index equals the oracle's (= the reference's);
  * on exact ties the index follows the rule above (checked by brute force) and the mismatch
    rate against the reference's rule is printed."""
import numpy as np
import pytest

from conftest import load_golden, same_float
from oracle import pyoracle as po
from sdf_amd import SDF, meshes

pytestmark = pytest.mark.gpu


def _d2(q, p):
    """nanoflann's metric association ((dx*dx)+dy*dy)+dz*dz in fp32 (nanoflann.hpp:432-440)"""
    d = (q - p).astype(np.float32)
    return ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)


def _expand10(v):
    v = v.astype(np.uint32) & np.uint32(1023)
    v = (v | (v << np.uint32(16))) & np.uint32(0x030000FF)
    v = (v | (v << np.uint32(8))) & np.uint32(0x0300F00F)
    v = (v | (v << np.uint32(4))) & np.uint32(0x030C30C3)
    v = (v | (v << np.uint32(2))) & np.uint32(0x09249249)
    return v


def _morton30(v):
    lo, hi = v.min(0), v.max(0)
    ext = (hi - lo).astype(np.float32)
    inv = np.where(ext > 0, np.float32(1024.0) / np.where(ext > 0, ext, 1).astype(np.float32), 0).astype(np.float32)
    f = ((v - lo).astype(np.float32) * inv).astype(np.float32)
    c = np.clip(f, 0, 1023).astype(np.uint32)
    return (_expand10(c[:, 0]) << np.uint32(2)) | (_expand10(c[:, 1]) << np.uint32(1)) | _expand10(c[:, 2])


def _fast_rule_bruteforce(v, q):
    """argmin over (distance, Morton code, index), by brute force"""
    order = np.lexsort((np.arange(v.shape[0]), _morton30(v)))  # position -> vertex
    out = np.empty(q.shape[0], np.int32)
    for a in range(0, q.shape[0], 4096):
        d = _d2(q[a:a + 4096, None, :], v[None, order, :])
        out[a:a + 4096] = order[d.argmin(1)]  # argmin: first (lowest position) among equals
    return out


def test_fast_nn_unique_minimisers_match_reference_and_ties_follow_the_rule():
    rng = np.random.default_rng(11)
    g = np.stack(np.meshgrid(*[np.arange(-6, 7)] * 3, indexing="ij"), -1).reshape(-1, 3)
    v = (g[rng.permutation(g.shape[0])[:1500]] * 0.125).astype(np.float32)
    v = np.concatenate([v, v[:40]])  # exact duplicates too
    f = np.array([[0, 1, 2], [3, 4, 5]], np.uint32)
    q = np.concatenate([
        (rng.integers(-14, 15, (40000, 3)) * 0.0625),
        rng.uniform(-1, 1, (20000, 3)),
        v[rng.integers(0, v.shape[0], 5000)],
    ]).astype(np.float32)
    o = po.OracleSDF(v, f, False)
    ref = o.nn(q)
    s = SDF(v, f, robust=False)
    s.set_option("fast_nn", 1)
    s.update()
    want = _fast_rule_bruteforce(v, q)
    for batch in (q, q[:3000]):  # sorted packet launch / small unsorted launch
        got = s.nn(batch)
        assert (got == want[: batch.shape[0]]).all()
        assert (_d2(batch, v[got]) == _d2(batch, v[ref[: batch.shape[0]]])).all()
    s.set_option("sort_queries", 0)
    assert (s.nn(q) == want).all()
    got = s.nn(q)
    print(f"\nfast_nn on a tie-heavy lattice: {np.mean(got != ref) * 100:.2f} % of queries pick a different "
          f"(equidistant) vertex than the reference's rule")
    # back to strict mode on the same handle
    s.set_option("fast_nn", 0)
    s.set_option("sort_queries", 1)
    s.update()
    assert (s.nn(q) == ref).all()


@pytest.mark.parametrize("name", ["torus150", "teapot"])
def test_fast_nn_sdf_contains_on_meshes(name):
    if name == "torus150":
        v, f = meshes.torus(150, 150)
    else:
        g = load_golden("teapot")
        v, f = g["verts"], g["faces"]
    R = meshes.random_rotation(77)
    rng = np.random.default_rng(4)
    pts = np.concatenate([rng.uniform(-1, 1, (150000, 3)), meshes.grid_points(40)]).astype(np.float32)
    o = po.OracleSDF(v, f, True, R)
    s = SDF(v, f, robust=True, copy=True)
    s.set_raycast_axes(R)
    s.set_option("fast_nn", 1)
    s.update()
    ref = o.nn(pts)
    got = s.nn(pts)
    dg, dr = _d2(pts, v[got]), _d2(pts, v[ref])
    assert (dg == dr).all()  # always a true nearest vertex, at exactly the same fp32 distance
    same = got == ref
    print(f"\nfast_nn {name}: nn differs from the reference's tie rule on {np.mean(~same) * 100:.4f} % of queries")
    assert (got[:20000] == _fast_rule_bruteforce(v, pts[:20000])).all()
    # contains never looks at the vertex tree; sdf is the reference's wherever nn is
    assert (s.contains(pts) == o.contains(pts)).all()
    d, od = s(pts), o(pts)
    assert same_float(d[same], od[same]).all()
    assert ((d > 0) == (od > 0))[~same].all()
    # deformation + update in fast mode
    v2 = (v * (1 + 0.05 * np.sin(1.0 + 4 * v[:, 1:2]))).astype(np.float32)
    s.vertices_mutable = v2
    s.update()
    o2 = po.OracleSDF(v2, f, True, R)
    got2, ref2 = s.nn(pts), o2.nn(pts)
    assert (_d2(pts, v2[got2]) == _d2(pts, v2[ref2])).all()
    assert (s.contains(pts) == o2.contains(pts)).all()


@pytest.mark.parametrize("fast", [0, 1])
def test_non_finite_and_huge_query_points_terminate(fast):
    """inf / NaN / 1e30 coordinates (every distance overflows): undefined in the reference
    (sdf.hpp:152-153) but the walk must end, and finite points of the same batch are unaffected."""
    v, f = meshes.torus(83, 83)
    R = meshes.random_rotation(3)
    o = po.OracleSDF(v, f, True, R)
    s = SDF(v, f, robust=True, copy=True)
    s.set_raycast_axes(R)
    if fast:
        s.set_option("fast_nn", 1)
        s.update()
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1, 1, (40000, 3)).astype(np.float32)
    bad = rng.choice(40000, 600, replace=False)
    pts[bad[:200]] = np.float32(1e30) * rng.choice([-1, 1], (200, 3))
    pts[bad[200:400], rng.integers(0, 3, 200)] = np.inf
    pts[bad[400:], rng.integers(0, 3, 200)] = np.nan
    ok = np.ones(40000, bool)
    ok[bad] = False
    for batch_ok, batch in ((ok, pts), (ok[:2000], pts[:2000])):
        n = s.nn(batch)
        c = s.contains(batch)
        d = s(batch)
        fin = batch[batch_ok]
        if fast:
            assert (_d2(fin, v[n[batch_ok]]) == _d2(fin, v[o.nn(fin)])).all()
        else:
            assert (n[batch_ok] == o.nn(fin)).all()
            assert same_float(d[batch_ok], o(fin)).all()
        assert (c[batch_ok] == o.contains(fin)).all()
        assert ((n >= -1) & (n < v.shape[0])).all()
        assert not c[~batch_ok][np.isfinite(batch[~batch_ok]).all(1)].any()  # 1e30: outside the AABB


def test_fast_nn_heavy_wave_path_forced():
    g = load_golden("teapot")
    v, f = g["verts"], g["faces"]
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.uniform(-1, 1, (60000, 3)), meshes.grid_points(24)]).astype(np.float32)
    s = SDF(v, f, robust=False)
    s.set_option("fast_nn", 1)
    s.update()
    want = s.nn(pts)
    assert (want[:15000] == _fast_rule_bruteforce(v, pts[:15000])).all()
    d = s(pts)
    s.set_option("heavy_budget", 1)
    s.set_option("profile", 1)
    assert (s.nn(pts) == want).all()
    assert s.stats["last_heavy_points"] > pts.shape[0] // 2
    assert same_float(s(pts), d).all()


def test_fast_nn_on_then_off_on_a_tiny_mesh_with_moved_vertices():
    """the approximate pass's vertex copy has a capacity of its own.  A fast_nn update may grow the
    shared position capacity past it; back in strict mode with moved vertices (a differently padded tree) the
    fill must not run past the older, smaller array."""
    import sdf_amd
    from oracle import pyoracle as po
    from sdf_amd import meshes
    from conftest import same_float
    v, f = meshes.icosphere(1)  # 42 vertices
    rng = np.random.default_rng(0)
    s = sdf_amd.SDF(v, f, robust=True, copy=True)
    R = meshes.random_rotation(2)
    s.set_raycast_axes(R)
    s.set_option("approx", 1)
    pts = rng.uniform(-1.5, 1.5, (20_000, 3)).astype(np.float32)
    for rnd in range(6):
        s.set_option("fast_nn", 1)
        s.update()
        s.nn(pts)
        # clustered vertices: many small kd leaves -> more padded positions than before
        vv = v.copy()
        vv[: 20 + 3 * rnd] *= np.float32(1.0 + 0.3 * rng.random())
        vv += rng.normal(0, 1e-3, vv.shape).astype(np.float32)
        s.vertices_mutable = vv
        s.set_option("fast_nn", 0)
        s.update()
        o = po.OracleSDF(vv, f, True, R)
        assert (s.nn(pts) == o.nn(pts)).all()
        assert same_float(s(pts), o(pts)).all()
