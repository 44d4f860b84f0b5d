"""sample_surface (sdf.cpp:176-234): CPU checks of the oracle restatement, GPU parity against it.

The reference's samples come from a time-seeded mt19937 (util.cpp:32-40), so nothing can pin the
values against the reference: "parity unpinned".  What is checked: the oracle restatement has the
reference's distribution (points on their face, face frequencies proportional to area), and the
GPU kernel is bit-exact against the oracle given the same counter-based random stream."""
import numpy as np
import pytest

from oracle import pyoracle as po
from sdf_amd import meshes


def _mesh(name):
    if name == "torus":
        return meshes.torus(24, 16)
    if name == "ico":
        return meshes.icosphere(2)
    v, f = meshes.cube()
    return v, f


def _total_area(o):
    s = np.float32(0)
    for a in o.face_areas.astype(np.float32):  # sequential fp32 sum == sdfhip_surface_area
        s = np.float32(s + a)
    return s


@pytest.mark.parametrize("name", ["cube", "torus", "ico"])
def test_oracle_samples_lie_on_faces_with_area_weights(name):
    v, f = _mesh(name)
    o = po.OracleSDF(v, f, False, None)
    n = 200_000
    pts = o.sample_surface(n, 12345, _total_area(o))
    assert pts.shape == (n, 3) and np.isfinite(pts).all()
    # every sample lies in the plane of some face.  (Not necessarily inside that face: the
    # reference mirrors the far half of the parallelogram over the LINE bc, sdf.cpp:222-230,
    # which maps it onto the triangle only when |ab| == |ac|; restated as is.)
    F = f.astype(np.int64)
    nrm, a0 = o.face_normals.astype(np.float64), v[F[:, 0]].astype(np.float64)
    sub = pts[:20_000].astype(np.float64)
    dpl = np.abs(np.einsum("pfk,fk->pf", sub[:, None, :] - a0[None], nrm)).min(1)
    assert dpl.max() < 1e-5
    # distribution: an independent numpy restatement of sdf.cpp:176-234 with numpy's generator
    # must fill coarse spatial histograms like the oracle does (two-sample comparison)
    ref = _numpy_restatement(v, f, o.face_normals, o.face_areas, n, np.random.default_rng(3))
    for axis in range(3):
        lo = min(pts[:, axis].min(), ref[:, axis].min())
        hi = max(pts[:, axis].max(), ref[:, axis].max())
        bins = np.linspace(lo, hi + 1e-6, 7)
        got, _ = np.histogram(pts[:, axis], bins=bins)
        exp, _ = np.histogram(ref[:, axis], bins=bins)
        assert np.all(np.abs(got - exp) < 6 * np.sqrt(got + exp + 1)), (axis, got, exp)


def _numpy_restatement(v, f, normals, areas, n, rng):
    cum = np.cumsum(areas, dtype=np.float32)
    r = rng.uniform(0, 1, n).astype(np.float32) * np.float32(areas.sum(dtype=np.float32))
    fi = np.clip(np.searchsorted(cum, r, side="left"), 0, areas.shape[0] - 1)
    F = f[fi].astype(np.int64)
    a, b, c = v[F[:, 0]], v[F[:, 1]], v[F[:, 2]]
    ab, ac, bc = b - a, c - a, c - b
    perp = np.cross(bc, normals[fi])
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    a_dir = (ab * perp).sum(1) < 0
    u = rng.uniform(0, 1, (n, 1)).astype(np.float32)
    w = rng.uniform(0, 1, (n, 1)).astype(np.float32)
    res = a + u * ab + w * ac
    bp = ((res - b) * perp).sum(1)
    flip = (bp >= 0) != a_dir
    res[flip] -= (bp[flip, None] * perp[flip]) * 2.0
    return res.astype(np.float32)


def test_oracle_sampler_is_deterministic_and_counter_based():
    v, f = _mesh("torus")
    o = po.OracleSDF(v, f, False, None)
    ta = _total_area(o)
    a = o.sample_surface(1000, 7, ta)
    b = o.sample_surface(2000, 7, ta)
    c = o.sample_surface(1000, 8, ta)
    assert np.array_equal(a, b[:1000])  # sample j depends on (seed, j) only
    assert not np.array_equal(a, c)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cube", "torus", "ico"])
def test_gpu_sample_surface_bit_exact_vs_oracle(name):
    import sdf_amd
    v, f = _mesh(name)
    s = sdf_amd.SDF(v, f, robust=False)
    o = po.OracleSDF(v, f, False, None)
    ta = np.float32(s.surface_area)
    assert ta == _total_area(o)
    for n, seed in ((1, 0), (1000, 42), (300_001, 2**63 + 5)):
        got = s.sample_surface(n, seed=seed)
        want = o.sample_surface(n, seed, ta)
        assert got.dtype == np.float32 and got.shape == (n, 3)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # device variant returns the same values in a torch tensor
    t = s.sample_surface(1000, seed=42, device=True)
    assert t.is_cuda and np.array_equal(t.cpu().numpy(), s.sample_surface(1000, seed=42))
    # unseeded calls differ (fresh entropy, as the reference's time-seeded generator)
    assert not np.array_equal(s.sample_surface(100), s.sample_surface(100))
    assert s.sample_surface(0).shape == (0, 3)


@pytest.mark.gpu
def test_gpu_sample_surface_follows_update():
    """the running sum of areas is rebuilt after update() (the reference keeps a stale one)"""
    import sdf_amd
    v, f = _mesh("torus")
    s = sdf_amd.SDF(v, f, robust=False, copy=True)
    s.sample_surface(10, seed=1)
    vm = s.verts_mutable
    vm[:] = vm * np.float32(1.5)
    s.update()
    v2 = (v * np.float32(1.5)).astype(np.float32)
    o = po.OracleSDF(v2, f, False, None)
    got = s.sample_surface(5000, seed=9)
    want = o.sample_surface(5000, 9, np.float32(s.surface_area))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
