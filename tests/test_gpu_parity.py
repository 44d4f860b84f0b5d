"""GPU parity tests (run on the MI355X box with -m gpu): the HIP path, called through the C ABI,
against the oracle on the same seeded inputs and against the committed golden vectors.
Bar: nn / contains / per-ray parities bit-exact; SDF bit-exact too (NaN == NaN), which is
stricter than the 1e-5 relative tolerance BASELINE.json states."""
import numpy as np
import pytest

from conftest import GOLDEN_NAMES, load_golden, same_float
from oracle import pyoracle as po
from sdf_amd import SDF, meshes

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["1", "0"], ids=["approx", "exact"])
def packet_walk(request, monkeypatch):
    """every test of this module runs with both walks of the packet pass: the approximate first
    pass + exact second phase (forced on, whatever the automatic criterion says) and the exact walk"""
    monkeypatch.setenv("SDFHIP_APPROX", request.param)
    return request.param


@pytest.fixture(scope="module", params=GOLDEN_NAMES)
def gold(request):
    g = load_golden(request.param)
    g["name"] = request.param
    return g


def test_nn_golden(gold):
    s = SDF(gold["verts"], gold["faces"], robust=False)
    assert (s.nn(gold["points"]) == gold["nn"]).all()


def test_ray_parities_and_contains_golden(gold):
    s = SDF(gold["verts"], gold["faces"], robust=True)
    s.set_raycast_axes(gold["R"])
    assert (s.raycast_detail(gold["points"]) == gold["ray"]).all()
    assert (s.contains(gold["points"]) == gold["contains_robust"]).all()


def test_sdf_golden(gold):
    pts = gold["points"]
    s = SDF(gold["verts"], gold["faces"], robust=True)
    s.set_raycast_axes(gold["R"])
    assert same_float(s(pts), gold["sdf_robust"]).all()
    assert same_float(s(pts, True), gold["sdf_robust_trunc"]).all()
    sn = SDF(gold["verts"], gold["faces"], robust=False)
    assert same_float(sn(pts), gold["sdf_plain"]).all()
    assert same_float(sn(pts, True), gold["sdf_plain_trunc"]).all()
    assert (sn.contains(pts) == gold["contains_plain"]).all()


def test_precompute_golden(gold):
    s = SDF(gold["verts"], gold["faces"], robust=True)
    assert same_float(s.face_normals, gold["face_normals"]).all()
    assert same_float(s.face_areas, gold["face_areas"]).all()
    assert (s.aabb == gold["aabb"]).all()
    assert np.isclose(s.surface_area, gold["face_areas"].astype(np.float64).sum(), rtol=1e-4)


@pytest.mark.parametrize("sort", [0, 1])
def test_oracle_parity_medium_mesh_sorted_and_unsorted(sort):
    v, f = meshes.torus(150, 150)  # 22,500 v / 45,000 f
    R = meshes.random_rotation(77)
    rng = np.random.default_rng(4)
    pts = np.concatenate([rng.uniform(-1, 1, (150000, 3)), meshes.grid_points(40)]).astype(np.float32)
    o = po.OracleSDF(v, f, True, R)
    s = SDF(v, f, robust=True)
    s.set_raycast_axes(R)
    s.set_option("sort_queries", sort)
    s.set_option("sort_min_contains", 0)  # robust contains through the packet kernel too (default: >= 12M points)
    assert (s.nn(pts) == o.nn(pts)).all()
    assert (s.contains(pts) == o.contains(pts)).all()
    assert same_float(s(pts), o(pts)).all()
    assert (s.raycast_detail(pts) == o.raycast_detail(pts)).all()


def test_update_after_vertex_mutation():
    v, f = meshes.torus(60, 60)
    R = meshes.random_rotation(5)
    s = SDF(v, f, robust=True, copy=True)
    s.set_raycast_axes(R)
    rng = np.random.default_rng(9)
    pts = rng.uniform(-1, 1, (60000, 3)).astype(np.float32)
    for t in range(3):
        v2 = (v * (1 + 0.05 * np.sin(2 * np.pi * t / 3 + 4 * v[:, 1:2]))).astype(np.float32)
        s.vertices_mutable = v2
        s.update()
        o = po.OracleSDF(v2, f, True, R)
        assert (s.nn(pts) == o.nn(pts)).all()
        assert (s.contains(pts) == o.contains(pts)).all()
        assert same_float(s(pts), o(pts)).all()


def test_copy_false_rereads_caller_memory_on_update():
    v, f = meshes.torus(30, 30)
    v = np.ascontiguousarray(v)
    s = SDF(v, f, robust=False, copy=False)
    pts = np.random.default_rng(1).uniform(-1, 1, (5000, 3)).astype(np.float32)
    v *= np.float32(0.5)
    s.update()
    o = po.OracleSDF(v, f, False)
    assert (s.nn(pts) == o.nn(pts)).all() and same_float(s(pts), o(pts)).all()
    assert s.verts_mutable.shape == (0, 3) and not s.own_data


def test_edge_cases():
    v, f = meshes.cube(0.5)
    s = SDF(v, f)
    # empty batch, single point as a length-3 list (README.md:103-106), float64 input
    assert s(np.zeros((0, 3), np.float32)).shape == (0,)
    assert s([0.1, 0.2, 0.05]).shape == (1,) and s([0.1, 0.2, 0.05])[0] > 0
    assert s.contains(np.array([[0.1, 0.2, 0.3], [2, 2, 2]], np.float64)).tolist() == [True, False]
    # trunc_aabb quirk: NaN outside the box (sdf.cpp:113-123 falls through)
    r = s(np.array([[0.9, 0, 0], [0.1, 0.2, 0.05]], np.float32), True)
    assert np.isnan(r[0]) and r[1] > 0
    # isolated vertex: sqrt(FLT_MAX)
    v2 = np.concatenate([v, np.array([[3, 3, 3]], np.float32)])
    s2 = SDF(v2, f, robust=False)
    assert s2(np.array([[2.9, 3, 3]], np.float32))[0] == np.sqrt(np.float32(np.finfo(np.float32).max))
    assert s2.nn(np.array([[2.9, 3, 3]], np.float32))[0] == 8
    # tiny meshes: single leaf kd-tree, 1-face BVH
    v3 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    f3 = np.array([[0, 1, 2]], np.uint32)
    o3 = po.OracleSDF(v3, f3, True, np.eye(3, dtype=np.float32))
    s3 = SDF(v3, f3)
    s3.set_raycast_axes(np.eye(3, dtype=np.float32))
    p = np.random.default_rng(0).uniform(-1, 2, (3000, 3)).astype(np.float32)
    assert same_float(s3(p), o3(p)).all() and (s3.nn(p) == o3.nn(p)).all()
    assert repr(s) == "<sdf.SDF(verts.shape=(8, 3), faces.shape=(12, 3), robust=True)>"
    assert s.adj_faces(0) == po.OracleSDF(v, f, True).adj_faces(0).tolist()


def test_ray_grid_big_faces_and_coarsening():
    """column grid of the parity rays: faces spanning many cells (the "big" list), a soup of
    large overlapping faces (the grid coarsens itself) and points outside the grid's range"""
    rng = np.random.default_rng(11)
    R = meshes.random_rotation(5)
    # (a) fine sphere inside a huge box: 12 faces cover every cell
    vs, fs = meshes.icosphere(4, radius=0.5)
    vb, fb = meshes.cube(0.9)
    v = np.concatenate([vs, vb]).astype(np.float32)
    f = np.concatenate([fs, fb + vs.shape[0]]).astype(np.uint32)
    # (b) 3000 large random triangles, not watertight
    v2 = rng.uniform(-1, 1, (9000, 3)).astype(np.float32)
    f2 = np.arange(9000, dtype=np.uint32).reshape(3000, 3)
    for vv, ff in ((v, f), (v2, f2)):
        o = po.OracleSDF(vv, ff, True, R)
        s = SDF(vv, ff)
        s.set_raycast_axes(R)
        p = rng.uniform(-1.2, 1.2, (150_000, 3)).astype(np.float32)
        p[:2000] = vv[rng.integers(0, vv.shape[0], 2000)]  # on vertices: edge / vertex hits
        assert (s.raycast_detail(p) == o.raycast_detail(p)).all()
        assert (s.contains(p) == o.contains(p)).all()
        assert same_float(s(p), o(p)).all()
        s.set_option("sort_min_contains", 0)
        for opt in (0, 1):  # strict and packet kernels share the grid
            s.set_option("sort_queries", opt)
            assert (s.contains(p[:20000]) == o.contains(p[:20000])).all()
        for mode in (1, 2, 0):  # packet kernel: column grid / cooperative LBVH walk / automatic
            s.set_option("ray_mode", mode)
            assert (s.contains(p) == o.contains(p)).all()
            assert same_float(s(p[:50000]), o(p[:50000])).all()


@pytest.mark.parametrize("x_grid", [1, 0])
def test_open_meshes_x_tiebreak_ray(x_grid):
    """z+ and z- disagree wherever a mesh is not closed, and the x+ ray decides (sdf.cpp:326): on its own
    column grid over (y', z') (default) or on the LBVH (x_grid = 0), same parities as the oracle per ray"""
    rng = np.random.default_rng(31)
    R = meshes.random_rotation(6)
    vt, ft = meshes.torus(90, 90)
    ft = ft[rng.uniform(size=ft.shape[0]) > 0.3]  # a torus with 30 % of its faces missing
    vn, fn = meshes.icosphere(5, radius=0.7, noise=0.05, seed=2)
    fn = fn[rng.uniform(size=fn.shape[0]) > 0.1]  # a rough sphere with holes
    for v, f in ((vt, ft), (vn, fn)):
        f = np.ascontiguousarray(f)
        o = po.OracleSDF(v, f, True, R)
        s = SDF(v, f)
        s.set_option("x_grid", x_grid)
        s.set_raycast_axes(R)
        p = rng.uniform(-1, 1, (120_000, 3)).astype(np.float32)
        p[:1500] = v[rng.integers(0, v.shape[0], 1500)]
        want = o.raycast_detail(p)
        assert (want[:, 2] >= 0).sum() > 3000  # the x+ ray really is cast
        assert (s.raycast_detail(p) == want).all()
        s.set_option("sort_min_contains", 0)
        assert (s.contains(p) == o.contains(p)).all()
        assert same_float(s(p), o(p)).all()


def test_degenerate_vertex_sets_use_deep_trees():
    """geometric vertex sequence -> very deep nanoflann tree -> global-memory stack variant"""
    rng = np.random.default_rng(2)
    v = np.concatenate([(2.0 ** -np.arange(0, 120, 0.25))[:, None] * np.ones((1, 3)),
                        rng.uniform(-1, 1, (200, 3))]).astype(np.float32)
    f = np.array([[0, 1, 2]], np.uint32)
    s = SDF(v, f, robust=False)
    o = po.OracleSDF(v, f, False)
    assert s.stats["kd_depth"] == o.kd_max_depth()
    pts = np.concatenate([rng.uniform(-1, 1, (20000, 3)), rng.uniform(0, 1e-3, (20000, 3))]).astype(np.float32)
    assert (s.nn(pts) == o.nn(pts)).all()


def test_all_vertices_equal_and_large_parallel_build():
    """(a) every vertex identical: all spans are zero, no split dimension qualifies and nanoflann clamps
    the cut to the elements of dimension 0 (its second computeMinMax) -> halving splits; every query
    is a many-way exact tie.  (b) a vertex set large enough for the data-parallel top of the host
    build (>= 20000): the tree must still be nanoflann's."""
    rng = np.random.default_rng(21)
    v = np.tile(np.array([[0.25, -0.5, 0.125]], np.float32), (300, 1))
    f = np.array([[0, 1, 2]], np.uint32)
    q = rng.uniform(-1, 1, (20000, 3)).astype(np.float32)
    s, o = SDF(v, f, robust=False), po.OracleSDF(v, f, False)
    assert s.stats["kd_depth"] == o.kd_max_depth()
    assert (s.nn(q) == o.nn(q)).all()
    s.set_option("sort_queries", 0)
    assert (s.nn(q) == o.nn(q)).all()
    # (b) lattice with many equal coordinates (equal keys in every partition), shuffled
    g = np.stack(np.meshgrid(*[np.arange(32)] * 3, indexing="ij"), -1).reshape(-1, 3)
    v = (g[rng.permutation(g.shape[0])] * 0.0625 - 1.0).astype(np.float32)
    s, o = SDF(v, f, robust=False), po.OracleSDF(v, f, False)
    assert s.stats["kd_depth"] == o.kd_max_depth()
    q = np.concatenate([rng.uniform(-1.2, 1.2, (60000, 3)), rng.integers(-40, 40, (20000, 3)) * 0.03125]).astype(np.float32)
    assert (s.nn(q) == o.nn(q)).all()


def test_tie_heavy_lattices_packet_and_strict_paths():
    """integer-lattice vertices + half-integer / on-vertex queries: up to 8-way exact distance
    ties, resolved by nanoflann's visiting order.  Large batch -> packet pass (+ tie tracking,
    path-bound check, deferred strict fallback); small batch / sort off -> strict kernel."""
    rng = np.random.default_rng(11)
    g = np.stack(np.meshgrid(*[np.arange(-6, 7)] * 3, indexing="ij"), -1).reshape(-1, 3)
    v = (g[rng.permutation(g.shape[0])[:1500]] * 0.125).astype(np.float32)
    v = np.concatenate([v, v[:40]])  # exact duplicates too
    f = np.array([[0, 1, 2], [3, 4, 5]], np.uint32)
    o = po.OracleSDF(v, f, False)
    q = np.concatenate([
        (rng.integers(-14, 15, (40000, 3)) * 0.0625),          # lattice + half-lattice points
        rng.uniform(-1, 1, (20000, 3)),
        v[rng.integers(0, v.shape[0], 5000)],                  # exactly on vertices
    ]).astype(np.float32)
    ref = o.nn(q)
    d = ((v[:, None, :] - q[None, :2000, :]) ** 2).sum(-1)
    assert ((d == d.min(0, keepdims=True)).sum(0) > 1).mean() > 0.2  # plenty of exact ties
    s = SDF(v, f, robust=False)
    s.set_option("profile", 1)
    assert (s.nn(q) == ref).all()
    assert same_float(s(q), o(q)).all()
    s.set_option("packet", 0)
    assert (s.nn(q) == ref).all()
    s.set_option("sort_queries", 0)
    assert (s.nn(q) == ref).all()
    assert (s.nn(q[:5000]) == ref[:5000]).all()


def test_near_tie_queries_on_symmetry_axes():
    """queries on / near the torus axis and tube-centre circle: hundreds of near-equidistant
    vertices (the deferred-fallback and path-bound cases of the packet pass)."""
    v, f = meshes.torus(200, 200)
    R = meshes.random_rotation(8)
    o = po.OracleSDF(v, f, True, R)
    s = SDF(v, f)
    s.set_raycast_axes(R)
    rng = np.random.default_rng(3)
    n = 60000
    z = rng.uniform(-1, 1, n)
    axis = np.stack([rng.normal(0, 1e-6, n), rng.normal(0, 1e-6, n), z], 1)
    axis[: n // 4, :2] = 0
    ang = rng.uniform(0, 2 * np.pi, n)
    circ = np.stack([0.7 * np.cos(ang), 0.7 * np.sin(ang), np.zeros(n)], 1) + rng.normal(0, 1e-6, (n, 3))
    q = np.concatenate([axis, circ]).astype(np.float32)
    assert (s.nn(q) == o.nn(q)).all()
    assert same_float(s(q), o(q)).all()
    assert (s.contains(q) == o.contains(q)).all()


def test_device_pointer_entry_points_with_torch():
    import torch
    v, f = meshes.torus(83, 83)
    R = meshes.random_rotation(3)
    s = SDF(v, f)
    s.set_raycast_axes(R)
    o = po.OracleSDF(v, f, True, R)
    g = torch.Generator(device="cuda").manual_seed(7)
    pts = (torch.rand((200000, 3), device="cuda", generator=g) * 2 - 1)
    out = s(pts)
    assert out.is_cuda and out.dtype == torch.float32
    h = pts.cpu().numpy()
    assert same_float(out.cpu().numpy(), o(h)).all()
    assert (s.contains(pts).cpu().numpy() == o.contains(h)).all()
    assert (s.nn(pts).cpu().numpy() == o.nn(h)).all()


def test_dlpack_device_arrays_reach_the_device_entry_points():
    """a non-torch DLPack producer (stand-in: an object exposing only __dlpack__ / __dlpack_device__) holding
    device memory is answered on the device, without a host round trip, and equals the numpy path"""
    import torch

    class Capsule:  # what CuPy / JAX arrays look like to a consumer
        def __init__(self, t):
            self._t = t

        def __dlpack__(self, stream=None, **kw):
            return self._t.__dlpack__()

        def __dlpack_device__(self):
            return self._t.__dlpack_device__()

    v, f = meshes.torus(60, 60)
    s = SDF(v, f)
    s.set_raycast_axes(meshes.random_rotation(2))
    pts = np.random.default_rng(4).uniform(-1, 1, (60_000, 3)).astype(np.float32)
    t = torch.from_numpy(pts).cuda()
    d, c, n = s(Capsule(t)), s.contains(Capsule(t)), s.nn(Capsule(t))
    assert d.is_cuda and c.is_cuda and n.is_cuda
    assert same_float(d.cpu().numpy(), s(pts)).all()
    assert (c.cpu().numpy() == s.contains(pts)).all() and (n.cpu().numpy() == s.nn(pts)).all()
    back = torch.from_dlpack(d)  # the result is itself a DLPack producer
    assert back.data_ptr() == d.data_ptr()
    h = s(Capsule(torch.from_numpy(pts)))  # a host DLPack array takes the numpy path
    assert isinstance(h, np.ndarray) and same_float(h, s(pts)).all()


def test_async_queries_on_two_streams_and_update_do_not_race():
    """`_device` calls are asynchronous on the caller's stream and share the handle's workspace:
    a call on a second stream and update() must be ordered after the kernels of the call before."""
    import torch
    v, f = meshes.torus(150, 150)
    R = meshes.random_rotation(5)
    s = SDF(v, f)
    s.set_raycast_axes(R)
    g = torch.Generator(device="cuda").manual_seed(9)
    pa = torch.rand((3_000_000, 3), device="cuda", generator=g) * 2 - 1
    pb = torch.rand((3_000_000, 3), device="cuda", generator=g) * 2 - 1
    ref_a, ref_b = s(pa).clone(), s.nn(pb).clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(sa):
            oa = s(pa)
        with torch.cuda.stream(sb):
            ob = s.nn(pb)
        torch.cuda.synchronize()
        assert same_float(oa.cpu().numpy(), ref_a.cpu().numpy()).all()
        assert (ob == ref_b).all()
    # an update right behind an asynchronous query must not change what that query reads
    v2 = (v * 0.9).astype(np.float32)
    with torch.cuda.stream(sa):
        oa = s(pa)
    s.vertices_mutable = v2
    s.update()
    torch.cuda.synchronize()
    assert same_float(oa.cpu().numpy(), ref_a.cpu().numpy()).all()
    o2 = po.OracleSDF(v2, f, True, R)
    h = pa[:100000].cpu().numpy()
    assert same_float(s(pa[:100000].contiguous()).cpu().numpy(), o2(h)).all()


@pytest.mark.parametrize("offset", [(100.0, -50.0, 25.0), (0.7, 0.0, -3.0), (1e4, 1e4, -1e4)])
def test_mesh_far_from_the_origin(offset, packet_walk):
    """a() = |v|^2 - 2 q.v loses its resolution when the mesh sits far from the origin against its size: the
    approximate pass then works on fl(v - c), fl(q - c) around the centre c of the mesh (two more roundings in
    its error model) and stays in use by the automatic criterion; at 1e4 the fp32 grid itself is as coarse as
    the mesh's edges (duplicate vertices, ties)."""
    v, f = meshes.torus(60, 60)
    v = (v + np.array(offset)).astype(np.float32)
    R = meshes.random_rotation(12)
    rng = np.random.default_rng(13)
    pts = (rng.uniform(-1.2, 1.2, (30000, 3)) + np.array(offset)).astype(np.float32)
    pts[:500] = v[rng.integers(0, v.shape[0], 500)]
    s = SDF(v, f)
    s.set_raycast_axes(R)
    o = po.OracleSDF(v, f, True, R)
    assert (s.nn(pts) == o.nn(pts)).all()
    assert same_float(s(pts), o(pts)).all()
    if packet_walk == "1" and offset[0] < 1e3:  # the approximate pass settles (nearly) everything itself
        s.set_option("profile", 1)
        s.nn(pts)
        assert s.stats["last_heavy_points"] < 300


def test_sampled_query_bounds_with_outliers():
    """Batches of >= 8M points take the Morton bounds of the query sort from a strided sample; points
    outside them (here: far outliers and non-finite points in runs the sample skips) are clamped into
    the border cells, which may cost coherence but never a result."""
    import torch
    v, f = meshes.torus(83, 83)
    R = meshes.random_rotation(8)
    s = SDF(v, f)
    s.set_raycast_axes(R)
    n = 9_000_000
    g = torch.Generator(device="cuda").manual_seed(31)
    pts = torch.rand((n, 3), device="cuda", generator=g) * 2 - 1
    special = torch.tensor([[50.0, -50.0, 50.0], [-1e6, 3.0, 0.0], [float("nan"), 0.0, 0.0],
                            [0.0, float("inf"), 0.0], [7.0, 7.0, -7.0]], device="cuda")
    idx = torch.tensor([300, 777, 5000, 4_000_300, n - 3], device="cuda")  # runs 1, 3, 19, ...: not sampled
    pts[idx] = special
    d, c, nn = s(pts), s.contains(pts), s.nn(pts)
    sub = torch.cat([idx, torch.randint(0, n, (30000,), device="cuda", generator=g)])
    h = pts[sub].cpu().numpy()
    o = po.OracleSDF(v, f, True, R)
    assert same_float(d[sub].cpu().numpy(), o(h)).all()
    assert (c[sub].cpu().numpy() == o.contains(h)).all()
    assert (nn[sub].cpu().numpy() == o.nn(h)).all()


@pytest.mark.parametrize("start", [1, 2, 3])
def test_unaligned_device_points_keep_the_query_sort_coherent(start):
    """A slice such as pts[1:] (or a shard that starts at an odd point) is not 16-byte aligned.  The sampled
    bounds of the query sort must still see it: with empty bounds every Morton key is 0, the sort is the
    identity and nearly every query ends in the brute-force scan (correct, but seconds instead of ms)."""
    import torch
    v, f = meshes.torus(83, 83)
    s = SDF(v, f)
    s.set_raycast_axes(meshes.random_rotation(8))
    s.set_option("profile", 1)  # the heavy-list length is read back only while profiling
    n = 8_500_000
    g = torch.Generator(device="cuda").manual_seed(5)
    base = torch.rand((n + 4, 3), device="cuda", generator=g) * 2 - 1
    sl = base[start:start + n]
    assert sl.data_ptr() % 16 != 0 and sl.is_contiguous()
    al = sl.clone()
    assert al.data_ptr() % 16 == 0
    for fn in (s.nn, s.__call__):
        a = fn(al)
        heavy_al = s.stats["last_heavy_points"]
        b = fn(sl)
        heavy_sl = s.stats["last_heavy_points"]
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        assert heavy_sl <= max(2 * heavy_al, n // 1000), (heavy_al, heavy_sl)


def test_large_mesh_properties_full_size():
    """BASELINE config-4 scale mesh: size-independent properties + a sampled oracle check."""
    v, f = meshes.torus(424, 424)
    R = meshes.random_rotation(1234)
    s = SDF(v, f)
    s.set_raycast_axes(R)
    rng = np.random.default_rng(7)
    pts = rng.uniform(-1, 1, (2_000_000, 3)).astype(np.float32)
    d = s(pts)
    c = s.contains(pts)
    n = s.nn(pts)
    # sign agrees with contains; nn is the true nearest vertex; |sdf| <= dist to nearest vertex
    assert ((d > 0) == c)[d != 0].all()
    dn = np.linalg.norm(v[n] - pts, axis=1)
    assert (np.abs(d) <= dn * (1 + 1e-6)).all()
    # analytic torus: inside iff (sqrt(x^2+y^2)-R)^2 + z^2 < r^2, away from the surface
    rho = np.sqrt(pts[:, 0] ** 2 + pts[:, 1] ** 2) - 0.7
    ana = 0.24 - np.sqrt(rho ** 2 + pts[:, 2] ** 2)
    far = np.abs(ana) > 2e-3
    assert ((ana > 0) == c)[far].all()
    assert np.abs(d - ana).max() < 2e-3
    # idempotence + sharding invariance (the multi-GPU partition is by index range)
    assert same_float(s(pts[:500000]), d[:500000]).all()
    k = 300000
    o = po.OracleSDF(v, f, True, R)
    assert (n[:k] == o.nn(pts[:k])).all()
    assert (c[:k] == o.contains(pts[:k])).all()
    assert same_float(d[:k], o(pts[:k])).all()


def test_cpp_facade_runs_on_gpu(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "facade_smoke")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "facade_smoke.cpp"),
                           "-L" + os.path.join(root, "sdf_amd"), "-lsdfhip",
                           "-Wl,-rpath," + os.path.join(root, "sdf_amd"), "-Wl,-rpath,/opt/rocm/lib",
                           "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_robust_handle_without_faces():
    """vertices only (nf = 0), robust: the parity rays find an empty grid cell; every vertex is
    isolated -> sqrt(FLT_MAX) with the ray sign (sdf.cpp:146-148)"""
    rng = np.random.default_rng(4)
    v = rng.uniform(-0.5, 0.5, (300, 3)).astype(np.float32)
    f = np.zeros((0, 3), np.uint32)
    R = meshes.random_rotation(1)
    s = SDF(v, f, robust=True)
    s.set_raycast_axes(R)
    o = po.OracleSDF(v, f, True, R)
    for n in (2000, 20000):  # strict kernel / packet kernel
        p = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        assert not s.contains(p).any()
        assert (s.nn(p) == o.nn(p)).all() and same_float(s(p), o(p)).all()


@pytest.mark.parametrize("budget", [1, 40])
def test_heavy_wave_path_forced(budget):
    """tail limiter of the packet pass: with a tiny work budget every wave (budget 1) or the long
    ones (budget 40) hand their queries to the one-wave-per-query kernel; results must not move.
    Covers unique minimisers, near ties (path check), exact ties (DFS fold) and the sdf tail."""
    rng = np.random.default_rng(21)
    # (a) tie-heavy lattice, non-robust
    g = np.stack(np.meshgrid(*[np.arange(-6, 7)] * 3, indexing="ij"), -1).reshape(-1, 3)
    v = (g[rng.permutation(g.shape[0])[:1500]] * 0.125).astype(np.float32)
    v = np.concatenate([v, v[:40]])
    f = np.array([[0, 1, 2], [3, 4, 5]], np.uint32)
    q = np.concatenate([(rng.integers(-14, 15, (30000, 3)) * 0.0625), rng.uniform(-1, 1, (20000, 3)),
                        v[rng.integers(0, v.shape[0], 3000)]]).astype(np.float32)
    o = po.OracleSDF(v, f, False)
    s = SDF(v, f, robust=False)
    s.set_option("heavy_budget", budget)
    s.set_option("profile", 1)
    assert (s.nn(q) == o.nn(q)).all()
    assert s.stats["last_heavy_points"] > (q.shape[0] // 2 if budget == 1 else 0)
    assert same_float(s(q), o(q)).all()
    # (b) torus, robust: sign by ray parity recomputed in the heavy kernel
    v, f = meshes.torus(120, 120)
    R = meshes.random_rotation(9)
    o = po.OracleSDF(v, f, True, R)
    s = SDF(v, f)
    s.set_raycast_axes(R)
    s.set_option("heavy_budget", budget)
    p = np.concatenate([rng.uniform(-1, 1, (80000, 3)), meshes.grid_points(30),
                        rng.normal(0, 1e-5, (3000, 3))]).astype(np.float32)  # + the symmetry centre
    assert (s.nn(p) == o.nn(p)).all()
    assert same_float(s(p), o(p)).all() and same_float(s(p, True), o(p, True)).all()
    assert (s.contains(p) == o.contains(p)).all()
    sn = SDF(v, f, robust=False)
    sn.set_option("heavy_budget", budget)
    on = po.OracleSDF(v, f, False)
    assert same_float(sn(p), on(p)).all() and (sn.contains(p) == on.contains(p)).all()


@pytest.mark.parametrize("budget", [4, 60])
def test_cooperative_continuation_forced(budget):
    """k_query_packet_coop: with a tiny hand-over budget (nearly) every packet's walk is finished by eight
    waves from the saved lists and merged; the hard budget stays automatic, so the answers come from
    that kernel and not from the exact scan.  Same bits as the oracle: ties, near ties, sdf tail."""
    rng = np.random.default_rng(22)
    g = np.stack(np.meshgrid(*[np.arange(-6, 7)] * 3, indexing="ij"), -1).reshape(-1, 3)
    v = (g[rng.permutation(g.shape[0])[:1500]] * 0.125).astype(np.float32)
    v = np.concatenate([v, v[:40]])
    f = np.array([[0, 1, 2], [3, 4, 5]], np.uint32)
    q = np.concatenate([(rng.integers(-14, 15, (30000, 3)) * 0.0625), rng.uniform(-1, 1, (20000, 3)),
                        v[rng.integers(0, v.shape[0], 3000)]]).astype(np.float32)
    o = po.OracleSDF(v, f, False)
    s = SDF(v, f, robust=False)
    for key, val in (("approx", 1), ("coop_budget", budget), ("profile", 1)):
        s.set_option(key, val)
    assert (s.nn(q) == o.nn(q)).all()
    assert s.stats["last_heavy_points"] < q.shape[0] // 4
    assert same_float(s(q), o(q)).all()
    v, f = meshes.torus(120, 120)
    R = meshes.random_rotation(10)
    o = po.OracleSDF(v, f, True, R)
    s = SDF(v, f)
    s.set_raycast_axes(R)
    for key, val in (("approx", 1), ("coop_budget", budget)):
        s.set_option(key, val)
    p = np.concatenate([rng.uniform(-1, 1, (80000, 3)), meshes.grid_points(30),
                        rng.normal(0, 1e-5, (3000, 3))]).astype(np.float32)
    want_nn, want = o.nn(p), o(p)
    assert (s.nn(p) == want_nn).all()
    assert same_float(s(p), want).all() and same_float(s(p, True), o(p, True)).all()
    s.set_option("coop_budget", 0)  # and without the kernel
    assert (s.nn(p) == want_nn).all() and same_float(s(p), want).all()


def test_sphere_centre_queries_take_the_heavy_path():
    """icosphere: queries near the centre need (nearly) every vertex; with the automatic budget
    their waves give up and the heavy kernel answers — same bits as the oracle."""
    v, f = meshes.icosphere(6, radius=0.8)  # 40,962 v / 81,920 f
    R = meshes.random_rotation(4)
    o = po.OracleSDF(v, f, True, R)
    s = SDF(v, f)
    s.set_raycast_axes(R)
    s.set_option("profile", 1)
    rng = np.random.default_rng(8)
    p = np.concatenate([rng.uniform(-1, 1, (150000, 3)), rng.uniform(-0.02, 0.02, (4000, 3)),
                        np.zeros((64, 3))]).astype(np.float32)
    n = s.nn(p)
    assert s.stats["last_heavy_points"] >= 64
    assert (n == o.nn(p)).all()
    assert same_float(s(p), o(p)).all()


def test_automatic_walk_choice_looks_at_local_vertex_density(monkeypatch):
    """A fine part in a coarse scene: the average vertex density admits the approximate pass, the density of the
    fine part's groups does not (every query nearest to it would meet four candidate groups and go to the exact
    scan).  The automatic choice must fall back to the exact walk there, and keep the pass on the plain torus."""
    monkeypatch.delenv("SDFHIP_APPROX", raising=False)
    v, f = meshes.torus(200, 200)
    v2 = (v * np.float32(0.01) + np.float32([0.5, 0.0, 0.6])).astype(np.float32)
    V = np.concatenate([v, v2])
    F = np.concatenate([f, f + v.shape[0]]).astype(np.uint32)
    rng = np.random.default_rng(4)
    q = rng.uniform(-1, 1, (300000, 3)).astype(np.float32)
    kinds = {}
    for name, (vv, ff) in {"plain": (v, f), "fine part": (V, F)}.items():
        s = SDF(vv, ff, robust=False)
        s.set_option("scan_max", 0)
        s.set_option("profile", 1)
        got = s.nn(q)
        kinds[name] = s.stats["last_kernel"]
        o = po.OracleSDF(vv, ff, False)
        assert (got == o.nn(q)).all()
    assert kinds["plain"] == 3 and kinds["fine part"] == 2, kinds  # SDFHIP_KERNEL_PACKET_APPROX / _EXACT


@pytest.mark.parametrize("far_packets", [1, 0])
@pytest.mark.parametrize("half, n", [(300.0, 400_000), (30.0, 2_000_000)])
def test_batches_spread_far_beyond_the_mesh(half, n, far_packets):
    """Queries spread over a box 30 / 300 times the mesh.  far_packets=1 (default): far queries are sorted by the
    direction they see the mesh from and walked as far packets (k_query_packet_far; 8M points in the +-300 box:
    50 -> 7 ms), few are left to the lists.  far_packets=0, the way of round 2: Morton packets as wide as the mesh go
    to the heavy list unwalked (packet_walk), and a heavy list this long is answered by the strict kernel instead of
    the exact scan (554 -> 50 ms).  Results must not notice either."""
    v, f = meshes.torus(424, 424)
    R = meshes.random_rotation(11)
    s = SDF(v, f)
    s.set_raycast_axes(R)
    s.set_option("profile", 1)
    s.set_option("far_packets", far_packets)
    o = po.OracleSDF(v, f, True, R)
    q = (np.random.default_rng(21).uniform(-1, 1, (n, 3)) * half).astype(np.float32)
    assert (s.nn(q) == o.nn(q)).all()
    if far_packets:
        assert s.stats["last_heavy_points"] < n // 50
    else:
        assert s.stats["last_heavy_points"] > 210_000  # (the strict kernel's share: above 4e10 / tree positions)
    assert same_float(s(q), o(q)).all()
    assert (s.contains(q) == o.contains(q)).all()
