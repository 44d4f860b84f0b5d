"""GPU parity tests at BASELINE.json's configurations (through the C ABI / pysdf surface).
Full-size oracle comparisons where the CPU oracle finishes in seconds; size-independent
properties otherwise."""
import numpy as np
import pytest

from conftest import load_golden, same_float
from oracle import pyoracle as po
from sdf_amd import SDF, meshes

pytestmark = pytest.mark.gpu


def test_config1_teapot_robust_grid64():
    g = load_golden("teapot")
    v, f, R = g["verts"], g["faces"], g["R"]
    pts = meshes.grid_points(64)  # 262,144 points, i-major, [-1,1]^3
    o = po.OracleSDF(v, f, True, R)
    s = SDF(v, f, robust=True)
    s.set_raycast_axes(R)
    d = s(pts)
    assert same_float(d, o(pts)).all()
    assert (s.contains(pts) == o.contains(pts)).all()
    assert (s.nn(pts) == o.nn(pts)).all()
    assert same_float(s(pts, True), o(pts, True)).all()


def test_config2_torus_obj_nonrobust_random():
    g = load_golden("torus")
    v, f = g["verts"], g["faces"]
    rng = np.random.default_rng(7)
    n = 10_000_000
    pts = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    s = SDF(v, f, robust=False)
    d = s(pts)
    c = s.contains(pts)
    k = 2_000_000
    o = po.OracleSDF(v, f, False)
    assert same_float(d[:k], o(pts[:k])).all()
    assert (c[:k] == o.contains(pts[:k])).all()
    # contains == (sdf(trunc_aabb=True) >= 0) for non-robust handles (sdf.cpp:171-172)
    dt = s(pts, True)
    assert (c == (dt >= 0)).all()
    box = np.all((pts >= s.aabb[:3]) & (pts <= s.aabb[3:]), axis=1)
    assert np.isnan(dt[~box]).all() and same_float(dt[box], d[box]).all()


def test_config3_smpl_scale_grid256_robust():
    v, f = meshes.torus(83, 83)  # 6,889 v / 13,778 f
    R = meshes.random_rotation(1234)
    s = SDF(v, f)
    s.set_raycast_axes(R)
    t = np.linspace(-1, 1, 256, dtype=np.float32)
    pts = np.stack(np.meshgrid(t, t, t, indexing="ij"), -1).reshape(-1, 3)
    d = s(pts)
    o = po.OracleSDF(v, f, True, R)
    assert same_float(d, o(pts)).all()  # 16.7M points, full size
    # sharding invariance: any contiguous split gives the same values (multi-GPU partition)
    m = pts.shape[0] // 3
    assert same_float(s(pts[m:2 * m]), d[m:2 * m]).all()


def test_config5_deforming_mesh_frames():
    v0, f = meshes.torus(160, 160)  # 25,600 v / 51,200 f
    R = meshes.random_rotation(4)
    s = SDF(v0, f, robust=True, copy=True)
    s.set_raycast_axes(R)
    pts = np.random.default_rng(2).uniform(-1, 1, (400_000, 3)).astype(np.float32)
    for t in (0, 33, 71):
        v = (v0 * (1 + 0.05 * np.sin(2 * np.pi * t / 100 + 4 * v0[:, 1:2]))).astype(np.float32)
        s.vertices_mutable = v
        s.update()
        o = po.OracleSDF(v, f, True, R)
        assert (s.contains(pts) == o.contains(pts)).all()
        assert (s.nn(pts) == o.nn(pts)).all()
        assert same_float(s.face_normals, o.face_normals).all()
    # unpinned handles draw a new rotation per update (sdf.cpp:43-48): still a rotation, and
    # contains stays right away from the surface
    s2 = SDF(v0, f)
    Ra = s2.raycast_axes.copy()
    s2.update()
    Rb = s2.raycast_axes
    assert not np.allclose(Ra, Rb) and np.allclose(Rb @ Rb.T, np.eye(3), atol=1e-5)
    rho = np.sqrt(pts[:, 0] ** 2 + pts[:, 1] ** 2) - 0.7
    ana = 0.24 - np.sqrt(rho ** 2 + pts[:, 2] ** 2)
    far = np.abs(ana) > 5e-3
    assert ((ana > 0) == s2.contains(pts))[far].all()


def _perf_guards():
    """Wall-clock guards (about 1.5 x the figures measured on an idle MI355X; each is the best of several
    repeats, so a noisy neighbour has to be there every time).  SDFHIP_PERF_GUARDS=0 keeps the measurements
    (printed) and drops the assertions, for boxes that are shared or throttled."""
    import os
    return os.environ.get("SDFHIP_PERF_GUARDS", "1") != "0"


def test_config5_full_size_100_frames():
    """BASELINE config 5 as stated: torus 424 x 424 (359,552 faces), 100 frames of vertices_mutable + update() +
    contains() + nn() on the same 4M device-resident seeded points.  Pass 1 issues the frames back to back (the
    queries are asynchronous on the stream; the host kd-tree build of frame t+1 overlaps the GPU work of frame
    t) and keeps every frame's results on the device; six of the hundred frames are then checked against the
    oracle on a sample.  Pass 2 runs the same frames synchronously for the per-frame split update / contains /
    nn that SURVEY §8(d) asks for (bench.py reports the same split under extra.c5_strict / c5_fast_nn)."""
    import time
    import torch
    v0, f = meshes.torus(424, 424)
    R = meshes.random_rotation(4)
    s = SDF(v0, f, robust=True, copy=True)
    s.set_raycast_axes(R)
    dev = torch.device("cuda", 0)
    pts = torch.rand((4_000_000, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(7)).mul_(2).sub_(1)
    frames, warm = 100, 2
    y4 = 4 * v0[:, 1:2]
    vs = [(v0 * (1 + 0.05 * np.sin(2 * np.pi * t / 100 + y4))).astype(np.float32) for t in range(frames)]
    res = []
    best = 1e9
    for rep in range(2):
        res = []
        for t in range(frames):
            if t == warm:  # the first two frames warm the workspaces up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            s.vertices_mutable = vs[t]
            s.update()
            res.append((s.contains(pts), s.nn(pts)))
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (frames - warm) * 1e3)
    print(f"config 5, full size, {frames} frames back to back: {best:.2f} ms per frame (update + contains(4M) + nn(4M))")
    if _perf_guards():
        # 3.4-3.9 ms on most boxes of the pool, 4.7-5.0 on the slowest seen (the frame is bound by the HOST kd build and
        # group pass, 1.8-3.0 ms depending on the box's CPU share); catches a lost overlap or a slow path
        assert best < 6.5
    sub = torch.randint(0, pts.shape[0], (3000,), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    h = pts[sub].cpu().numpy()
    for t in (0, 1, 24, 50, 75, 99):
        o = po.OracleSDF(vs[t], f, True, R)
        assert (res[t][0][sub].cpu().numpy().astype(bool) == o.contains(h)).all()
        assert (res[t][1][sub].cpu().numpy() == o.nn(h)).all()
    inside = [int(r[0].sum()) for r in res]
    assert len(set(inside)) > 50  # the mesh really moved from frame to frame
    del res
    tu = tc = tn = 0.0
    for t in range(frames):
        torch.cuda.synchronize()
        a = time.perf_counter()
        s.vertices_mutable = vs[t]
        s.update()
        torch.cuda.synchronize()
        b = time.perf_counter()
        c = s.contains(pts)
        torch.cuda.synchronize()
        d = time.perf_counter()
        n = s.nn(pts)
        torch.cuda.synchronize()
        e = time.perf_counter()
        if t >= warm:
            tu, tc, tn = tu + (b - a), tc + (d - b), tn + (e - d)
    k = (frames - warm) / 1e3
    print(f"config 5 split per frame (synchronous): update {tu / k:.2f} ms  contains(4M) {tc / k:.2f} ms  nn(4M) {tn / k:.2f} ms")
    assert int(c.sum()) == inside[-1] and n.shape[0] == pts.shape[0]


def test_update_from_device_tensor():
    import torch
    v0, f = meshes.torus(60, 60)
    R = meshes.random_rotation(6)
    s = SDF(v0, f, copy=True)
    s.set_raycast_axes(R)
    v1 = (v0 * np.float32(0.9)).astype(np.float32)
    s.update_from_device(torch.from_numpy(v1).cuda())
    o = po.OracleSDF(v1, f, True, R)
    pts = np.random.default_rng(1).uniform(-1, 1, (50_000, 3)).astype(np.float32)
    assert same_float(s(pts), o(pts)).all() and (s.verts == v1).all()


def test_config4_full_shard_device_resident_properties():
    """BASELINE config 4 at its full per-GPU size (125M device-resident points, the bench.py
    workload): the CPU oracle cannot cover it, so size-independent properties — results do not
    depend on which batch a point is evaluated in (the Morton sort, the wave it lands in and
    the chunking all change), sign == contains, the analytic torus — plus an oracle check of a
    random sample of the very rows of the big result."""
    import torch
    v, f = meshes.torus(424, 424)
    R = meshes.random_rotation(1234)
    s = SDF(v, f, robust=True)
    s.set_raycast_axes(R)
    dev = torch.device("cuda", 0)
    n = 125_000_000
    g = torch.Generator(device=dev).manual_seed(7)
    pts = torch.rand((n, 3), device=dev, generator=g, dtype=torch.float32).mul_(2).sub_(1)
    d = s(pts)
    c = s.contains(pts)
    nn = s.nn(pts)
    assert s.stats["last_fallback_points"] >= 0
    assert bool(((d > 0) == c)[d != 0].all())
    rho = torch.sqrt(pts[:, 0] ** 2 + pts[:, 1] ** 2) - 0.7
    ana = 0.24 - torch.sqrt(rho ** 2 + pts[:, 2] ** 2)
    far = ana.abs() > 2e-3
    assert bool(((ana > 0) == c)[far].all())
    assert float((d - ana).abs().max()) < 2e-3
    del rho, ana, far
    # nn is a true nearest vertex: no vertex of the mesh is closer (checked through |sdf| <= dist)
    tv = torch.from_numpy(v).to(dev)
    dn = (tv[nn.long()] - pts).norm(dim=1)
    assert bool((d.abs() <= dn * (1 + 1e-6)).all())
    del dn
    # batch independence: a strided subset evaluated alone gives the same bits
    idx = torch.arange(0, n, 97, device=dev)
    sub = pts[idx].contiguous()
    assert bool((s(sub).view(torch.int32) == d[idx].view(torch.int32)).all())
    assert bool((s.contains(sub) == c[idx]).all())
    assert bool((s.nn(sub) == nn[idx]).all())
    # and a slice through the host entry points (chunked, overlapped copies)
    hs = pts[: 3_000_000].cpu().numpy()
    assert same_float(s(hs), d[: 3_000_000].cpu().numpy()).all()
    # oracle on a random sample of rows of the big result
    k = torch.randint(0, n, (200_000,), device=dev, generator=g)
    hp = pts[k].cpu().numpy()
    o = po.OracleSDF(v, f, True, R)
    assert same_float(d[k].cpu().numpy(), o(hp)).all()
    assert (c[k].cpu().numpy() == o.contains(hp)).all()
    assert (nn[k].cpu().numpy() == o.nn(hp)).all()


def test_config4_mesh_grid256_oracle_sample():
    """the metric's second figure (BASELINE.json: ms for a 256^3 grid on the ~360k-face mesh): 16.8M perfectly
    coherent grid points, device resident; an oracle check of a random sample of rows, sign == contains,
    batch independence of a strided subset, and the time of the call (bench.py reports it as grid256_ms)"""
    import time
    import torch
    v, f = meshes.torus(424, 424)
    R = meshes.random_rotation(1234)
    s = SDF(v, f, robust=True)
    s.set_raycast_axes(R)
    dev = torch.device("cuda", 0)
    t = torch.linspace(-1, 1, 256, device=dev)
    pts = torch.stack(torch.meshgrid(t, t, t, indexing="ij"), -1).reshape(-1, 3).contiguous()
    d = s(pts)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); d = s(pts); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"256^3 grid, 359,552-face torus: {best * 1e3:.2f} ms")
    if _perf_guards():
        assert best < 5e-3, best  # 3.3 ms on an idle MI355X (5.1 in round 1); best of five
    c, nn = s.contains(pts), s.nn(pts)
    assert bool(((d > 0) == c)[d != 0].all())
    g = torch.Generator(device=dev).manual_seed(3)
    k = torch.randint(0, pts.shape[0], (150_000,), device=dev, generator=g)
    hp = pts[k].cpu().numpy()
    o = po.OracleSDF(v, f, True, R)
    assert same_float(d[k].cpu().numpy(), o(hp)).all()
    assert (c[k].cpu().numpy() == o.contains(hp)).all()
    assert (nn[k].cpu().numpy() == o.nn(hp)).all()
    idx = torch.arange(0, pts.shape[0], 53, device=dev)
    sub = pts[idx].contiguous()
    assert bool((s(sub).view(torch.int32) == d[idx].view(torch.int32)).all())


def test_page_locked_host_arrays_and_out_argument():
    """host entry points on page-locked memory (sdfhip_host_alloc) and with a caller-provided
    result array: same bits as the default"""
    from sdf_amd import pinned_empty
    v, f = meshes.torus(83, 83)
    R = meshes.random_rotation(2)
    s = SDF(v, f)
    s.set_raycast_axes(R)
    n = 1_500_000
    pts = np.random.default_rng(5).uniform(-1, 1, (n, 3)).astype(np.float32)
    d, c, k = s(pts), s.contains(pts), s.nn(pts)
    pp = pinned_empty((n, 3), np.float32)
    pp[:] = pts
    od, oc, ok = pinned_empty(n, np.float32), pinned_empty(n, np.bool_), pinned_empty(n, np.int32)
    assert s(pp, out=od) is od and same_float(od, d).all()
    assert (s.contains(pp, out=oc) == c).all() and (s.nn(pp, out=ok) == k).all()
    with pytest.raises(TypeError):
        s(pts, out=np.empty(n, np.float64))
    o = po.OracleSDF(v, f, True, R)
    assert same_float(od[:100000], o(pts[:100000])).all()


def test_config4_icosphere7_variant_oracle_sample():
    """BASELINE config 4's second mesh: icosphere subdiv-7 (163,842 v / 327,680 f),
    radius 0.8 with 5 % seeded radial noise; robust, random points; oracle on a sample."""
    v, f = meshes.icosphere(7, radius=0.8, noise=0.05, seed=3)
    assert v.shape[0] == 163_842 and f.shape[0] == 327_680
    R = meshes.random_rotation(1234)
    s = SDF(v, f, robust=True)
    s.set_raycast_axes(R)
    rng = np.random.default_rng(7)
    pts = rng.uniform(-1, 1, (4_000_000, 3)).astype(np.float32)
    d, c, n = s(pts), s.contains(pts), s.nn(pts)
    assert ((d > 0) == c)[d != 0].all()
    rad = np.linalg.norm(pts, axis=1)
    rv = np.linalg.norm(v, axis=1)  # the noise is radial: star-shaped around the origin
    assert c[rad < rv.min() * 0.98].all() and not c[rad > rv.max()].any()
    k = 250_000
    o = po.OracleSDF(v, f, True, R)
    assert (n[:k] == o.nn(pts[:k])).all()
    assert (c[:k] == o.contains(pts[:k])).all()
    assert same_float(d[:k], o(pts[:k])).all()
