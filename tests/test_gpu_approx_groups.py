"""The approximate first pass of the packet kernel estimates squared distances relative to the centre of each packet
group (kernels_update.h: k_apx_fill; kernels_packet.h: group_approx, apx_err): its error is of the size of the
distance and of a group, not of the scene, so it stays on for meshes far off the origin and for dense meshes.
Results may never depend on it: bit-equal to the oracle (oracle/pyoracle.py <- reference sdf.cpp:110-160,
nanoflann.hpp:432-440) wherever the mesh lies, whatever its groups look like."""
import numpy as np
import pytest

from conftest import same_float
from oracle import pyoracle as po
from sdf_amd import SDF, meshes

pytestmark = pytest.mark.gpu
PACKET_APPROX = 3  # SDFHIP_KERNEL_PACKET_APPROX (include/sdfhip.h)


def _check(s, o, q, kinds=("nn", "sdf")):
    if "nn" in kinds:
        assert (s.nn(q) == o.nn(q)).all()
    if "sdf" in kinds:
        assert same_float(s(q), o(q)).all()


@pytest.mark.parametrize("scale, shift", [(1e-3, (500.0, -300.0, 200.0)), (1.0, (4096.0, 4096.0, -4096.0)),
                                          (250.0, (1e5, 0.0, -3e4)), (1e-4, (0.0, 0.0, 0.0))])
def test_pass_on_meshes_far_off_the_origin_and_of_any_size(scale, shift):
    """fp32 spacing at the mesh is a sizeable fraction of the edge length in the first two cases: many exact ties and
    near ties, all of which must resolve as nanoflann resolves them"""
    v, f = meshes.torus(160, 160)
    v = (v * np.float32(scale) + np.asarray(shift, np.float32)).astype(np.float32)
    R = meshes.random_rotation(21)
    rng = np.random.default_rng(5)
    ext = float(np.abs(v - v.mean(0)).max())
    c = v.mean(0)
    q = np.concatenate([c + rng.uniform(-1.5 * ext, 1.5 * ext, (150_000, 3)),      # around the mesh
                        v[rng.integers(0, v.shape[0], 20_000)] + rng.normal(0, 0.01 * ext, (20_000, 3)),  # at it
                        v[rng.integers(0, v.shape[0], 2_000)],                      # on vertices
                        c + rng.normal(0, 1e-4 * ext, (2_000, 3)),                  # the centre of symmetry
                        c + rng.uniform(-40 * ext, 40 * ext, (20_000, 3))]).astype(np.float32)  # far away
    o = po.OracleSDF(v, f, True, R)
    s = SDF(v, f, robust=True)
    s.set_raycast_axes(R)
    s.set_option("approx", 1)
    s.set_option("scan_max", 0)
    s.set_option("profile", 1)
    _check(s, o, q)
    assert s.stats["last_kernel"] == PACKET_APPROX
    s.set_option("cube_lists", 1)
    _check(s, o, q)


@pytest.mark.parametrize("wild", [False, True])
def test_pass_with_outliers_duplicates_and_huge_groups(wild):
    """one group's radius sets the absolute term of the error model for every query: vertices far from everything
    (a group spanning the scene) and exact duplicates keep the pass.  wild: vertices at 1e30 and 3e38 as well — their
    group is so wide that |v - centre|^2 overflows for its ordinary vertices too, which the pass could then never
    find: it must step aside for the whole mesh (exact walk), whatever the option says"""
    v, f = meshes.icosphere(5, radius=0.7, noise=0.01, seed=2)
    rng = np.random.default_rng(6)
    extra = np.array([[40.0, 0, 0], [0, -55.0, 3.0], [1e30, 0, 0], [0, 3e38, 0], [-3e38, -3e38, 3e38]], np.float32)
    extra = extra if wild else extra[:2]
    dup = v[rng.integers(0, v.shape[0], 300)]
    V = np.concatenate([v, extra, dup]).astype(np.float32)
    F = np.concatenate([f, [[0, v.shape[0], v.shape[0] + 1]]]).astype(np.uint32)
    q = np.concatenate([rng.uniform(-1, 1, (120_000, 3)), dup[:200] + rng.normal(0, 1e-6, (200, 3)),
                        rng.uniform(-60, 60, (30_000, 3)), extra[:2] + rng.normal(0, 0.5, (2, 3)),
                        [[9e29, 1.0, 0.0], [0.0, 2.9e38, 0.0], [0.0, 3e38, 0.0]]]).astype(np.float32)
    o = po.OracleSDF(V, F, False)
    for group in (0, 24, 60):
        s = SDF(V, F, robust=False)
        s.set_option("approx", 1)
        s.set_option("scan_max", 0)
        s.set_option("profile", 1)
        if group:
            s.set_option("packet_group", group)
            s.update()
        _check(s, o, q)
        assert s.stats["last_kernel"] == (2 if wild else PACKET_APPROX)  # 2: SDFHIP_KERNEL_PACKET_EXACT


def test_dense_mesh_takes_the_pass_by_itself(monkeypatch):
    """490k vertices: the mesh-centred pass of round 2 had to leave such meshes to the exact walk (its window held
    ~0.5 vertices on the average); the group-centred one keeps them, without sending queries to the exact scan"""
    monkeypatch.delenv("SDFHIP_APPROX", raising=False)
    import torch
    v, f = meshes.torus(700, 700)
    s = SDF(v, f, robust=False)
    s.set_option("profile", 1)
    dev = torch.device("cuda", 0)
    pts = torch.rand((4_000_000, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(9)).mul_(2).sub_(1)
    got = s.nn(pts)
    st = s.stats
    assert st["last_kernel"] == PACKET_APPROX
    assert st["last_heavy_points"] < 4000 and st["last_fallback_points"] < 100, st
    k = torch.randint(0, pts.shape[0], (30_000,), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    hp = pts[k].cpu().numpy()
    o = po.OracleSDF(v, f, False)
    assert (got[k].cpu().numpy() == o.nn(hp)).all()
    s.set_option("approx", 0)
    assert bool((s.nn(pts) == got).all())


def test_lanes_that_met_fewer_than_four_groups_are_not_fourth_candidate_queries(monkeypatch):
    """coarse mesh, dense grid: most waves settle after one to three groups, their lanes' third and fourth minima stay
    +inf — which the window tests must read as "outside" (a NaN from inf - inf read as "inside" once sent 1.2 % of
    such a grid to the exact scan: nn 1.2 -> 1.8 ms)"""
    import torch
    monkeypatch.delenv("SDFHIP_APPROX", raising=False)
    v, f = meshes.torus(83, 83)
    s = SDF(v, f, robust=False)
    s.set_option("profile", 1)
    t = torch.linspace(-1, 1, 128, device="cuda")
    pts = torch.stack(torch.meshgrid(t, t, t, indexing="ij"), -1).reshape(-1, 3).contiguous()
    got = s.nn(pts)
    st = s.stats
    assert st["last_kernel"] == PACKET_APPROX
    assert st["last_heavy_points"] < 500 and st["last_fallback_points"] < 100, st
    k = torch.randint(0, pts.shape[0], (30_000,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    o = po.OracleSDF(v, f, False)
    assert (got[k].cpu().numpy() == o.nn(pts[k].cpu().numpy())).all()
