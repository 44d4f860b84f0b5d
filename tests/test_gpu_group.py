"""The multi-GPU group handle of the C ABI (sdfhip_group_*) on the one-GPU box: the listed
devices are {0, 0}, i.e. two members on one card — every piece of the partitioning (ranges, per-member
threads / staging / streams, replicated mesh and rotation, update()) runs; only the second card is missing.
Checked against the unsharded handle and the oracle."""
import numpy as np
import pytest

from conftest import same_float
from oracle import pyoracle as po
from sdf_amd import SDF, GroupSDF, meshes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ndev", [2, 3])
def test_group_matches_single_handle_and_oracle(ndev):
    v, f = meshes.torus(83, 83)
    R = meshes.random_rotation(77)
    g = GroupSDF(v, f, robust=True, devices=[0] * ndev)
    assert len(g) == ndev
    g.set_raycast_axes(R)
    s = SDF(v, f, robust=True)
    s.set_raycast_axes(R)
    rng = np.random.default_rng(3)
    for n in (1, 5, 100_003, 700_001):  # fewer points than members, ragged ranges, sorted + packet path
        pts = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        assert same_float(g(pts), s(pts)).all()
        assert (g.contains(pts) == s.contains(pts)).all()
        assert (g.nn(pts) == s.nn(pts)).all()
    pts = rng.uniform(-1, 1, (20_000, 3)).astype(np.float32)
    o = po.OracleSDF(v, f, True, R)
    assert same_float(g(pts, True), o(pts, True)).all() and (g.nn(pts) == o.nn(pts)).all()


def test_group_update_replicates_vertices_and_one_rotation():
    """three frames of the deforming-mesh loop (demo.cpp:109-121): vertices written through member 0's
    host copy, update() on the group; un-pinned, member 0 draws R and every member uses the same one"""
    v, f = meshes.torus(60, 60)
    g = GroupSDF(v, f, robust=True, devices=[0, 0])
    pts = np.random.default_rng(9).uniform(-1, 1, (300_000, 3)).astype(np.float32)
    R_prev = g.raycast_axes.copy()
    L = g._L
    for t in (1, 2, 3):
        vt = (v * (1 + 0.05 * np.sin(2 * np.pi * t / 100 + 4 * v[:, 1:2]))).astype(np.float32)
        g.vertices_mutable[:] = vt
        g.update()
        R = g.raycast_axes.copy()
        assert not np.array_equal(R, R_prev)  # a fresh rotation per update, like the reference
        R_prev = R
        from sdf_amd.pysdf import _fp
        for k in range(len(g)):  # the same matrix on every member
            Rk = np.empty(9, np.float32)
            assert L.sdfhip_get_raycast_axes(L.sdfhip_group_member(g._g, k), Rk.ctypes.data_as(_fp)) == 0
            assert np.array_equal(Rk.reshape(3, 3), R)
        o = po.OracleSDF(vt, f, True, R)
        sub = slice(0, 40_000)  # a prefix of the first range and ...
        sub2 = slice(260_000, 300_000)  # ... a suffix of the second
        c, n = g.contains(pts), g.nn(pts)
        for sl in (sub, sub2):
            assert (c[sl] == o.contains(pts[sl])).all() and (n[sl] == o.nn(pts[sl])).all()
    # pinned by the caller: update keeps it
    Rp = meshes.random_rotation(5)
    g.set_raycast_axes(Rp)
    g.update()
    assert np.array_equal(g.raycast_axes, Rp)
    assert same_float(g(pts[:30_000]), po.OracleSDF(vt, f, True, Rp)(pts[:30_000])).all()


def _member_handle(g, k):
    return g._L.sdfhip_group_member(g._g, k)


def test_group_builds_the_tree_once_and_replicates_device_to_device():
    """one kd build per create / update for the whole group (member 0, all build threads);
    the other members receive vertices + tree device to device and never build (counter kd_builds), yet answer
    bit for bit like the oracle — on host ranges and on device-resident shards."""
    import torch
    v, f = meshes.torus(200, 200)  # 40,000 v / 80,000 f
    R = meshes.random_rotation(21)
    g = GroupSDF(v, f, robust=True, devices=[0, 0, 0, 0])
    g.set_raycast_axes(R)
    single = SDF(v, f, robust=True)
    single.set_raycast_axes(R)
    single.update()
    st = [g.member_stats(k) for k in range(4)]
    assert st[0]["kd_builds"] == 1 and [x["kd_builds"] for x in st[1:]] == [0, 0, 0]
    assert all(x["kd_nodes"] == st[0]["kd_nodes"] and x["kd_depth"] == st[0]["kd_depth"] for x in st)
    assert all(x["last_kd_build_ms"] == 0.0 for x in st[1:])
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(11)
    sizes = [300_000, 0, 1, 170_003]  # ragged shards: an idle member, a single point
    o = None
    for frame in range(3):
        if frame:
            vt = (v * (1 + 0.05 * np.sin(2 * np.pi * frame / 100 + 4 * v[:, 1:2]))).astype(np.float32)
            if frame == 1:
                g.vertices_mutable[:] = vt
                g.update()
            else:  # vertices produced on the GPU
                g.update_from_device(torch.from_numpy(vt).to(dev))
                assert (g.vertices_mutable == vt).all()
        else:
            vt = v
        st = [g.member_stats(k) for k in range(4)]
        assert st[0]["kd_builds"] == 1 + frame and all(x["kd_builds"] == 0 for x in st[1:])
        o = po.OracleSDF(vt, f, True, R)
        shards = [torch.rand((n, 3), device=dev, generator=gen).mul_(2).sub_(1) for n in sizes]
        d, c, k = g.sdf_device(shards), g.contains_device(shards), g.nn_device(shards)
        torch.cuda.synchronize()
        for m, n in enumerate(sizes):
            assert d[m].shape[0] == n and c[m].shape[0] == n and k[m].shape[0] == n
            take = min(n, 30_000)
            if take == 0:
                continue
            hp = shards[m][:take].cpu().numpy()
            assert same_float(d[m][:take].cpu().numpy(), o(hp)).all()
            assert (c[m][:take].cpu().numpy() == o.contains(hp)).all()
            assert (k[m][:take].cpu().numpy() == o.nn(hp)).all()
        # every member holds the same structures: the same shard gives the same bits on each of them
        same = [shards[0], shards[0], shards[0], shards[0]]
        dd = g.sdf_device(same)
        assert all(bool((dd[0].view(torch.int32) == x.view(torch.int32)).all()) for x in dd[1:])
    # a group update costs the leader what a single handle's update costs: one build with all threads
    kd_group, kd_single = [], []
    for _ in range(5):
        g.update()
        single.update()
        kd_group.append(g.member_stats(0)["last_kd_build_ms"])
        kd_single.append(single.stats["last_kd_build_ms"])
    print(f"kd build per update: group leader {min(kd_group):.2f} ms, single handle {min(kd_single):.2f} ms")
    assert min(kd_group) < 2.0 * min(kd_single) + 0.5


def test_group_fast_nn_followers_build_on_their_own_gpu():
    v, f = meshes.torus(120, 120)
    g = GroupSDF(v, f, robust=True, devices=[0, 0])
    R = meshes.random_rotation(3)
    g.set_raycast_axes(R)
    L = g._L
    assert L.sdfhip_set_option(_member_handle(g, 0), b"fast_nn", 1) == 0  # the leader's option leads
    g.update()
    pts = np.random.default_rng(4).uniform(-1, 1, (200_000, 3)).astype(np.float32)
    n = g.nn(pts)
    o = po.OracleSDF(v, f, True, R)
    on = o.nn(pts)
    dg, do = ((pts - v[n]) ** 2).sum(-1), ((pts - v[on]) ** 2).sum(-1)
    assert np.allclose(dg, do, rtol=1e-6, atol=0)  # a nearest vertex (the tie rule differs by design)
    assert (g.contains(pts) == o.contains(pts)).all()
    assert g.member_stats(1)["kd_builds"] == 0 and g.member_stats(0)["kd_builds"] == 1
