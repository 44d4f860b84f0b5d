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
