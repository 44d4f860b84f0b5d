"""Cube lists of the packet pass (kernels_packet.h: k_cube_lists, option "cube_lists"): the query domain of a
sorted launch is cut into cubes, one wave per cube lists the packet groups its queries can need, and the packet
waves test that list instead of walking the boxes from the top.  A list may only leave out groups that are
farther from every query of the cube than the query's final threshold, so results must not depend on it:
bit-equal to the plain walk and to the oracle, for every cube size, also where lists overflow, queries lie
outside the sampled bounds, waves straddle cubes or the mesh moved.  The tuning build counts the waves that
really took a list (child process, so that this process keeps the product library)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import same_float
from oracle import pyoracle as po
from sdf_amd import SDF, meshes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STATS = os.path.join(ROOT, "sdf_amd", "libsdfhip_stats.so")


def _both(s, pts, kinds=("nn", "sdf")):
    """results with the lists off and on (torch tensors on the device)"""
    out = {}
    for lists in (0, 1):
        s.set_option("cube_lists", lists)
        out[lists] = {"nn": s.nn(pts) if "nn" in kinds else None, "sdf": s(pts) if "sdf" in kinds else None}
    return out


@pytest.mark.parametrize("approx", [0, 1])
@pytest.mark.parametrize("level", [2, 4, 5])
def test_lists_equal_the_walk_and_the_oracle(level, approx):
    import torch
    v, f = meshes.torus(150, 150)
    R = meshes.random_rotation(31)
    s = SDF(v, f, robust=True)
    s.set_raycast_axes(R)
    s.set_option("approx", approx)
    s.set_option("cube_level", level)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(level)
    # a dense box (many waves per cube), a thin far shell (cubes as wide as the mesh: unusable), the symmetry axis
    pts = torch.cat([torch.rand((3_000_000, 3), device=dev, generator=g).mul_(2).sub_(1),
                     torch.randn((200_000, 3), device=dev, generator=g).mul_(6.0),
                     torch.randn((50_000, 3), device=dev, generator=g).mul_(torch.tensor([1e-3, 1e-3, 0.3], device=dev))])
    r = _both(s, pts)
    assert bool((r[0]["nn"] == r[1]["nn"]).all())
    assert bool((r[0]["sdf"].view(torch.int32) == r[1]["sdf"].view(torch.int32)).all())
    k = torch.randint(0, pts.shape[0], (60_000,), device=dev, generator=g)
    hp = pts[k].cpu().numpy()
    o = po.OracleSDF(v, f, True, R)
    assert (r[1]["nn"][k].cpu().numpy() == o.nn(hp)).all()
    assert same_float(r[1]["sdf"][k].cpu().numpy(), o(hp)).all()


def test_lists_with_outliers_beyond_the_sampled_bounds_nan_points_and_updates():
    """launches of >= 8M points take their bounds from a 1/16 sample: queries outside them are clamped into the
    border cubes by index and must be caught by the wave's own containment check; non-finite points; then the
    mesh moves and shrinks (update()) and the same queries are asked again"""
    import torch
    v, f = meshes.icosphere(5, radius=0.8, noise=0.02, seed=1)
    R = meshes.random_rotation(5)
    s = SDF(v, f, robust=True, copy=True)
    s.set_raycast_axes(R)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    n = 9_000_000
    pts = torch.rand((n, 3), device=dev, generator=g).mul_(1.6).sub_(0.8)
    # outliers at positions the strided sample (runs of 256 points, one run in 16) does not read
    idx = torch.arange(300, n, 4096 * 7, device=dev)
    pts[idx] = torch.randn((idx.shape[0], 3), device=dev, generator=g) * 3.0 + 2.0
    pts[idx[::5], 1] = float("nan")
    pts[idx[1::5], 0] = float("inf")
    for frame in range(2):
        if frame:
            s.vertices_mutable = (v * np.float32(0.5) + np.float32([0.2, -0.1, 0.05])).astype(np.float32)
            s.update()
        r = _both(s, pts)
        assert bool((r[0]["nn"] == r[1]["nn"]).all())
        a, b = r[0]["sdf"], r[1]["sdf"]
        assert bool(((a.view(torch.int32) == b.view(torch.int32)) | (a.isnan() & b.isnan())).all())
        fin = torch.isfinite(pts).all(dim=1)
        k = torch.cat([idx[fin[idx]][:2000], torch.randint(0, n, (40_000,), device=dev, generator=g)])
        k = k[fin[k]]
        hp = pts[k].cpu().numpy()
        o = po.OracleSDF(np.asarray(s.verts), f, True, R)
        assert (r[1]["nn"][k].cpu().numpy() == o.nn(hp)).all()
        assert same_float(b[k].cpu().numpy(), o(hp)).all()


def test_lists_on_a_tie_lattice_and_fast_nn():
    import torch
    rng = np.random.default_rng(2)
    g = np.stack(np.meshgrid(*[np.arange(-6, 7)] * 3, indexing="ij"), -1).reshape(-1, 3)
    v = (g[rng.permutation(g.shape[0])[:1500]] * 0.125).astype(np.float32)
    f = rng.integers(0, v.shape[0], (3000, 3)).astype(np.uint32)
    q = np.concatenate([rng.integers(-14, 15, (600_000, 3)) * 0.0625, rng.uniform(-1, 1, (600_000, 3))]).astype(np.float32)
    qt = torch.from_numpy(q).cuda()
    o = po.OracleSDF(v, f, False)
    ref_nn, ref_d = o.nn(q[:150_000]), o(q[:150_000])
    for approx in (0, 1):
        s = SDF(v, f, robust=False)
        s.set_option("approx", approx)
        for level in (3, 5):
            s.set_option("cube_level", level)
            r = _both(s, qt)
            assert bool((r[0]["nn"] == r[1]["nn"]).all())
            assert (r[1]["nn"][:150_000].cpu().numpy() == ref_nn).all()
            assert same_float(r[1]["sdf"][:150_000].cpu().numpy(), ref_d).all()
    # fast_nn (GPU-built tree, exact walk, lowest-position tie rule): lists on == lists off
    v2, f2 = meshes.torus(120, 120)
    s = SDF(v2, f2, robust=True)
    s.set_option("fast_nn", 1)
    s.update()
    p = torch.rand((2_000_000, 3), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)).mul_(2).sub_(1)
    s.set_option("cube_level", 4)
    r = _both(s, p)
    assert bool((r[0]["nn"] == r[1]["nn"]).all())
    assert bool((r[0]["sdf"].view(torch.int32) == r[1]["sdf"].view(torch.int32)).all())


CHILD = r"""
import ctypes as C, json, sys
import numpy as np, torch
sys.path.insert(0, %r)
import sdf_amd
from sdf_amd import meshes
dev = torch.device("cuda", 0)
v, f = meshes.torus(200, 200)
s = sdf_amd.SDF(v, f, robust=False)
L = s._L
st = (C.c_ulonglong * 12)()
out = {}
for level, n in ((4, 4_000_000), (5, 12_000_000)):
    pts = torch.rand((n, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(level)).mul_(2).sub_(1)
    s.set_option("cube_level", level)
    res = {}
    for lists in (0, 1):
        s.set_option("cube_lists", lists)
        L.sdfhip_debug_packet_stats(st, 1)
        res[lists] = s.nn(pts)
        torch.cuda.synchronize()
        L.sdfhip_debug_packet_stats(st, 1)
        out["level %%d lists %%d" %% (level, lists)] = {"waves": int(st[0]), "list_waves": int(st[11]), "wide": int(st[1]), "groups": int(st[2])}
    out["level %%d equal" %% level] = bool((res[0] == res[1]).all())
print("RESULT " + json.dumps(out))
""" % ROOT


def test_most_waves_of_a_dense_launch_take_their_cube_list():
    if not os.path.exists(STATS):
        pytest.skip("libsdfhip_stats.so not built (make -C sdf_amd/csrc stats)")
    env = dict(os.environ, SDFHIP_LIB=STATS)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    print(res)
    for level in (4, 5):
        assert res[f"level {level} equal"]
        off, on = res[f"level {level} lists 0"], res[f"level {level} lists 1"]
        assert off["list_waves"] == 0
        assert on["list_waves"] > 0.5 * on["waves"], res     # the lists are really used ...
        assert on["wide"] < 0.6 * off["wide"], res            # ... and replace most of the wide box tests
