"""The leaf-box form of the packet pass's path check (kd_path_clear, kernels_packet.h) must never pass a
query whose full path walk (kd_path_max_bound: nanoflann.hpp:1396-1399 replayed with nanoflann's own
arithmetic) fails.  The tuning build (`make -C sdf_amd/csrc stats`, libsdfhip_stats.so) runs both for
every near-tie lane and counts the disagreements; it is loaded in a child process so that the product
library of this process stays the one under test everywhere else."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STATS = os.path.join(ROOT, "sdf_amd", "libsdfhip_stats.so")

CHILD = r"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, %r)
import sdf_amd
from sdf_amd import meshes
rng = np.random.default_rng(5)
def lattice():
    g = np.stack(np.meshgrid(*[np.arange(-6, 7)] * 3, indexing="ij"), -1).reshape(-1, 3)
    v = (g[rng.permutation(g.shape[0])[:1500]] * 0.125).astype(np.float32)
    return v, np.array([[0, 1, 2], [3, 4, 5]], np.uint32)
def far(n, r0, r1):
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return (d * rng.uniform(r0, r1, (n, 1))).astype(np.float32)
tv, tf = meshes.torus(200, 200)
iv, iF = meshes.icosphere(5, radius=0.8)
lv, lf = lattice()
cases = {
    "torus, box + far field": (tv, tf, np.concatenate([rng.uniform(-1, 1, (600000, 3)).astype(np.float32), far(400000, 2, 50)])),
    "torus at (100, -50, 25)": (tv + np.float32([100, -50, 25]), tf, (rng.uniform(-1.5, 1.5, (600000, 3)) + [100, -50, 25]).astype(np.float32)),
    "sphere, centre + far field": (iv, iF, np.concatenate([rng.normal(0, 1e-3, (200000, 3)).astype(np.float32), far(400000, 1, 1e3)])),
    "tie lattice": (lv, lf, np.concatenate([rng.integers(-14, 15, (300000, 3)) * 0.0625, rng.uniform(-2, 2, (300000, 3))]).astype(np.float32)),
    "one-leaf tree": (tv[:7].copy(), np.array([[0, 1, 2]], np.uint32), rng.uniform(-2, 2, (300000, 3)).astype(np.float32)),
}
out = {}
for name, (v, f, q) in cases.items():
    s = sdf_amd.SDF(v, f, robust=False)
    s.set_option("scan_max", 0)  # (the packet pass, not the exact scan of small work products)
    st = (C.c_ulonglong * 12)()
    s._L.sdfhip_debug_packet_stats(st, 1)
    s.nn(q)
    s._L.sdfhip_debug_packet_stats(st, 1)
    out[name] = {"near_tie_lanes": int(st[6]), "box_passes": int(st[8]), "box_pass_walk_fail": int(st[9]), "walks": int(st[10])}
    if name.startswith("torus, box"):  # the boxes follow update(): same queries against the moved, shrunk mesh
        s.vertices_mutable = (v * np.float32(0.6) + np.float32([0.3, -0.2, 0.1])).astype(np.float32)
        s.update()
        s.nn(q)
        s._L.sdfhip_debug_packet_stats(st, 1)
        out[name + ", after update()"] = {"near_tie_lanes": int(st[6]), "box_passes": int(st[8]), "box_pass_walk_fail": int(st[9]), "walks": int(st[10])}
print("RESULT " + json.dumps(out))
""" % ROOT


@pytest.mark.gpu
def test_leaf_box_path_check_never_passes_what_the_walk_rejects():
    if not os.path.exists(STATS):
        pytest.skip("libsdfhip_stats.so not built (make -C sdf_amd/csrc stats)")
    env = dict(os.environ, SDFHIP_LIB=STATS)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    checked = 0
    for name, c in res.items():
        assert c["box_pass_walk_fail"] == 0, (name, c)
        checked += c["box_passes"] + c["walks"]
    # the cases do reach the check, through both outcomes of the quick test
    assert checked > 10000, res
    assert sum(c["box_passes"] for c in res.values()) > 1000 and sum(c["walks"] for c in res.values()) > 100, res
