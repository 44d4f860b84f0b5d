"""The allocation-free entry point (include/sdfhip.h: sdfhip_sdf_device_noalloc) against the entry point
it is a variant of, sdfhip_sdf_device: same results (bit for bit, NaN == NaN), through the C ABI, on two
calls in a row, the second with a smaller input — the static scratch keeps the first call's contents."""
import numpy as np
import pytest

from conftest import same_float

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def _both(s, pts_t, trunc=0):
    torch = _torch()
    L = s._L
    n = pts_t.shape[0]
    stream = torch.cuda.current_stream().cuda_stream
    ref = torch.full((n,), 7.0, dtype=torch.float32, device=pts_t.device)
    out = torch.full((n,), -7.0, dtype=torch.float32, device=pts_t.device)
    assert L.sdfhip_sdf_device(s._h, pts_t.data_ptr(), n, trunc, ref.data_ptr(), stream) == 0, L.sdfhip_last_error()
    assert L.sdfhip_sdf_device_noalloc(s._h, pts_t.data_ptr(), n, trunc, out.data_ptr(), stream) == 0, \
        L.sdfhip_last_error()
    torch.cuda.synchronize()
    return ref.cpu().numpy(), out.cpu().numpy()


@pytest.mark.parametrize("robust", [True, False])
def test_noalloc_entry_matches_sdf_device_on_two_calls_in_a_row(robust):
    torch = _torch()
    import sdf_amd
    from sdf_amd import meshes
    v, f = meshes.torus(83, 83)
    s = sdf_amd.SDF(v, f, robust=robust)
    if robust:
        s.set_raycast_axes(meshes.random_rotation(1234))
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(5)
    big = torch.rand((300_000, 3), device=dev, generator=g).mul_(2).sub_(1)      # sorted + packet pass
    small = torch.rand((50_000, 3), device=dev, generator=g).mul_(2.4).sub_(1.2)  # smaller, partly outside
    for pts, trunc in ((big, 0), (small, 1)):
        ref, out = _both(s, pts, trunc)
        assert same_float(ref, out).all()
    # a tiny batch (strict kernel / scan path) after the large ones, and the oracle as a second opinion
    tiny = small[:777].contiguous()
    ref, out = _both(s, tiny)
    assert same_float(ref, out).all()
    from oracle import pyoracle as po
    o = po.OracleSDF(v, f, robust, s.raycast_axes if robust else None)
    assert same_float(o(tiny.cpu().numpy()), out).all()


def test_noalloc_entry_rejects_batches_beyond_its_limit():
    torch = _torch()
    import sdf_amd
    from sdf_amd import meshes
    v, f = meshes.cube()
    s = sdf_amd.SDF(v, f)
    L = s._L
    n = (1 << 20) + 1
    pts = torch.zeros((n, 3), device="cuda")
    out = torch.zeros(n, device="cuda")
    rc = L.sdfhip_sdf_device_noalloc(s._h, pts.data_ptr(), n, 0, out.data_ptr(), None)
    assert rc == -1 and b"at most" in L.sdfhip_last_error()
    # exactly the limit is served, and equals the allocating entry point
    ref, res = _both(s, pts[: 1 << 20].contiguous())
    assert same_float(ref, res).all()
