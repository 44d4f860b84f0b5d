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


def test_noalloc_entry_under_hip_graph_capture_and_with_profiling_on():
    """the callers it is for: captured into a HIP graph (torch.cuda.graph: capture_error_mode "global" turns
    any hipMalloc / event creation during the capture into an error), replayed twice on new inputs, against
    sdfhip_sdf_device; option "profile" on must not make it create events"""
    torch = _torch()
    import sdf_amd
    from sdf_amd import meshes
    v, f = meshes.torus(83, 83)
    s = sdf_amd.SDF(v, f, robust=True)
    s.set_raycast_axes(meshes.random_rotation(8))
    s.set_option("profile", 1)
    L = s._L
    dev = torch.device("cuda", 0)
    n = 200_000
    gen = torch.Generator(device=dev).manual_seed(9)
    pts = torch.rand((n, 3), device=dev, generator=gen).mul_(2).sub_(1)
    out = torch.zeros(n, dtype=torch.float32, device=dev)
    # warm-up outside the capture (first-call lazy state of the runtime), on the capture stream
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        assert L.sdfhip_sdf_device_noalloc(s._h, pts.data_ptr(), n, 0, out.data_ptr(), side.cuda_stream) == 0
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        rc = L.sdfhip_sdf_device_noalloc(s._h, pts.data_ptr(), n, 0, out.data_ptr(),
                                         torch.cuda.current_stream(dev).cuda_stream)
        assert rc == 0, L.sdfhip_last_error()
    launches_before = s.stats["query_kernel_launches"]
    for seed in (1, 2):
        new = torch.rand((n, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(seed)).mul_(2.2).sub_(1.1)
        pts.copy_(new)
        out.fill_(-3.0)
        graph.replay()
        torch.cuda.synchronize()
        ref = torch.empty_like(out)
        assert L.sdfhip_sdf_device(s._h, pts.data_ptr(), n, 0, ref.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert same_float(ref.cpu().numpy(), out.cpu().numpy()).all()
    # profiling timed the two sdfhip_sdf_device calls only: the no-alloc entry created no event pair
    assert s.stats["query_kernel_launches"] - launches_before == 2
    # and the handle still serves ordinary large calls afterwards (its own workspace is back)
    big = torch.rand((2_000_000, 3), device=dev, generator=gen).mul_(2).sub_(1)
    assert bool(torch.isfinite(s(big)).all())
