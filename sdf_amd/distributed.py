"""Multi-GPU layer: one process per GPU, query points sharded by contiguous index range,
mesh replicated by a broadcast (RCCL over xGMI when the backend is "nccl").

The reference has no distributed code; its only parallelism is a thread pool
over point indices (util.cpp:7-29).  Points are independent, so the path shards with NO
data-path collective: the only exchange is the one-off replication of the mesh (verts,
faces) and of the ray-cast rotation R — R must be the same on every rank or shards would
disagree on edge hits (sdf.cpp:43-48 draws it at random).  Every rank then builds the same
deterministic structures from the same bytes.  An all-gather of results is offered for
callers that want the full array everywhere; no benchmark configuration needs it.
"""
from __future__ import annotations

import numpy as np


def shard_range(n: int, rank: int, world: int):
    """Contiguous range of rank `rank`: [r*ceil(n/w), min(n, (r+1)*ceil(n/w)))."""
    per = -(-n // world) if world > 0 else n
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def _bcast_tensor(arr, shape_dtype, src, dist, device):
    """Broadcast an array (known on `src`; shape/dtype known everywhere) and return the received torch
    tensor ON `device` — with the "nccl" backend that is this rank's GPU, and the bytes came over xGMI."""
    import torch
    shape, dtype = shape_dtype
    tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.int32): torch.int32}[np.dtype(dtype)]
    if dist.get_rank() == src:
        if hasattr(arr, "data_ptr"):  # a torch tensor, possibly already on the device
            t = arr.detach().to(device=device, dtype=tdt).contiguous()
        else:
            t = torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype)).to(device)
    else:
        t = torch.empty(shape, dtype=tdt, device=device)
    dist.broadcast(t, src)
    return t


def _bcast_array(arr, shape_dtype, src, dist, device):
    """Broadcast a numpy array (known on `src`, shape/dtype known everywhere) via torch."""
    return _bcast_tensor(arr, shape_dtype, src, dist, device).cpu().numpy()


def _bcast_header(values, src, dist, device, error=None):
    """One int64 header from `src`.  `error` (a string, src only) turns it into a poisoned header: EVERY rank
    raises ValueError together, before any payload broadcast — a caller error on src must not leave the other
    ranks waiting in a collective."""
    import torch
    hdr = torch.zeros(len(values) + 1, dtype=torch.int64, device=device)
    if dist.get_rank() == src:
        hdr[0] = -1 if error else 0
        for k, x in enumerate(values):
            hdr[k + 1] = int(x)
    dist.broadcast(hdr, src)
    out = [int(x) for x in hdr.cpu()]
    if out[0] != 0:
        raise ValueError(error if dist.get_rank() == src else f"rank {src} passed invalid arguments")
    return out[1:]


def _check_mesh(verts, faces, R):
    """argument errors of the source rank as a string (None: fine) — checked BEFORE the first collective"""
    if verts is None or faces is None:
        return "the source rank must pass verts and faces"
    vs, fs = tuple(np.shape(verts)), tuple(np.shape(faces))
    if len(vs) != 2 or vs[1] != 3 or len(fs) != 2 or fs[1] != 3:
        return f"verts must be (V, 3) and faces (F, 3), got {vs} and {fs}"
    if R is not None and int(np.size(R)) != 9:
        return f"R must have 9 entries, got shape {tuple(np.shape(R))}"
    return None


def broadcast_mesh(verts, faces, R=None, src: int = 0, device="cpu"):
    """Replicate (verts, faces, R) from rank `src` to every rank of the default process group.
    Non-source ranks may pass None.  Returns numpy (verts f32 [V,3], faces u32 [F,3], R f32 [3,3])."""
    import torch.distributed as dist
    rank = dist.get_rank()
    err = _check_mesh(verts, faces, R) if rank == src else None
    vals = (0, 0, 0) if (rank != src or err) else (np.shape(verts)[0], np.shape(faces)[0], 0 if R is None else 1)
    nv, nf, has_r = _bcast_header(vals, src, dist, device, err)
    v = _bcast_array(verts, ((nv, 3), np.float32), src, dist, device)
    f = _bcast_array(None if faces is None else np.asarray(faces).astype(np.int32),
                     ((nf, 3), np.int32), src, dist, device).astype(np.uint32)
    r = None
    if has_r:
        r = _bcast_array(None if R is None else np.asarray(R, dtype=np.float32).reshape(3, 3),
                         ((3, 3), np.float32), src, dist, device)
    return v, f, r


class ShardedSDF:
    """`SDF` replicated on every rank; queries take the FULL point array (or this rank's shard)
    and evaluate only the rank's index range.

    engine_factory(verts, faces, robust) -> object with __call__/contains/nn and, for robust
    parity, set_raycast_axes(R).  Default: sdf_amd.SDF on this rank's GPU.  (Tests inject a CPU
    checker here to exercise the sharding/replication logic under gloo.)"""

    def __init__(self, verts=None, faces=None, robust=True, R=None, src=0, device=None,
                 engine_factory=None):
        import torch.distributed as dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if dist.get_backend() == "gloo":
            bdev = "cpu"
        else:  # one process per GPU: broadcast through, and build on, THIS rank's device
            import torch
            if device is None:
                device = torch.cuda.current_device()
            bdev = f"cuda:{int(device)}"
        self._bdev, self._src, self.robust = bdev, src, bool(robust)
        if robust and R is None and self.rank == src:
            from .meshes import random_rotation
            R = random_rotation(int(np.random.SeedSequence().entropy % (2 ** 31)))
        self.verts, self.faces, self.R = broadcast_mesh(verts, faces, R if robust else None, src, bdev)
        if engine_factory is None:
            from .pysdf import SDF
            self.engine = SDF(self.verts, self.faces, robust=robust, copy=True, device=device)
        else:
            self.engine = engine_factory(self.verts, self.faces, robust)
        if robust and self.R is not None and hasattr(self.engine, "set_raycast_axes"):
            self.engine.set_raycast_axes(self.R)

    def update(self, verts=None, R=None):
        """SDF::update() across the ranks (sdf.cpp:391, 31-73; the deforming-mesh loop of demo.cpp:109-121):
        rank `src` passes the new vertex positions (None: unchanged; numpy, or a torch tensor that may already
        live on the GPU) and, for a robust mesh, optionally a rotation; by default it draws a fresh one, as the
        reference's update() does.  Both are broadcast and every rank rebuilds with the SAME R.  With the
        "nccl" backend the vertices arrive in this rank's HBM over xGMI and go straight into the engine
        (`update_from_device`): no host round trip in this layer.  Argument errors on `src` raise ValueError
        on every rank, before any payload is broadcast."""
        import torch.distributed as dist
        err, has_v = None, 0
        if self.rank == self._src:
            if verts is not None:
                has_v = 1
                if tuple(verts.shape) != tuple(self.verts.shape):
                    err = (f"update() cannot change the number of vertices: got {tuple(verts.shape)}, "
                           f"the mesh has {tuple(self.verts.shape)}")
            if R is not None and int(np.size(R)) != 9:
                err = f"R must have 9 entries, got shape {tuple(np.shape(R))}"
            if self.robust and R is None:
                from .meshes import random_rotation
                R = random_rotation(int(np.random.SeedSequence().entropy % (2 ** 31)))
        has_v, has_r = _bcast_header((has_v, 1 if self.robust else 0), self._src, dist, self._bdev, err)
        if has_r:
            self.R = _bcast_array(None if R is None else np.asarray(R, dtype=np.float32).reshape(3, 3),
                                  ((3, 3), np.float32), self._src, dist, self._bdev)
            if hasattr(self.engine, "set_raycast_axes"):
                try:
                    self.engine.set_raycast_axes(self.R, pin=2)  # takes effect at the update below
                except TypeError:
                    self.engine.set_raycast_axes(self.R)
        if has_v:
            t = _bcast_tensor(verts, (self.verts.shape, np.float32), self._src, dist, self._bdev)
            self._verts_t = t
            if t.is_cuda and hasattr(self.engine, "update_from_device"):
                self._verts_host = None  # (the `verts` property fetches the engine's refreshed host copy)
                self.engine.update_from_device(t)  # device tensor -> sdfhip_update_from_device
                return
            self.verts = t.numpy()  # a CPU tensor (gloo): shares its memory, no copy
            self.engine.vertices_mutable[:] = self.verts
        self.engine.update()

    @property
    def verts(self):
        if self._verts_host is None:
            self._verts_host = np.array(self.engine.verts, dtype=np.float32)
        return self._verts_host

    @verts.setter
    def verts(self, v):
        self._verts_host = v

    def local_range(self, n):
        return shard_range(n, self.rank, self.world)

    def _local(self, fn, points, *a):
        lo, hi = self.local_range(points.shape[0])
        return lo, hi, fn(points[lo:hi], *a)

    def sdf_local(self, points, trunc_aabb=False):
        return self._local(self.engine.__call__, points, trunc_aabb)

    def contains_local(self, points):
        return self._local(self.engine.contains, points)

    def nn_local(self, points):
        return self._local(self.engine.nn, points)

    def gather(self, n, lo, hi, local, dtype):
        """Optional: assemble the full result on every rank (all_gather of padded shards)."""
        import torch
        import torch.distributed as dist
        per = -(-n // self.world)
        buf = torch.zeros(per, dtype=dtype)
        loc = (local.detach().cpu() if hasattr(local, "detach") else torch.as_tensor(np.asarray(local))).to(dtype)
        buf[: hi - lo] = loc
        outs = [torch.zeros(per, dtype=dtype) for _ in range(self.world)]
        if self._bdev != "cpu":  # the device this object broadcasts through, not whatever is current
            buf = buf.to(self._bdev)
            outs = [o.to(self._bdev) for o in outs]
        dist.all_gather(outs, buf)
        return torch.cat([o.cpu() for o in outs])[:n].numpy()
