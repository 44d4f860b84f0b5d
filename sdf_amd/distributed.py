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


def _bcast_array(arr, shape_dtype, src, dist, device):
    """Broadcast a numpy array (known on `src`, shape/dtype known everywhere) via torch."""
    import torch
    shape, dtype = shape_dtype
    tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.int32): torch.int32}[np.dtype(dtype)]
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype)).to(device)
    else:
        t = torch.empty(shape, dtype=tdt, device=device)
    dist.broadcast(t, src)
    return t.cpu().numpy()


def broadcast_mesh(verts, faces, R=None, src: int = 0, device="cpu"):
    """Replicate (verts, faces, R) from rank `src` to every rank of the default process group.
    Non-source ranks may pass None.  Returns numpy (verts f32 [V,3], faces u32 [F,3], R f32 [3,3])."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    hdr = torch.zeros(3, dtype=torch.int64, device=device)
    if rank == src:
        hdr[0], hdr[1], hdr[2] = verts.shape[0], faces.shape[0], 0 if R is None else 1
    dist.broadcast(hdr, src)
    nv, nf, has_r = (int(x) for x in hdr.cpu())
    v = _bcast_array(verts, ((nv, 3), np.float32), src, dist, device)
    f = _bcast_array(None if faces is None else np.asarray(faces).astype(np.int32),
                     ((nf, 3), np.int32), src, dist, device).astype(np.uint32)
    r = None
    if has_r:
        r = _bcast_array(R, ((3, 3), np.float32), src, dist, device)
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
        rank `src` passes the new vertex positions (None: unchanged) and, for a robust mesh, optionally a
        rotation; by default it draws a fresh one, as the reference's update() does.  Both are broadcast,
        every rank writes the vertices into its engine (`vertices_mutable`) and rebuilds with the SAME R."""
        import torch
        import torch.distributed as dist
        hdr = torch.zeros(2, dtype=torch.int64, device=self._bdev)
        if self.rank == self._src:
            if self.robust and R is None:
                from .meshes import random_rotation
                R = random_rotation(int(np.random.SeedSequence().entropy % (2 ** 31)))
            hdr[0], hdr[1] = 0 if verts is None else 1, 1 if self.robust else 0
        dist.broadcast(hdr, self._src)
        has_v, has_r = (int(x) for x in hdr.cpu())
        if has_v:
            if self.rank == self._src and np.asarray(verts).shape != self.verts.shape:
                raise ValueError("update() cannot change the number of vertices")
            self.verts = _bcast_array(verts, (self.verts.shape, np.float32), self._src, dist, self._bdev)
            self.engine.vertices_mutable[:] = self.verts
        if has_r:
            self.R = _bcast_array(R, ((3, 3), np.float32), self._src, dist, self._bdev)
            if hasattr(self.engine, "set_raycast_axes"):
                try:
                    self.engine.set_raycast_axes(self.R, pin=2)  # takes effect at the update below
                except TypeError:
                    self.engine.set_raycast_axes(self.R)
        self.engine.update()

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
        if dist.get_backend() != "gloo":
            dev = f"cuda:{torch.cuda.current_device()}"
            buf = buf.to(dev)
            outs = [o.to(dev) for o in outs]
        dist.all_gather(outs, buf)
        return torch.cat([o.cpu() for o in outs])[:n].numpy()
