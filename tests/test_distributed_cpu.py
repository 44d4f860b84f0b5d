"""world_size-2 gloo test of the multi-GPU layer (sharding by index range + mesh/R replication).
The compute engine is injected: on CPU the oracle stands in as the checker, so this covers the
host logic of the N>1 path; the GPU engine itself is covered by the -m gpu tests."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import pyoracle as po
    from sdf_amd import meshes
    from sdf_amd.distributed import ShardedSDF, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == 0:
        v, f = meshes.torus(40, 40)
        R = meshes.random_rotation(99)
    else:
        v = f = R = None

    class Engine(po.OracleSDF):
        def set_raycast_axes(self, R):
            self.update(R)

    s = ShardedSDF(v, f, robust=True, R=R, engine_factory=lambda vv, ff, rb: Engine(vv, ff, rb))
    n = 10007  # ragged: not divisible by world
    pts = np.random.default_rng(5).uniform(-1, 1, (n, 3)).astype(np.float32)
    lo, hi, d = s.sdf_local(pts)
    _, _, c = s.contains_local(pts)
    _, _, nn = s.nn_local(pts)
    assert (lo, hi) == shard_range(n, rank, world)
    full_d = s.gather(n, lo, hi, d, torch.float32)
    full_nn = s.gather(n, lo, hi, nn, torch.int32)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), lo=lo, hi=hi, d=d, c=c, nn=nn, full_d=full_d,
             full_nn=full_nn, verts=s.verts, R=s.R)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    from sdf_amd.distributed import shard_range
    for n in (0, 1, 7, 64, 1000, 10007):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert all(hi - lo <= -(-n // w) for lo, hi in r)


def test_two_rank_gloo_sharded_queries_match_single_process(tmp_path):
    from oracle import pyoracle as po
    from sdf_amd import meshes
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    v, f = meshes.torus(40, 40)
    R = meshes.random_rotation(99)
    o = po.OracleSDF(v, f, True, R)
    pts = np.random.default_rng(5).uniform(-1, 1, (10007, 3)).astype(np.float32)
    ref_d, ref_c, ref_nn = o(pts), o.contains(pts), o.nn(pts)
    got = [np.load(os.path.join(tmp_path, f"r{k}.npz")) for k in range(world)]
    assert int(got[0]["lo"]) == 0 and int(got[0]["hi"]) == int(got[1]["lo"]) and int(got[1]["hi"]) == 10007
    for g in got:
        lo, hi = int(g["lo"]), int(g["hi"])
        assert (g["verts"] == v).all() and (g["R"] == R).all()          # replicated bytes
        assert (g["d"].view(np.uint32) == ref_d[lo:hi].view(np.uint32)).all()
        assert (g["c"] == ref_c[lo:hi]).all() and (g["nn"] == ref_nn[lo:hi]).all()
        assert (g["full_d"].view(np.uint32) == ref_d.view(np.uint32)).all()
        assert (g["full_nn"] == ref_nn).all()
