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
        def set_raycast_axes(self, R, pin=1):
            self.R = np.ascontiguousarray(R, dtype=np.float32).reshape(3, 3)
            if pin != 2:
                po.OracleSDF.update(self, self.R)

        @property
        def vertices_mutable(self):
            return self.verts  # the oracle reads this array again on update(), like Eigen::Ref

        def update(self):
            po.OracleSDF.update(self, self.R)

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
    # deforming mesh (BASELINE config 5, demo.cpp:109-121): rank 0 moves the vertices, update() replicates
    # them and a rotation (frame 1: drawn by rank 0 at random, frames 2-3: given), every rank rebuilds
    v0 = s.verts.copy()
    frames = {}
    for t in (1, 2, 3):
        vt = (v0 * (1 + 0.05 * np.sin(2 * np.pi * t / 100 + 4 * v0[:, 1:2]))).astype(np.float32) if rank == 0 else None
        s.update(vt, None if t == 1 else (meshes.random_rotation(100 + t) if rank == 0 else None))
        lo, hi, ct = s.contains_local(pts)
        _, _, nt = s.nn_local(pts)
        _, _, dt = s.sdf_local(pts)
        frames[f"c{t}"], frames[f"n{t}"], frames[f"d{t}"] = ct, nt, dt
        frames[f"v{t}"], frames[f"R{t}"] = s.verts.copy(), s.R.copy()
    # a caller error on the source rank (wrong vertex count, malformed R) raises on EVERY rank before any
    # payload is broadcast — no rank is left waiting in a collective — and the object stays usable
    bad = 0
    for kw in (dict(verts=np.zeros((7, 3), np.float32) if rank == 0 else None),
               dict(R=np.eye(4, dtype=np.float32) if rank == 0 else None)):
        try:
            s.update(**kw)
        except ValueError:
            bad += 1
    s.update(v0 if rank == 0 else None, meshes.random_rotation(7) if rank == 0 else None)
    _, _, c_after = s.contains_local(pts)
    frames["bad_raised"], frames["c_after"] = bad, c_after
    np.savez(os.path.join(outdir, f"f{rank}.npz"), **frames)
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
    # the three deformation frames: same vertices and R on both ranks, results = a single-process oracle
    fr = [np.load(os.path.join(tmp_path, f"f{k}.npz")) for k in range(world)]
    for t in (1, 2, 3):
        vt = (v * (1 + 0.05 * np.sin(2 * np.pi * t / 100 + 4 * v[:, 1:2]))).astype(np.float32)
        assert (fr[0][f"v{t}"] == vt).all() and (fr[1][f"v{t}"] == vt).all()
        assert (fr[0][f"R{t}"] == fr[1][f"R{t}"]).all()
        if t > 1:
            assert (fr[0][f"R{t}"] == meshes.random_rotation(100 + t)).all()
        ot = po.OracleSDF(vt, f, True, fr[0][f"R{t}"])
        rc, rn, rd = ot.contains(pts), ot.nn(pts), ot(pts)
        for k, g in enumerate(got):
            lo, hi = int(g["lo"]), int(g["hi"])
            assert (fr[k][f"c{t}"] == rc[lo:hi]).all() and (fr[k][f"n{t}"] == rn[lo:hi]).all()
            assert (fr[k][f"d{t}"].view(np.uint32) == rd[lo:hi].view(np.uint32)).all()
    oa = po.OracleSDF(v, f, True, meshes.random_rotation(7)).contains(pts)
    for k, g in enumerate(got):
        assert int(fr[k]["bad_raised"]) == 2  # both bad calls raised on both ranks, nobody hung
        assert (fr[k]["c_after"] == oa[int(g["lo"]):int(g["hi"])]).all()


def test_bench_gpus_flag_starts_that_many_ranks(tmp_path):
    """`python bench.py --gpus N` must measure N ranks by itself: the parent starts the
    ranks before anything touches the GPU.  Without a GPU every rank stops at bench.py's "needs a GPU" line —
    after the launch, which is the part under test: the flag decides the world size."""
    import subprocess
    env = dict(os.environ, SDFHIP_BENCH_ECHO_WORLD=str(tmp_path))
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--one-device", "--points", "1000", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    seen = sorted(os.listdir(tmp_path))
    assert seen == ["rank0_of_2", "rank1_of_2"], (seen, r.stderr[-2000:])
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs a GPU" in r.stderr  # a failing rank fails the command
