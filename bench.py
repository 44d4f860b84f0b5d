#!/usr/bin/env python3
"""bench.py — robust SDF evals/s on the ~360k-face configuration of BASELINE.json.

Workload (BASELINE.json configs[3], the one the metric is quoted on): synthetic torus
424 x 424 (179,776 verts / 359,552 faces), robust `SDF.__call__` on uniform random points in
[-1,1]^3 generated ON DEVICE (seed 7 + rank), 1e9 / 8 = 125,000,000 points per GPU per step
(weak scaling: every rank processes its own shard; no data-path collective).  The mesh and the
ray-cast rotation are created on rank 0 and replicated with an RCCL broadcast; every rank then
builds the same structures.  One "step" = one pass of the hot path over the rank's batch, with
inputs resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W          # starts its N ranks itself (see spawn_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import sdf_amd  # noqa: E402
from sdf_amd import meshes  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_EVAL = 16    # 12 B point read + 4 B result write


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=125_000_000, help="points per GPU per step")
    ap.add_argument("--mesh-res", type=int, default=424, help="torus tessellation (424 -> 359,552 faces)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grid", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="skip the 100-frame deforming-mesh split (configs[4]) in `extra`")
    ap.add_argument("--c5-frames", type=int, default=100)
    ap.add_argument("--no-sort", action="store_true")
    # rehearsal of the N>1 path on a box with one GPU: every rank on cuda:0, mesh broadcast over gloo
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--one-device", action="store_true")
    return ap.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process becomes the parent of
    N ranks started through torch.distributed.run — as a CHILD process, and before anything here has touched
    the GPU runtime (`import torch` and `torch.cuda.device_count()` do not).  The ranks inherit stdout, so
    rank 0's single JSON line is this command's output; the exit code is the launcher's (non-zero if any
    rank failed)."""
    if not a.one_device:
        have = torch.cuda.device_count()
        if have < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: this node shows {have} GPU(s) "
                             "(--one-device --backend gloo rehearses the N-rank path on one card)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
    return subprocess.run(cmd, env=env).returncode


def c5_split(v0, f, R, dev, frames, fast_nn):
    """BASELINE.json configs[4]: `frames` frames of vertices_mutable + update() + contains() + nn() on the same
    4M device-resident seeded points (demo.cpp:109-121 is the reference's only update loop).
    Synchronous frames give the per-frame split; the same frames issued back to back give the pipelined rate."""
    s = sdf_amd.SDF(v0, f, robust=True, copy=True, device=dev.index)
    s.set_raycast_axes(R)
    if fast_nn:
        s.set_option("fast_nn", 1)
    pts = torch.rand((4_000_000, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(7)).mul_(2).sub_(1)
    y4 = 4 * v0[:, 1:2]

    def frame_verts(t):
        return (v0 * (1 + 0.05 * np.sin(2 * np.pi * t / 100 + y4))).astype(np.float32)

    tu = tc = tn = kd = 0.0
    warm = 2
    for t in range(frames + warm):
        vt = frame_verts(t)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        s.vertices_mutable = vt
        s.update()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        c = s.contains(pts)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        k = s.nn(pts)
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        if t >= warm:
            tu += t1 - t0
            tc += t2 - t1
            tn += t3 - t2
            kd += s.stats["last_kd_build_ms"]
    vs = [frame_verts(t) for t in range(min(frames, 20))]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    keep = None
    for vt in vs:
        s.vertices_mutable = vt
        s.update()
        keep = (s.contains(pts), s.nn(pts))
    torch.cuda.synchronize(dev)
    piped = (time.perf_counter() - t0) / len(vs) * 1e3
    del keep, c, k
    return {"frames": frames, "points_per_frame": 4_000_000, "update_ms": tu / frames * 1e3,
            "kd_build_ms": kd / frames, "contains_ms": tc / frames * 1e3, "nn_ms": tn / frames * 1e3,
            "frame_ms": (tu + tc + tn) / frames * 1e3, "pipelined_frame_ms": piped,
            "nn_tie_rule": "lowest (Morton code, index)" if fast_nn else "nanoflann first-visited (reference)"}


def cpu_baseline(v, f, R, seed):
    """The oracle (a port of the reference's CPU path with the same self-scheduling threads,
    util.cpp:7-29) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import pyoracle as po
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit():
        cores = min(cores, int(env))
    threads = max(1, min(cores, 32))  # pybind.cpp:12 caps the default at 32
    o = po.OracleSDF(v, f, True, R)
    rng = np.random.default_rng(seed)
    n = 500_000
    pts = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    t0 = time.perf_counter()
    o(pts, n_threads=threads)
    dt = time.perf_counter() - t0
    rate = n / dt
    n2 = int(min(max(rate * 15.0, n), 80_000_000))
    if n2 > n:
        pts = rng.uniform(-1, 1, (n2, 3)).astype(np.float32)
        t0 = time.perf_counter()
        o(pts, n_threads=threads)
        dt = time.perf_counter() - t0
        n = n2
    return {"value": n / dt, "unit": "evals/s", "cores": threads, "kind": "port",
            # what kind of port: it flatters the CPU against the reference proper
            "variant": "port (grid-filtered rays): ray candidates from a uniform grid over pre-rotated faces "
                       "instead of the reference's R-tree search + per-candidate rotation (sdf.cpp:297-322); "
                       "kd search and face distances as the reference",
            "sample": f"{n} of the step's uniform random points, same mesh and rotation, "
                      f"{dt:.1f} s wall, oracle/liboracle.so with {threads} threads"}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(a))  # (nothing has touched the GPU yet: see spawn_ranks)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}",
              file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SDFHIP_BENCH_ECHO_WORLD"):  # tests: which ranks did the launch produce?
        open(os.path.join(os.environ["SDFHIP_BENCH_ECHO_WORLD"], f"rank{rank}_of_{world}"), "w").close()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    if a.one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    ranks_seen = None  # ranks the collective backend really connected: an all_reduce of ones
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        ones = torch.ones(1, dtype=torch.int64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())

    # ---- mesh: built on rank 0, replicated by RCCL broadcast (sdf_amd/distributed.py) -------
    res = a.mesh_res
    nv, nf = res * res, 2 * res * res
    v = f = R = None
    if rank == 0:
        v, f = meshes.torus(res, res)
        R = meshes.random_rotation(1234)
    if dist is not None:
        # the library's own multi-GPU layer: mesh + R broadcast from rank 0 (RCCL over xGMI with "nccl"),
        # every rank builds the same structures on its GPU
        from sdf_amd.distributed import ShardedSDF
        sh = ShardedSDF(v, f, robust=True, R=R, src=0, device=local)
        s, v, f, R = sh.engine, sh.verts, sh.faces, sh.R
    else:
        s = sdf_amd.SDF(v, f, robust=True, copy=True, device=local)
        s.set_raycast_axes(R)
    if a.no_sort:
        s.set_option("sort_queries", 0)

    # ---- inputs resident in HBM ----------------------------------------------------------------
    n = a.points
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    pts = torch.rand((n, 3), device=dev, generator=g, dtype=torch.float32).mul_(2).sub_(1)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    L = s._L
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        rc = L.sdfhip_sdf_device(s._h, pts.data_ptr(), n, 0, out.data_ptr(), stream)
        if rc:
            raise RuntimeError(L.sdfhip_last_error().decode())

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(dev)
    s.set_option("profile", 2)  # hipEvents around every query-kernel launch, counters reset
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    st = s.stats
    s.set_option("profile", 0)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    total = n * world * a.steps
    value = total / dt
    # dominant kernel: the one the library reports for the last call (sdfhip_stats_t::last_kernel);
    # algorithmic bytes per launch = 16 B x points/launch
    kl = max(1, st["query_kernel_launches"])
    k_ms = st["query_kernel_ms"] / kl
    pts_per_launch = st["query_kernel_points"] / kl
    achieved = BYTES_PER_EVAL * pts_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": {2: "k_query_packet<Q_SDF,robust,exact walk>", 3: "k_query_packet<Q_SDF,robust,approx walk>",
                           1: "k_query<Q_SDF,robust> (strict)", 4: "k_query_heavy<Q_SDF,robust> (scan)"}.get(
                               st.get("last_kernel", 0), "unknown"),
                "kernel_ms_avg": k_ms,
                "points_per_launch": pts_per_launch, "launches": kl,
                "kernel_share_of_step": st["query_kernel_ms"] / (dt * 1e3)}
    # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes of this same
    # command, tools/collect_profiles.sh -> profiles/*_pmc.json); cannot be read live
    try:
        import glob
        def round_of(f):  # profiles/r02_pmc.json -> (2, 1); r02a_pmc.json (mid-round) -> (2, 0)
            tag = os.path.basename(f).split("_")[0]
            digits = "".join(ch for ch in tag if ch.isdigit())
            return (int(digits or 0), 1 if tag[-1:].isdigit() else 0)
        pj = max(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), key=round_of)
        pm = json.load(open(pj))
        if pm.get("traffic_bytes_per_launch") and n == 125_000_000 and not a.no_sort:
            # NOT measured by this run: PMC counters need their own rocprofv3 passes
            roofline["traffic"] = pm["traffic_bytes_per_launch"]
            roofline["traffic_source"] = "stored profile " + os.path.relpath(pj, ROOT)
            for k in ("traffic_low", "traffic_high"):
                if pm.get(k) is not None:
                    roofline[k] = pm[k]
            # VALU utilisation of the same launch from the SQ counters of the same profile
            # (issue slots of 1024 SIMDs at 2 cycles per wave64 instruction; active lanes)
            for k in ("valu_issue_util", "valu_lane_util"):
                if pm.get(k) is not None:
                    roofline[k] = pm[k]
    except (IndexError, OSError, ValueError):
        pass
    extra = {}
    if not a.no_grid:
        # the metric's second figure: ms for a 256^3 grid (16,777,216 points) on the same mesh
        t = torch.linspace(-1, 1, 256, device=dev)
        gp = torch.stack(torch.meshgrid(t, t, t, indexing="ij"), -1).reshape(-1, 3).contiguous()
        go = torch.empty(gp.shape[0], dtype=torch.float32, device=dev)
        ms = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            L.sdfhip_sdf_device(s._h, gp.data_ptr(), gp.shape[0], 0, go.data_ptr(), stream)
            torch.cuda.synchronize(dev)
            ms.append((time.perf_counter() - t1) * 1e3)
        extra["grid256_ms"] = min(ms)
        extra["grid256_inside"] = int((go > 0).sum().item())
    # update() in steady state (the construction-time one also allocates): same vertices, twice
    for _ in range(2):
        s.update()
    extra["update_ms"] = s.stats["last_update_ms"]
    extra["kd_build_ms"] = s.stats["last_kd_build_ms"]
    if world == 1 and not a.no_c5 and a.c5_frames > 0:
        del pts, out
        torch.cuda.empty_cache()
        extra["c5_strict"] = c5_split(v, f, R, dev, a.c5_frames, False)
        extra["c5_fast_nn"] = c5_split(v, f, R, dev, a.c5_frames, True)
    line = {
        "metric": "robust SDF evals/s, ~360k-face mesh",
        "value": value, "unit": "evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[3]: torus {res}x{res} ({nv} verts / {nf} faces), robust "
                               f"SDF on {n} uniform random points in [-1,1]^3 per GPU per step"
                               + (" (1e9/8 per GPU)" if n == 125_000_000 else "") + ", device-resident, seed 7+rank",
                   "points_per_gpu": n, "faces": nf, "verts": nv, "sort_queries": not a.no_sort},
        "roofline": roofline,
    }
    if world > 1:
        line["rccl_ranks"] = ranks_seen if a.backend == "nccl" else None
        line["collective"] = {"backend": a.backend, "all_reduce_of_ones": ranks_seen}
    if world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(v, f, R, 7)
    line["extra"] = extra
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
