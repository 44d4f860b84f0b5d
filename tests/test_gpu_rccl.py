"""The one-process-per-GPU path over the real backend: RCCL (`nccl`) with world size 1 on the one-GPU box — the
calls bench.py and sdf_amd.distributed make with N > 1 (init with device_id, device-tensor broadcasts of mesh and
R, all_gather, ShardedSDF.update).  The N = 2 logic runs under gloo in tests/test_distributed_cpu.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world_size_one_broadcast_update_gather():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_single_rank_check.py")], env=env,
                       capture_output=True, text=True, timeout=150)
    assert r.returncode == 0 and "nccl world-size-1 ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
