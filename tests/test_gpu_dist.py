"""GPU-side check of the multi-GPU code path on a one-GPU box: a real single-rank RCCL process group
(tests/rccl_world1_check.py under torch.distributed.run).  World sizes > 1 are covered on CPU by the gloo tests
(tests/test_dist_gloo.py) and on hardware by the driver's scaling run."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_rccl_single_rank_group_runs_the_sharded_path():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "rccl_world1_check.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
