"""GPU-side checks of the multi-GPU code path on a one-GPU box: (1) a real single-rank RCCL process group
(tests/rccl_world1_check.py under torch.distributed.run); (2) TWO ranks sharing the GPU with gloo between them -- real
kernels, real shards, sharded search == single-process search bit for bit (tests/two_ranks_one_gpu_check.py), and
`bench.py --gpus 2` through its own launcher in the same configuration.  RCCL across several GPUs is measured by the
driver's scaling run; its call pattern is what (1) exercises."""
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


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_sharded_search_equals_single_process(world):
    r = subprocess.run([sys.executable, "-c",
                        "import sys; sys.path.insert(0, %r); from tvretrieval_amd import launch; "
                        "sys.exit(launch.spawn_local_ranks(%r, [], %d, timeout=500))"
                        % (os.path.dirname(HERE), os.path.join(HERE, "two_ranks_one_gpu_check.py"), world)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "TWO_RANKS_ONE_GPU_OK world=%d" % world in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("extra", [[], ["--exact-rank"]])
def test_bench_two_ranks_on_one_gpu(extra):
    """`python bench.py --gpus 2` (self-spawned ranks, real kernels, shards of the tiny workload): one JSON line, n_gpus 2;
    also in exact-rank mode (every shard hands its exact local top-k to the merge).  Two ranks on ONE GPU cannot form an
    RCCL communicator, so the run asks for the torch.distributed exchange on purpose (--torch-collectives; without it the
    failed C-ABI self-check ends the run: test_bench_launcher.test_bench_refuses_to_measure_fallback_collectives)."""
    import json
    env = dict(os.environ, XML_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--workload", "tiny",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--torch-collectives"] + extra, env=env,
                       capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert len(out["config"]["videos_per_gpu"]) == 2 and sum(out["config"]["videos_per_gpu"]) > 0
    assert out["config"]["ranks_in_process_group"] == 2 and out["config"]["collectives_fallback"] == 1
