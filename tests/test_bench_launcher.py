"""`python bench.py --gpus N` must measure N ranks by itself (the driver's command line has no torchrun).  Here the very
same launcher code (tvretrieval_amd/launch.py + bench.py main, entered through tests/bench_cpu_entry.py) starts 2 gloo
ranks on CPU, with the kernels replaced by tests/cpu_backend.py, and must print exactly one JSON line that says n_gpus = 2."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run(argv, timeout=900, script="bench.py", **extra_env):
    env = dict(os.environ, PYTHONPATH=HERE + os.pathsep + ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    path = os.path.join(ROOT, script) if script == "bench.py" else os.path.join(HERE, script)
    return subprocess.run([sys.executable, path] + argv, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_self_spawns_two_gloo_ranks():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "tiny", "--no-cpu-baseline"],
             script="bench_cpu_entry.py")
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert lines[-1].startswith("{") and sum(ln.startswith("{") for ln in lines) == 1, lines   # ONE result line, last
    res = json.loads(lines[-1])
    assert res["n_gpus"] == 2 and res["config"]["ranks_in_process_group"] == 2
    assert res["config"]["launcher"] == "bench.py self-spawn"
    assert len(res["config"]["videos_per_gpu"]) == 2 and sum(res["config"]["videos_per_gpu"]) == 300
    assert res["value"] > 0 and res["scaling"] == "strong"
    stages = res["breakdown_ms"]
    assert "exchange+merge_topk" in stages and "q2c_k6" in stages, stages
    # both rerank schemes get a number: the headline's (query owner) and the north_star-literal fully sharded one
    alt = res["extras"]["other_rerank_scheme"]
    assert alt["value"] > 0 and alt["collectives_per_pass"] == 4 and res["config"]["rerank"].startswith("query owner")
    assert res["config"]["collectives_fallback"] == 0


def test_bench_self_spawns_eight_gloo_ranks():
    """The driver's 8-GPU command line, rehearsed on CPU: `bench.py --gpus 8` through its own launcher, 8 gloo ranks, the
    300-video test corpus cut into 8 shards, both rerank schemes timed, one JSON line."""
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0", "--workload", "tiny", "--no-cpu-baseline"],
             script="bench_cpu_entry.py", timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert lines[-1].startswith("{") and sum(ln.startswith("{") for ln in lines) == 1, lines
    res = json.loads(lines[-1])
    assert res["n_gpus"] == 8 and res["config"]["ranks_in_process_group"] == 8
    assert len(res["config"]["videos_per_gpu"]) == 8 and sum(res["config"]["videos_per_gpu"]) == 300
    assert min(res["config"]["videos_per_gpu"]) >= 37
    assert res["value"] > 0 and res["extras"]["other_rerank_scheme"]["value"] > 0


def test_bench_rank_failure_propagates():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "tiny"], timeout=300,
             script="bench_cpu_entry.py", XML_TEST_FAIL_RANK="1")
    assert r.returncode != 0


@pytest.mark.gpu
def test_bench_tiny_forced_sharded_on_gpu():
    """Same entry on the GPU box: the N > 1 code path through a real 1-rank RCCL group."""
    r = _run(["--gpus", "1", "--force-sharded", "--workload", "tiny", "--steps", "2", "--warmup", "1",
              "--no-cpu-baseline"])
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["config"]["backend"] == "hip" and res["config"]["ranks_in_process_group"] == 1
    assert "exchange+merge_topk" in res["breakdown_ms"] or "topk_k8" in res["breakdown_ms"]
    assert res["config"]["collectives"].startswith("libxmlhip RCCL")


@pytest.mark.gpu
def test_bench_refuses_to_measure_fallback_collectives():
    """A C-ABI collective that fails its self-check must END an N > 1 run with a non-zero exit code -- a line that silently
    measured torch.distributed would be read as a measurement of csrc/collectives.hip -- unless --torch-collectives asked
    for that exchange on purpose (the line then carries collectives_fallback = 1)."""
    argv = ["--gpus", "1", "--force-sharded", "--workload", "tiny", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    r = _run(argv, XML_TEST_BREAK_CABI_COLLECTIVES="1")
    assert r.returncode != 0, r.stdout[-2000:]
    assert "Refusing to measure" in r.stderr and not any(ln.startswith("{") for ln in r.stdout.splitlines())
    r = _run(argv + ["--torch-collectives"], XML_TEST_BREAK_CABI_COLLECTIVES="1")
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["config"]["collectives_fallback"] == 1 and res["config"]["collectives"].startswith("torch.distributed")
