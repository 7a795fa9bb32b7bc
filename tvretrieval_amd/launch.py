"""One process per GPU on one node, without torchrun: `spawn_local_ranks` starts N copies of a script with the
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT environment torch.distributed.run would give them.

The reference's measurement protocol is one process driving the whole measurement
(baselines/profiling/profile_main.py:457-461); `python bench.py --gpus N` keeps that command line and fans out here.
Rank 0 inherits stdout (the single JSON line); the other ranks' stdout goes to stderr so that library banners cannot
land behind the result line.  If any rank fails, the remaining ones (exact PIDs we started) are terminated and the
first non-zero exit code is returned.
"""
import os
import socket
import subprocess
import sys
import time

SPAWNED_FLAG = "XML_SELF_SPAWNED"


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def under_launcher():
    """True when this process already is one rank of a multi-process job (torch.distributed.run or spawn_local_ranks)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env[SPAWNED_FLAG] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    return env


def spawn_local_ranks(script, argv, nproc, timeout=None, poll_s=0.2):
    """Run `python script argv...` as `nproc` ranks on this node and wait for them.  Returns the job's exit code."""
    port = free_port()
    procs = []
    for r in range(nproc):
        out = None if r == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=rank_env(r, nproc, port), stdout=out))
    t0 = time.time()
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (timeout is not None and time.time() - t0 > timeout):
                if rc == 0:
                    rc = 124
                break
            time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc
