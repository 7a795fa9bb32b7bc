"""TEST INFRASTRUCTURE: bench.py's launcher and sharded orchestration on gloo ranks without a GPU.  This script is what
tests/test_bench_launcher.py starts in place of `python bench.py`: the same bench.main(), with the kernels replaced by
tests/cpu_backend.py (oracle formulation).  The measurement script itself has no such switch."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench  # noqa: E402
from cpu_backend import BenchBackend  # noqa: E402


def factory(local_rank):
    if os.environ.get("XML_TEST_FAIL_RANK") == os.environ.get("RANK", "0"):
        raise RuntimeError("injected rank failure (tests/test_bench_launcher.py)")
    return BenchBackend(local_rank)


if __name__ == "__main__":
    bench.main(backend_factory=factory, script=os.path.abspath(__file__))
