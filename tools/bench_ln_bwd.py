"""LayerNorm backward micro-benchmark at the training step's shapes (12 800 x 768 bf16 with a residual; 3 840 x 768 query rows).
   gpurun -- python tools/bench_ln_bwd.py"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tvretrieval_amd import train_ops as T  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    for rows, d, a_dt in ((12800, 768, torch.bfloat16), (12800, 768, torch.float32), (3840, 768, torch.bfloat16)):
        a = torch.randn(rows, d, device=dev).to(a_dt)
        b = torch.randn(rows, d, device=dev).to(torch.bfloat16)
        dy = torch.randn(rows, d, device=dev).to(torch.bfloat16)
        g = torch.randn(d, device=dev)
        dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
        us = timeit(lambda: T.layernorm_bwd(a, b, g, dy, dg=dg, dbeta=db))
        mb = (a.numel() * a.element_size() + 3 * b.numel() * 2) / 1e6
        print("layernorm_bwd %6d x %4d a=%s: %7.1f us  (%.0f MB -> %.2f TB/s)" % (rows, d, str(a_dt)[6:], us, mb, mb / us / 1e6 * 1e6 / 1e6))
        us = timeit(lambda: T.layernorm_bwd_drop(a, b, g, dy, 0.1, 11, 0.0, 0, dg=dg, dbeta=db))
        print("   with an input dropout site (extra dxa):   %7.1f us" % us)
        y = timeit(lambda: T.add_layernorm_drop(a, b, g, g, torch.bfloat16, 0.1, 11, 0.0, 0))
        print("   forward (input dropout site):             %7.1f us" % y)


if __name__ == "__main__":
    main()
