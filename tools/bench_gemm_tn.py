"""Weight-gradient GEMM dW = dY^T X: xml_gemm_tn (row-major operands, transpose reads) against the path it replaced
(two explicit transposes + the split-K NT kernel), training-step shapes.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvretrieval_amd import train_ops as T  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2]


import ctypes
lib = T._lib.load()
if hasattr(lib, "xml_debug_set_q2c_ablation"):
    lib.xml_debug_set_q2c_ablation(ctypes.c_int(int(os.environ.get("XML_ABL", "0"))))
g = torch.Generator(device="cuda").manual_seed(0)
for rows, n, k in [(12800, 768, 768), (12800, 2304, 768), (12800, 768, 3072), (3840, 768, 768), (3840, 2304, 768)]:
    dy = torch.randn(rows, n, device="cuda", generator=g).to(torch.bfloat16)
    x = torch.randn(rows, k, device="cuda", generator=g).to(torch.bfloat16)
    r8 = (rows + 7) // 8 * 8
    t_new = timed(lambda: T.gemm_tn(dy, x))
    t_old = timed(lambda: T.gemm_batched(T.transpose(dy, r8), T.transpose(x, r8), out_f32=True))
    gf = 2.0 * rows * n * k / 1e9
    print("rows %6d N %5d K %5d : gemm_tn %.1f us (%.0f TF)   transposes + split-K %.1f us (%.0f TF)"
          % (rows, n, k, t_new * 1e3, gf / t_new, t_old * 1e3, gf / t_old), flush=True)
