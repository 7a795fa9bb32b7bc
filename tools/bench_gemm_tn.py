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
SHAPES = [(12800, 768, 768), (12800, 1536, 768), (12800, 2304, 768), (12800, 768, 3072), (3840, 768, 768), (3840, 2304, 768)]
if os.environ.get("TN_SHAPE"):                 # "rows,N,K": one shape, accumulate mode (no fill), no comparison path -- profiling
    rows, n, k = (int(v) for v in os.environ["TN_SHAPE"].split(","))
    dy = torch.randn(rows, n, device="cuda", generator=g).to(torch.bfloat16)
    x = torch.randn(rows, k, device="cuda", generator=g).to(torch.bfloat16)
    acc = torch.zeros(n, k, device="cuda")
    t = timed(lambda: T.gemm_tn(dy, x, out=acc))
    print("rows %6d N %5d K %5d : gemm_tn (accumulate) %.1f us (%.0f TF)" % (rows, n, k, t * 1e3, 2.0 * rows * n * k / 1e9 / t))
    SHAPES = []
    if os.environ.get("XML_ABL") == "308":       # per-stage cycles of workgroup 0's waves, last launch
        buf = (ctypes.c_ulonglong * 64)()
        assert lib.xml_debug_read_tn_probe(buf) == 0
        print("wave: wait barrier pre-MFMA MFMA post-MFMA (cycles per step)")
        for w in range(8):
            n = max(buf[w * 8 + 5], 1)
            print("  %d: " % w + "  ".join("%6.0f" % (buf[w * 8 + i] / n) for i in range(5)))
for rows, n, k in SHAPES:
    dy = torch.randn(rows, n, device="cuda", generator=g).to(torch.bfloat16)
    x = torch.randn(rows, k, device="cuda", generator=g).to(torch.bfloat16)
    r8 = (rows + 7) // 8 * 8
    t_new = timed(lambda: T.gemm_tn(dy, x))
    acc = torch.zeros(n, k, device="cuda")
    t_acc = timed(lambda: T.gemm_tn(dy, x, out=acc))      # the training step's form: accumulate into .grad, no fill
    t_old = timed(lambda: T.gemm_batched(T.transpose(dy, r8), T.transpose(x, r8), out_f32=True))
    gf = 2.0 * rows * n * k / 1e9
    print("rows %6d N %5d K %5d : gemm_tn %.1f us (%.0f TF), accumulate %.1f us   transposes + split-K %.1f us (%.0f TF)"
          % (rows, n, k, t_new * 1e3, gf / t_new, t_acc * 1e3, t_old * 1e3, gf / t_old), flush=True)
