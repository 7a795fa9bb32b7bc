"""LayerNorm backward launches of the training step alone: the raw-feature input LayerNorm (12 800 x 3072 f32, parameter
gradients only, output-dropout site) and the hidden-size ones (12 800 x 768).  GPU box only."""
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
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2] * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
for rows, d, adt in [(12800, 3072, torch.float32), (12800, 768, torch.float32), (12800, 768, torch.bfloat16), (3840, 768, torch.bfloat16)]:
    a = torch.randn(rows, d, device="cuda", generator=g).to(adt)
    dy = torch.randn(rows, d, device="cuda", generator=g).to(torch.bfloat16)
    gam = torch.ones(d, device="cuda")
    dg, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
    wide = d > 1024
    if wide:
        t0 = timed(lambda: T.layernorm_bwd(a, None, gam, dy, need_dx=False, dg=dg, dbeta=db))
        t1 = timed(lambda: T.layernorm_bwd_drop(a, None, gam, dy, 0.0, 0, 0.1, 1234, need_dx=False, dg=dg, dbeta=db))
        print("rows %6d d %5d %s : params only %.1f us, with the output-dropout site %.1f us" % (rows, d, str(adt)[6:], t0, t1))
    else:
        b = torch.randn(rows, d, device="cuda", generator=g).to(torch.bfloat16)
        t0 = timed(lambda: T.layernorm_bwd(a, b, gam, dy, dg=dg, dbeta=db))
        t1 = timed(lambda: T.layernorm_bwd_drop(a, b, gam, dy, 0.1, 77, 0.1, 1234, dg=dg, dbeta=db))
        print("rows %6d d %5d %s : %.1f us, with both dropout sites %.1f us" % (rows, d, str(adt)[6:], t0, t1))
