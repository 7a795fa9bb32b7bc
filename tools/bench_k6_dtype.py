#!/usr/bin/env python
"""K6 (tiled persistent kernel, both modalities) on the same unit-norm operands in bf16, in f16, and in f16 with the low
mantissa bits of both operands cleared: how much of the f16 filter's extra time is operand entropy (the kernel is
power-limited) and what a k-bit-shorter mantissa buys back.  GPU box only.
usage: python tools/bench_k6_dtype.py [nq nv hidden]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvretrieval_amd import ops  # noqa: E402


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    nq, nv, h = (int(pos[0]), int(pos[1]), int(pos[2])) if len(pos) >= 3 else (10000, 21793, 768)
    g = torch.Generator(device="cuda").manual_seed(0)
    q32 = [torch.nn.functional.normalize(torch.randn(nq, h, device="cuda", generator=g), dim=-1) for _ in range(2)]
    c32 = []
    for _ in range(2):
        c = torch.empty(nv, 128, h, device="cuda")
        for b in range(0, nv, 2048):
            e = min(nv, b + 2048)
            c[b:e] = torch.nn.functional.normalize(torch.randn(e - b, 128, h, device="cuda", generator=g), dim=-1)
        c32.append(c)
    mask = torch.ones(nv, 128, device="cuda")
    out = torch.empty(nq, nv, device="cuda")
    flops = 2.0 * 2 * nq * nv * 128 * h

    def run(name, qs, cs):
        tiles = [ops.pack_q2c_corpus(c, mask) for c in cs]
        for _ in range(2):
            ops.q2c_scores_fused(qs, tiles, [mask, mask], out=out)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for s, e in evs:
            s.record(); ops.q2c_scores_fused(qs, tiles, [mask, mask], out=out); e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in evs)
        print("%-34s median %.2f ms  min %.2f ms -> %.0f TFLOP/s" % (name, ms[2], ms[0], flops / ms[2] / 1e9), flush=True)
        del tiles

    run("bf16", [q.to(torch.bfloat16) for q in q32], [c.to(torch.bfloat16) for c in c32])

    def f16(x, drop):
        hi = ops.split_f16_rows(x.contiguous(), ops.F16_UNIT_LOG2, want_hi=True)[1]
        if drop:
            hi = (hi.view(torch.int16) & ~((1 << drop) - 1)).view(torch.float16)      # truncate `drop` mantissa bits
        return hi.contiguous()
    for drop in (0, 1, 2, 3, 5):
        run("f16, %d low mantissa bits cleared" % drop, [f16(q, drop) for q in q32], [f16(c, drop) for c in c32])
    if "--one-side" in sys.argv:
        run("f16 corpus full, queries -3 bits", [f16(q, 3) for q in q32], [f16(c, 0) for c in c32])
        run("f16 queries full, corpus -3 bits", [f16(q, 0) for q in q32], [f16(c, 3) for c in c32])


if __name__ == "__main__":
    main()
