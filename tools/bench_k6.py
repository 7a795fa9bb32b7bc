#!/usr/bin/env python
"""Micro-benchmark of K6 (xml_q2c_scores) alone on random L2-normalised bf16 operands.  GPU box only.
usage: python tools/bench_k6.py [nq nv hidden] [--variants 2,1]"""
import ctypes
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvretrieval_amd import ops  # noqa: E402


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    nq, nv, h = (int(pos[0]), int(pos[1]), int(pos[2])) if len(pos) >= 3 else (10000, 21793, 768)
    variants = [2, 1]
    for a in sys.argv[1:]:
        if a.startswith("--variants="):
            variants = [int(x) for x in a.split("=")[1].split(",")]
    dtype = torch.float32 if "--f32" in sys.argv else torch.bfloat16
    abl = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--ablation=")]
    lib = ops._lib.load()
    assert hasattr(lib, "xml_debug_set_q2c_variant"), "needs the debug library: XML_DEBUG=1 bash tvretrieval_amd/csrc/build.sh; XMLHIP_LIB=$PWD/tvretrieval_amd/csrc/libxmlhip_dbg.so"
    if abl:
        lib.xml_debug_set_q2c_ablation(ctypes.c_int(abl[0]))
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.nn.functional.normalize(torch.randn(nq, h, device="cuda", generator=g), dim=-1).to(dtype)
    c = torch.empty(nv, 128, h, device="cuda", dtype=dtype)
    for b in range(0, nv, 2048):
        e = min(nv, b + 2048)
        c[b:e] = torch.nn.functional.normalize(torch.randn(e - b, 128, h, device="cuda", generator=g), dim=-1).to(dtype)
    mask = torch.ones(nv, 128, device="cuda")
    if "--ragged" in sys.argv:      # video lengths 32..128 clips: the masked paths (float patches / packed bits)
        lens = torch.randint(32, 129, (nv,), device="cuda", generator=g)
        mask = (torch.arange(128, device="cuda")[None] < lens[:, None]).float().contiguous()
    out = torch.empty(nq, nv, device="cuda")
    flops = 2.0 * nq * nv * 128 * h
    ref = None
    if "--tiled1" in sys.argv:      # one modality on slice-major tiles only (PMC calibration runs)
        t1 = ops.pack_q2c_corpus(c, mask)
        for _ in range(7):
            ops.q2c_scores_fused([q], [t1], [mask], out=out)
        torch.cuda.synchronize()
        return
    for v in variants:
        lib.xml_debug_set_q2c_variant(ctypes.c_int(v))
        for _ in range(2):
            ops.q2c_scores(q, c, mask, out=out)
        torch.cuda.synchronize()
        n = 5
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for s, e in evs:
            s.record(); ops.q2c_scores(q, c, mask, out=out); e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in evs)
        med = ms[n // 2]
        print("variant %d: median %.3f ms  min %.3f ms  -> %.1f TFLOP/s (median), %.1f (best)" %
              (v, med, ms[0], flops / med / 1e9, flops / ms[0] / 1e9), flush=True)
        if ref is None:
            ref = out[:256, :512].clone()
            want = torch.einsum("md,nld->mnl", q[:256].float(), c[:512].float()).max(-1)[0]
            print("  max |err| vs torch fp32 on a 256x512 corner: %.3e" % float((ref - want).abs().max()))
        else:
            print("  bitwise equal to first variant:", bool(torch.equal(ref, out[:256, :512])))
    lib.xml_debug_set_q2c_variant(ctypes.c_int(0))
    if "--fused" in sys.argv:   # both modalities in one launch (persistent kernel), reusing the same operands twice
        q2 = q.clone(); c2 = c.clone()
        for _ in range(2):
            ops.q2c_scores_fused([q, q2], [c, c2], [mask, mask], out=out)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for s, e in evs:
            s.record(); ops.q2c_scores_fused([q, q2], [c, c2], [mask, mask], out=out); e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in evs)
        print("fused x2 modalities: median %.3f ms -> %.1f TFLOP/s" % (ms[2], 2 * flops / ms[2] / 1e9), flush=True)
        print("  equals single-modality result (a+a)/2:", bool(torch.equal(ref, out[:256, :512])))
        keep = out.clone()
        t1, t2 = ops.pack_q2c_corpus(c, mask), ops.pack_q2c_corpus(c2, mask)   # slice-major tiles (the index layout);
        print("  tiles all_valid (masks skipped, 5-slot ring):", t1.all_valid, " packed mask bits:", t1.mask_bits is not None)
        for _ in range(2):
            ops.q2c_scores_fused([q, q2], [t1, t2], [mask, mask], out=out)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for s, e in evs:
            s.record(); ops.q2c_scores_fused([q, q2], [t1, t2], [mask, mask], out=out); e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in evs)
        print("fused x2, slice-major tiles (incl. tiling the queries): median %.3f ms -> %.1f TFLOP/s" %
              (ms[2], 2 * flops / ms[2] / 1e9), flush=True)
        print("  bitwise equal to the row-major result:", bool(torch.equal(keep, out)))


if __name__ == "__main__":
    main()
