#!/usr/bin/env python
"""A/B of K6's corpus walk order (q2c_persist.hip, `rsh`): rounds per Infinity-Cache-resident chunk = 2^v.
Needs the debug library:  XML_DEBUG=1 bash tvretrieval_amd/csrc/build.sh
    XMLHIP_LIB=tvretrieval_amd/csrc/libxmlhip_dbg.so python tools/bench_k6_chunk.py [--values 20,0,1,2,3,4] [--nq 10000]
v = 20 is the straight order (every query group streams the whole corpus from HBM)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvretrieval_amd import ops  # noqa: E402


def main():
    vals = [20, 0, 1, 2, 3, 4, -1]
    abls = [0]
    lines = [0]                     # --lines=0,2: rounds an XCD spends on ADJACENT clip tiles = 2^v (Q2cPersistArgs::lsh)
    nq, nv, h = 10000, 21793, 768
    for a in sys.argv[1:]:
        if a.startswith("--values="):
            vals = [int(x) for x in a.split("=")[1].split(",")]
        if a.startswith("--ablations="):
            abls = [int(x) for x in a.split("=")[1].split(",")]
        if a.startswith("--lines="):
            lines = [int(x) for x in a.split("=")[1].split(",")]
        if a.startswith("--nq="):
            nq = int(a.split("=")[1])
        if a.startswith("--nv="):
            nv = int(a.split("=")[1])
    lib = ops._lib.load()
    assert hasattr(lib, "xml_debug_set_q2c_variant"), "needs the debug library: XML_DEBUG=1 bash tvretrieval_amd/csrc/build.sh; XMLHIP_LIB=$PWD/tvretrieval_amd/csrc/libxmlhip_dbg.so"
    g = torch.Generator(device="cuda").manual_seed(0)
    dt = torch.bfloat16
    qs, cs = [], []
    for m in range(2):
        qs.append(torch.nn.functional.normalize(torch.randn(nq, h, device="cuda", generator=g), dim=-1).to(dt))
        c = torch.empty(nv, 128, h, device="cuda", dtype=dt)
        for b in range(0, nv, 2048):
            e = min(nv, b + 2048)
            c[b:e] = torch.nn.functional.normalize(torch.randn(e - b, 128, h, device="cuda", generator=g), dim=-1).to(dt)
        cs.append(c)
    mask = torch.ones(nv, 128, device="cuda")
    tiles = [ops.pack_q2c_corpus(c, mask) for c in cs]
    del cs
    out = torch.empty(nq, nv, device="cuda")
    flops = 2.0 * nq * nv * 128 * h * 2
    ref = None
    for rep in range(2):
        for v, ab, ln in [(v, ab, ln) for v in vals for ab in abls for ln in lines]:
            lib.xml_debug_set_q2c_chunk(ctypes.c_int(v))
            lib.xml_debug_set_q2c_line(ctypes.c_int(ln))
            lib.xml_debug_set_q2c_ablation(ctypes.c_int(ab))
            for _ in range(2):
                ops.q2c_scores_fused(qs, tiles, [mask, mask], out=out)
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for s, e in evs:
                s.record(); ops.q2c_scores_fused(qs, tiles, [mask, mask], out=out); e.record()
            torch.cuda.synchronize()
            ms = sorted(s.elapsed_time(e) for s, e in evs)
            same = ""
            if ref is None:
                ref = out.clone()
            else:
                same = "  bitwise equal to the first setting: %s" % bool(torch.equal(ref, out))
            print("chunk 2^%-2d rounds, ablation %2d, line 2^%d: median %.3f ms (min %.3f) -> %.1f TFLOP/s%s" %
                  (v, ab, ln, ms[2], ms[0], flops / ms[2] / 1e9, same), flush=True)
    lib.xml_debug_set_q2c_chunk(ctypes.c_int(-1))
    lib.xml_debug_set_q2c_line(ctypes.c_int(0))
    lib.xml_debug_set_q2c_ablation(ctypes.c_int(0))


if __name__ == "__main__":
    main()
