#!/usr/bin/env python
"""BertAttention block (QKV GEMM + attention core + output GEMM + LN) at the corpus-encode shape, with the attention core's
timing ablations of the debug library (XMLHIP_LIB=.../libxmlhip_dbg.so).  Differences between settings = the ablated part."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvretrieval_amd import ops  # noqa: E402


def main():
    lib = ops._lib.load()
    assert hasattr(lib, "xml_debug_set_q2c_variant"), "needs the debug library: XML_DEBUG=1 bash tvretrieval_amd/csrc/build.sh; XMLHIP_LIB=$PWD/tvretrieval_amd/csrc/libxmlhip_dbg.so"
    n, l, h = 512, 128, 768
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(n, l, h, device="cuda", generator=g).to(torch.bfloat16)
    mask = torch.ones(n, l, device="cuda")
    wqkv = (torch.randn(3 * h, h, device="cuda", generator=g) * h ** -0.5).to(torch.bfloat16)
    wo = (torch.randn(h, h, device="cuda", generator=g) * h ** -0.5).to(torch.bfloat16)
    z3, z1, o1 = torch.zeros(3 * h, device="cuda"), torch.zeros(h, device="cuda"), torch.ones(h, device="cuda")
    for abl, what in ((0, "full block"), (2, "no phase B (P V)"), (6, "no global loads (zeros)"), (7, "loads + K staging only"),
                      (0, "full block")):
        lib.xml_debug_set_q2c_ablation(ctypes.c_int(abl))
        for _ in range(3):
            ops.attention_block(x, mask, wqkv, z3, wo, z1, o1, z1, 4)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
        for s, e in evs:
            s.record(); ops.attention_block(x, mask, wqkv, z3, wo, z1, o1, z1, 4); e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in evs)
        print("ablation %d (%s): block median %.1f us" % (abl, what, ms[3] * 1e3), flush=True)
    lib.xml_debug_set_q2c_ablation(ctypes.c_int(8))
    for _ in range(3):
        ops.attention_block(x, mask, wqkv, z3, wo, z1, o1, z1, 4)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    lib.xml_debug_read_attn_probe.argtypes = [ctypes.c_void_p]
    assert lib.xml_debug_read_attn_probe(buf) == 0
    t = list(buf)[:7]
    names = ["loads issued + K stored", "barrier 1 (K visible)", "phase A (QK^T, softmax) x2 tiles", "barrier 2", "V staged + barrier 3",
             "phase B (P V, stores) x2 tiles"]
    print("probe (workgroup head 1 / sequence 300, wave 1), s_memtime ticks (100 MHz => x10 ns... see delta ratios):")
    for i, nm in enumerate(names):
        print("  %-36s %8d" % (nm, t[i + 1] - t[i]))
    lib.xml_debug_set_q2c_ablation(ctypes.c_int(0))


if __name__ == "__main__":
    main()
