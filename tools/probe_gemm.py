"""Phase probe of the persistent projection GEMM (debug library, XML_ABL=9): where a workgroup's time goes per tile."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvretrieval_amd import ops  # noqa: E402

lib = ops._lib.load()
assert hasattr(lib, "xml_debug_read_gemm_probe"), "needs XMLHIP_LIB=.../libxmlhip_dbg.so"
lib.xml_debug_set_q2c_ablation(ctypes.c_int(9))
g = torch.Generator(device="cuda").manual_seed(0)
for m, n, k in [(300000, 2304, 768), (300000, 768, 768), (262144, 768, 3072)]:
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.zeros(n, device="cuda")
    for _ in range(3):
        ops.linear(x, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.linear(x, w, b); e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    buf = (ctypes.c_ulonglong * 64)()
    lib.xml_debug_read_gemm_probe.argtypes = [ctypes.c_void_p]
    assert lib.xml_debug_read_gemm_probe(buf) == 0
    print("M %d N %d K %d: %.3f ms, workgroup 0 wave 0 ran %d s_memtime ticks in %d s_memrealtime ticks (100 MHz) = %.1f us -> "
          "%.0f s_memtime ticks per us" % (m, n, k, ms, buf[5], buf[7], buf[7] / 100.0, buf[5] / max(buf[7], 1) * 100.0))
    sb = (ctypes.c_uint * 512)()
    lib.xml_debug_read_gemm_steps.argtypes = [ctypes.c_void_p]
    assert lib.xml_debug_read_gemm_steps(sb) == 0
    nt = max(buf[6], 1)
    for w_ in (0, 4):
        print("  wave %d ticks per slice step (avg over tiles): %s" % (w_, " ".join("%d" % (sb[w_ * 64 + i] // nt) for i in range(min(k // 32, 64)))))
    for w_ in range(8):
        p = [buf[w_ * 8 + i] for i in range(8)]
        t = max(p[6], 1)
        print("  wave %d: tiles %3d  K loop %6.2f us  bias/drain %5.2f us  rows+stores %5.2f us  ln-sync %5.2f  ln-pass2 %5.2f   total/tile %6.2f us"
              % (w_, p[6], p[0] / t / 100, p[1] / t / 100, p[2] / t / 100, p[3] / t / 100, p[4] / t / 100, p[5] / t / 100))


def show(tag, ms, k):
    buf = (ctypes.c_ulonglong * 64)()
    assert lib.xml_debug_read_gemm_probe(buf) == 0
    print("%s: %.3f ms" % (tag, ms))
    for w_ in range(8):
        p = [buf[w_ * 8 + i] for i in range(8)]
        t = max(p[6], 1)
        print("  wave %d: tiles %3d  K loop %6.2f us  bias/drain %5.2f us  rows+stores %5.2f us  ln-sync %5.2f  ln-pass2 %5.2f   total/tile %6.2f us"
              % (w_, p[6], p[0] / t / 100, p[1] / t / 100, p[2] / t / 100, p[3] / t / 100, p[4] / t / 100, p[5] / t / 100))


# the LayerNorm-epilogue kernel through K1+K2 on packed query tokens (175 000 rows, N = K = 768)
rows, h = 175000, 768
x = torch.randn(1, rows, h, device="cuda", generator=g)
gam, bet = torch.ones(h, device="cuda"), torch.zeros(h, device="cuda")
w = (torch.randn(h, h, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
b = torch.zeros(h, device="cuda")
pos = (torch.randn(rows, h, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
for _ in range(3):
    ops.linear_ln_relu_pos(x, gam, bet, w, b, pos, gam, bet)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.linear_ln_relu_pos(x, gam, bet, w, b, pos, gam, bet); e1.record()
torch.cuda.synchronize()
show("K1+K2 (input LN launch + LN-epilogue GEMM), rows %d" % rows, e0.elapsed_time(e1), h)
