"""Fused training attention (attention_train.hip) alone at the training step's shape: 128 sequences x 100 clips, H = 768,
4 heads (dh = 192), dropout 0.1.  With a library built with XML_DEBUG_EXTRA=-DXML_AT_PROBE: the stage timers of one workgroup
of the backward kernel.  GPU box only."""
import ctypes
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


n, l, h, heads = (int(v) for v in (sys.argv[1:5] + ["128", "100", "768", "4"][len(sys.argv) - 1:]))
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(n, l, 3 * h, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
dout = (torch.randn(n, l, h, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
mask = torch.ones(n, l, device="cuda")
dqkv = torch.empty_like(qkv)
t_f = timed(lambda: T.attention_train_fwd(qkv, qkv, qkv, None, mask, heads, h, 0.1, 11, 0, h, 2 * h))
t_b = timed(lambda: T.attention_train_bwd(qkv, qkv, qkv, None, mask, dout, dqkv, dqkv, dqkv, heads, h, 0.1, 11, 0, h, 2 * h,
                                          0, h, 2 * h))
fl = 4.0 * n * heads * l * l * (h // heads)
print("n %d L %d H %d heads %d : forward %.1f us (%.0f TF), backward %.1f us (%.0f TF)"
      % (n, l, h, heads, t_f, fl / t_f / 1e6, t_b, 2.5 * fl / t_b / 1e6))
lib = T._lib.load()
if hasattr(lib, "xml_debug_read_at_probe"):
    buf = (ctypes.c_ulonglong * 64)()
    assert lib.xml_debug_read_at_probe(buf) == 0
    names = ["loads+stage K V", "S = Q K^T", "softmax", "dP = dO V^T", "dropout / dS", "dQ = dS K", "barrier", "Pd^T dS^T -> LDS",
             "stage dO", "dV", "stage Q", "dK"]
    for w in range(4):
        t = [buf[w * 16 + i] for i in range(13)]
        print("wave %d: " % w + "  ".join("%s %d" % (names[i], t[i + 1] - t[i]) for i in range(12)) + "   total %d" % (t[12] - t[0]))
