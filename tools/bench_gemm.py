"""Projection GEMM (xml_linear -> gemm256 when eligible) at the encoder shapes.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvretrieval_amd import ops  # noqa: E402

shapes = [(300000, 2304, 768), (300000, 768, 768), (65536, 768, 3072), (65536, 2304, 768), (65536, 768, 768),
          (16384, 2304, 768), (3840, 2304, 768), (300000, 2304, 3072)]
if os.environ.get("GEMM_SHAPES") == "train":      # the projections of one training step (BASELINE configs[4]: 128 x 100 clips)
    shapes = [(12800, 768, 768), (12800, 2304, 768), (12800, 1536, 768), (12800, 768, 3072), (12800, 3072, 768),
              (3840, 768, 768), (3840, 2304, 768), (128, 768, 768), (750, 256, 768), (750, 768, 256), (750, 256, 256)]
import ctypes
lib = ops._lib.load()
assert hasattr(lib, "xml_debug_set_q2c_variant"), "needs the debug library: XML_DEBUG=1 bash tvretrieval_amd/csrc/build.sh; XMLHIP_LIB=$PWD/tvretrieval_amd/csrc/libxmlhip_dbg.so"
g = torch.Generator(device="cuda").manual_seed(0)
abl = int(os.environ.get("XML_ABL", "0"))
lib.xml_debug_set_q2c_ablation(ctypes.c_int(abl))
lib.xml_debug_set_gemm_variant(ctypes.c_int(int(os.environ.get("XML_GEMM_VARIANT", "0"))))
for m, n, k in shapes:
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.zeros(n, device="cuda")
    for _ in range(3):
        ops.linear(x, w, b)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for s, e in evs:
        s.record(); ops.linear(x, w, b); e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)[3]
    print("M %7d N %5d K %5d : %.3f ms  %.0f TFLOP/s" % (m, n, k, ms, 2.0 * m * n * k / ms / 1e9), flush=True)
