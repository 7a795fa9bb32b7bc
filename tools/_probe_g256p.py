import ctypes, os, sys, torch
sys.path.insert(0, "/root/repo")
from tvretrieval_amd import ops
lib = ops._lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
for abl, m, n, k in ((23, 300000, 2304, 768), (24, 300000, 2304, 768), (26, 300000, 2304, 768)):
    lib.xml_debug_set_q2c_ablation(ctypes.c_int(abl))
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.zeros(n, device="cuda")
    for _ in range(3): ops.linear(x, w, b)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    lib.xml_debug_read_g256p_probe(buf)
    for wv in (0, 4):
        l, e, f, t = buf[wv*4:wv*4+4]
        print("abl %d K=%d wave %d: tiles %d  loop %.0f  epilogue %.0f  first-slice %.0f  (memtime ticks per tile)" % (abl, k, wv, t, l/max(t,1), e/max(t,1), f/max(t,1)))
