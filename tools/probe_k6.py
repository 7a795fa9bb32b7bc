"""K6 timing probe (ablation 8): cycles each wave of workgroup 0 spends in the counted wait and at the barrier."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvretrieval_amd import ops
lib = ops._lib.load()
assert hasattr(lib, "xml_debug_set_q2c_variant"), "needs the debug library: XML_DEBUG=1 bash tvretrieval_amd/csrc/build.sh; XMLHIP_LIB=$PWD/tvretrieval_amd/csrc/libxmlhip_dbg.so"
nq, nv, h = 10000, 21793, 768
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.nn.functional.normalize(torch.randn(nq, h, device="cuda", generator=g), dim=-1).bfloat16()
c = torch.empty(nv, 128, h, device="cuda", dtype=torch.bfloat16)
for b in range(0, nv, 2048):
    e = min(nv, b + 2048)
    c[b:e] = torch.nn.functional.normalize(torch.randn(e - b, 128, h, device="cuda", generator=g), dim=-1).bfloat16()
mask = torch.ones(nv, 128, device="cuda")
out = torch.empty(nq, nv, device="cuda")
lib.xml_debug_set_q2c_variant(ctypes.c_int(4))
lib.xml_debug_set_q2c_ablation(ctypes.c_int(8))
for _ in range(3):
    ops.q2c_scores(q, c, mask, out=out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
lib.xml_debug_read_k6_probe.argtypes = [ctypes.c_void_p]
assert lib.xml_debug_read_k6_probe(buf) == 0
for w in range(8):
    wait, bar, tot, real = buf[w * 4], buf[w * 4 + 1], buf[w * 4 + 2], buf[w * 4 + 3]
    print("wave %d: total %.2f Mcyc in %.2f ms (s_memrealtime) = %.0f MHz shader clock  counted-wait %.1f %%  barrier %.1f %%"
          % (w, tot / 1e6, real / 1e5, tot / max(real, 1) * 100.0, 100.0 * wait / tot, 100.0 * bar / tot))
lib.xml_debug_set_q2c_ablation(ctypes.c_int(0))
