"""Micro-benchmark of K9 (xml_moment_topk) on real pipeline inputs dumped by tools/dump_k9_inputs.py (64 queries,
tiled to 10 000)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvretrieval_amd import ops
d = torch.load("tools/_k9_inputs.pt")
rep = 10000 // 64 + 1
st = d["st"].repeat(rep, 1, 1)[:10000].cuda().contiguous()
ed = d["ed"].repeat(rep, 1, 1)[:10000].cuda().contiguous()
w = d["w"].repeat(rep, 1)[:10000].cuda().contiguous()
for name, ww in (("all pairs", w), ("1/8 owned", torch.where((torch.arange(100, device="cuda")[None] % 8) == 3, w, torch.zeros_like(w)).contiguous())):
    for _ in range(2):
        ops.moment_topk(st, ed, ww, 128, 2, 16, 200)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for s, e in evs:
        s.record(); sc, fl = ops.moment_topk(st, ed, ww, 128, 2, 16, 200); e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    print("%s: median %.3f ms" % (name, ms[2]))
# fixed-cost probes
one = torch.zeros_like(w); one[:, 0] = w[:, 0]
for name, ww, n_out in (("1 pair owned", one, 200), ("1 pair owned, n_out=16", one, 16)):
    for _ in range(2):
        ops.moment_topk(st, ed, ww, 128, 2, 16, n_out)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for s, e in evs:
        s.record(); ops.moment_topk(st, ed, ww, 128, 2, 16, n_out); e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    print("%s: median %.3f ms" % (name, ms[2]))
