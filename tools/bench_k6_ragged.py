#!/usr/bin/env python
"""K6 alone on a RAGGED corpus with the real TVR clip counts (tests/golden/tvr_clip_count_hist.json): packed image vs the
plain tiled image with bit masks, both modalities, random unit rows.  GPU box only.
usage: python tools/bench_k6_ragged.py [nq nv hidden [max_l]] [--f32]
       XML_PKG_ROOT=<tree> selects the package tree (A/B against a snapshot of an older round, e.g. _ab_r05)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("XML_PKG_ROOT", ROOT))
from tvretrieval_amd import ops  # noqa: E402


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    nq, nv, h = (int(pos[0]), int(pos[1]), int(pos[2])) if len(pos) >= 3 else (10000, 21793, 768)
    max_l = int(pos[3]) if len(pos) >= 4 else 128
    dtype = torch.float32 if "--f32" in sys.argv else torch.bfloat16
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "tvr_clip_count_hist.json")))
    pool = np.random.default_rng(2018).permutation(np.repeat(np.arange(len(rec["hist"])), rec["hist"]))
    lens = torch.from_numpy(np.maximum(np.minimum(pool[np.arange(nv) % len(pool)], max_l), 1)).cuda()
    if "--full" in sys.argv:        # every video 128 clips: the mask-free kernel of the headline
        lens = torch.full_like(lens, 128)
    mask = (torch.arange(128, device="cuda")[None] < lens[:, None]).float().contiguous()
    g = torch.Generator(device="cuda").manual_seed(0)
    qs = [torch.nn.functional.normalize(torch.randn(nq, h, device="cuda", generator=g), dim=-1).to(dtype) for _ in range(2)]
    cs = []
    for _ in range(2):
        c = torch.empty(nv, 128, h, device="cuda", dtype=dtype)
        for b in range(0, nv, 2048):
            e = min(nv, b + 2048)
            c[b:e] = (torch.nn.functional.normalize(torch.randn(e - b, 128, h, device="cuda", generator=g), dim=-1)
                      * mask[b:e, :, None]).to(dtype)
        cs.append(c)
    valid = float(lens.sum())
    out = torch.empty(nq, nv, device="cuda")
    res = {}
    for name in ("plain", "packed"):
        plan = ops.q2c_pack_plan([mask, mask]) if name == "packed" else None
        if name == "packed" and plan is None:
            print("packed: the plan declined this corpus" if "--full" not in sys.argv else "(full-length corpus: no packed image)")
            continue
        t = [ops.pack_q2c_corpus(c, mask, plan) for c in cs]
        rows = plan.n_tiles * 256 if plan is not None else nv * 128
        for _ in range(2):
            ops.q2c_scores_fused(qs, t, [mask, mask], out=out)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
        for s, e in evs:
            s.record(); ops.q2c_scores_fused(qs, t, [mask, mask], out=out); e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in evs)[3]
        alg = 2.0 * 2 * nq * valid * h / (ms * 1e-3) / 1e12
        exe = 2.0 * 2 * nq * rows * h / (ms * 1e-3) / 1e12
        peak = 157.3 if dtype == torch.float32 else 2500.0
        print("%-6s median %.3f ms  executed rows / valid %.4f  executed %.0f TF (%.3f)  algorithmic %.0f TF (%.3f of peak)"
              % (name, ms, rows / valid, exe, exe / peak, alg, alg / peak), flush=True)
        res[name] = out.clone()
        del t
    if len(res) == 2:
        print("packed == plain bitwise:", bool(torch.equal(res["plain"], res["packed"])))


if __name__ == "__main__":
    main()
