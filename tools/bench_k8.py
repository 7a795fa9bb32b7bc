"""Micro-benchmark of K8 (xml_topk_rows) on score rows shaped like K6's output (max over 128 clips of cosines of
768-d unit vectors: positive, a handful of distinct exponents).  GPU box only.
usage: python tools/bench_k8.py [rows n k]..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvretrieval_amd import ops  # noqa: E402


def main():
    a = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(10000, 21793, 100), (10000, 2752, 100),
                                                                  (1250, 800, 100), (1250, 1600, 200)]
    g = torch.Generator(device="cuda").manual_seed(0)
    for rows, n, k in shapes:
        x = torch.empty(rows, n, device="cuda")
        for b in range(0, rows, 1000):
            e = min(rows, b + 1000)
            x[b:e] = (torch.randn(e - b, n, 16, device="cuda", generator=g) * 0.036).max(-1)[0]
        for _ in range(2):
            v, i = ops.topk_rows(x, k, alpha=20.0)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
        for s, e in evs:
            s.record(); v, i = ops.topk_rows(x, k, alpha=20.0); e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in evs)[3]
        tv, ti = torch.topk(x[:64], k, dim=1)
        ok = bool(torch.equal(ti.int(), i[:64])) and bool(torch.allclose(torch.exp(20.0 * tv), v[:64], rtol=1e-6))
        print("rows %6d  n %6d  k %4d : %.3f ms  (%.2f TB/s of one pass)  matches torch.topk on 64 rows: %s" %
              (rows, n, k, ms, rows * n * 4 / ms / 1e9, ok), flush=True)


if __name__ == "__main__":
    main()
