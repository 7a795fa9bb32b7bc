"""Same-box A/B of K6's tile walk at BASELINE configs[2] (10 000 x 21 793 x 128, H = 768, bf16, both modalities, the index's
tiled operands): the XCD super-tile shape (qsh: 2^qsh query tiles x 2^(5 - qsh) clip tiles), the adjacent-tile line length
(lsh) and the Infinity-Cache chunk walk (rsh).  One process = one box; every configuration's median of 7 launches, the
default re-measured at the end (drift).  With K6_ONLY=<name> it runs that one configuration 3 times and exits -- for a
`rocprofv3 --pmc FETCH_SIZE` pass per configuration (tools/k6_l2_ab.sh).  Needs XMLHIP_LIB=.../libxmlhip_dbg.so."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tvretrieval_amd import ops

CONFIGS = [("default(qsh=3)", dict()), ("qsh=2", dict(qsh=2)), ("qsh=4", dict(qsh=4)), ("qsh=1", dict(qsh=1)),
           ("lsh=2", dict(line=2)), ("chunk=2", dict(chunk=2)), ("qsh=2,chunk=2", dict(qsh=2, chunk=2)),
           ("default again", dict())]


def main():
    nq, nv, h = 10000, 21793, 768
    lib = ops._lib.load()
    assert hasattr(lib, "xml_debug_set_q2c_qsh"), "needs XMLHIP_LIB=.../libxmlhip_dbg.so"
    g = torch.Generator(device="cuda").manual_seed(0)
    nrm = lambda x: torch.nn.functional.normalize(x, dim=-1).to(torch.bfloat16)       # noqa: E731
    qs = [nrm(torch.randn(nq, h, device="cuda", generator=g)) for _ in range(2)]
    cs = []
    for _ in range(2):
        c = torch.empty(nv, 128, h, device="cuda", dtype=torch.bfloat16)
        for b in range(0, nv, 2048):
            e = min(nv, b + 2048)
            c[b:e] = nrm(torch.randn(e - b, 128, h, device="cuda", generator=g))
        cs.append(c)
    mask = torch.ones(nv, 128, device="cuda")
    tiles = [ops.pack_q2c_corpus(c, mask) for c in cs]
    del cs
    out = torch.empty(nq, nv, device="cuda")
    flops = 2.0 * nq * nv * 128 * h * 2
    only = os.environ.get("K6_ONLY")
    res, ref = [], None
    for name, kw in CONFIGS:
        if only and name != only:
            continue
        lib.xml_debug_set_q2c_qsh(ctypes.c_int(kw.get("qsh", -1)))
        lib.xml_debug_set_q2c_line(ctypes.c_int(kw.get("line", 0)))
        lib.xml_debug_set_q2c_chunk(ctypes.c_int(kw.get("chunk", -1)))
        n = 3 if only else 7
        for _ in range(1 if only else 2):
            ops.q2c_scores_fused(qs, tiles, [mask, mask], out=out)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for s, e in evs:
            s.record(); ops.q2c_scores_fused(qs, tiles, [mask, mask], out=out); e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in evs)
        if ref is None:
            ref = out[:512, :1024].clone()
        same = bool(torch.equal(ref, out[:512, :1024]))
        res.append(dict(config=name, median_ms=round(ms[n // 2], 3), min_ms=round(ms[0], 3), tflops=round(flops / ms[n // 2] / 1e9, 1),
                        frac=round(flops / ms[n // 2] / 1e9 / 2500.0, 4), same_scores=same))
        print(json.dumps(res[-1]), flush=True)
    lib.xml_debug_set_q2c_qsh(ctypes.c_int(-1)); lib.xml_debug_set_q2c_line(ctypes.c_int(0)); lib.xml_debug_set_q2c_chunk(ctypes.c_int(-1))


if __name__ == "__main__":
    main()
