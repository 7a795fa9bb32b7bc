"""Corpus encode with context batches issued round-robin on S streams (independent batches: the HBM-bound kernels of one --
attention cores, LayerNorms -- can run under the MFMA-bound projections of another): videos/s for S = 1, 2, 3.
    python tools/bench_encode_streams.py [n_videos] [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tvretrieval_amd.model_xml import XML
nq, nv, l, hidden, dv, ds, dq, ctx_mode, _ = bench.WORKLOADS["c3"]
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
bench.CHUNK = bsz
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = XML(bench.model_config(hidden, dv, ds, dq, ctx_mode, l), compute_dtype=torch.bfloat16).to(dev).eval()
raw = list(bench.context_batches(0, nv, l, dv, ds, True, True, dev, None))
main = torch.cuda.current_stream(dev)


def run(n_streams):
    streams = [main] if n_streams == 1 else [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    outs = []
    start = torch.cuda.Event(enable_timing=True); end = torch.cuda.Event(enable_timing=True)
    start.record(main)
    for s in streams:
        if s is not main:
            s.wait_stream(main)
    with torch.no_grad():
        for i, b in enumerate(raw):
            with torch.cuda.stream(streams[i % len(streams)]):
                outs.append(model.encode_context(*b))
    for s in streams:
        if s is not main:
            main.wait_stream(s)
    end.record(main)
    torch.cuda.synchronize()
    return start.elapsed_time(end), outs


for n_streams in (1, 2, 3, 1, 2):
    for _ in range(2):
        run(n_streams)
    ts = sorted(run(n_streams)[0] for _ in range(5))
    print("%d stream(s), batches of %d: median %.2f ms -> %.0f videos/s" % (n_streams, bsz, ts[2], nv / ts[2] * 1e3))
a = run(1)[1]; b = run(2)[1]
print("same bits:", all(torch.equal(x, y) for p, q in zip(a, b) for x, y in zip(p, q)))
