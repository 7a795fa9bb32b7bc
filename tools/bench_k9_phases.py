"""K9 (xml_moment_topk_ex) at a bench workload's real pipeline inputs, with the debug library's early exits: 61 = after the
row-maxima histogram + bound, 62 = after the expansion into the candidate list, 0 = all (+ bitonic sort and output).
K9_WORKLOAD=tvr_val|c3|c3r (default tvr_val).  XMLHIP_LIB must point at libxmlhip_dbg.so."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tvretrieval_amd import inference as inf, ops
from tvretrieval_amd.model_xml import XML
WL = os.environ.get("K9_WORKLOAD", "tvr_val")
nq, nv, l, hidden, dv, ds, dq, ctx_mode, _ = bench.WORKLOADS[WL]
nq = int(os.environ.get("K9_NQ", nq))      # K9_NQ=50: one workgroup per CU at most -- the phases' latencies, not their throughput
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = XML(bench.model_config(hidden, dv, ds, dq, ctx_mode, l), compute_dtype=torch.bfloat16).to(dev).eval()
with torch.no_grad():
    lens = bench.real_clip_counts(nv, l) if WL in bench.RAGGED else None
    index = inf.build_corpus_index(m, bench.context_batches(0, nv, l, dv, ds, True, True, dev, lens), n_total=nv, l_ref=l)
    qf, qm = bench.synth_queries(nq, dq, dev)
    qvec = inf.stage_query_vectors(m, qf, qm)
    tw, ti = ops.topk_rows(inf.stage_q2c(index, qvec), 100, alpha=20.0)
    vl = inf.ragged_lengths(index)
    st, ed = inf.stage_span_probs(m, index, qvec, ti, **(dict(vid_len=vl) if vl is not None else {}))
    rk = dict(pair_vid=ti, vid_len=vl) if vl is not None else {}
    lib = ops._lib.load()
    assert hasattr(lib, "xml_debug_set_q2c_ablation"), "needs XMLHIP_LIB=.../libxmlhip_dbg.so"
    print("weights: w[0]/w[99] median %.2f; st row max median %.4f" % (float((tw[:, 0] / tw[:, -1]).median()), float(st.amax(-1).median())))
    for abl in (61, 62, 0):
        lib.xml_debug_set_q2c_ablation(ctypes.c_int(abl))
        for _ in range(2):
            ops.moment_topk(st, ed, tw, index.l_ref, 2, 16, 200, **rk)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for s, e in evs:
            s.record(); ops.moment_topk(st, ed, tw, index.l_ref, 2, 16, 200, **rk); e.record()
        torch.cuda.synchronize()
        print("exit %d: median %.3f ms" % (abl, sorted(s.elapsed_time(e) for s, e in evs)[2]))
    if hasattr(lib, "xml_debug_read_k9_stats"):      # how hard the expansion works: attempts, live rows, list entries per query
        buf = (ctypes.c_ulonglong * 8)()
        lib.xml_debug_read_k9_stats(buf, 1)
        lib.xml_debug_set_q2c_ablation(ctypes.c_int(63))
        ops.moment_topk(st, ed, tw, index.l_ref, 2, 16, 200, **rk)
        torch.cuda.synchronize()
        lib.xml_debug_read_k9_stats(buf, 1)
        n = max(buf[4], 1)
        print("per query: %.2f expansion attempts, %.0f live rows in the first, %.3f of the queries above 1024 rows, %.0f list entries"
              % (buf[0] / n, buf[1] / n, buf[2] / n, buf[3] / n))
    lib.xml_debug_set_q2c_ablation(ctypes.c_int(0))
