"""K7 (xml_convse_rerank) alone at the C3 shape with ablations: 0 full, 1 no GEMMs, 2 no epilogue, 40 register-staged
mainloop instead of the LDS-DMA ring.  K7_DTYPE=f32 for the exact-rank mode's f32 operands (default bf16)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tvretrieval_amd import inference as inf, ops
from tvretrieval_amd.model_xml import XML
WL = os.environ.get("K7_WORKLOAD", "c3")          # e.g. tvr_val: the as-trained shape (H = 256, real clip counts)
nq, nv, l, hidden, dv, ds, dq, ctx_mode, _ = bench.WORKLOADS[WL]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = XML(bench.model_config(hidden, dv, ds, dq, ctx_mode, l), compute_dtype=torch.float32 if os.environ.get("K7_DTYPE", "bf16") == "f32" else torch.bfloat16).to(dev).eval()
with torch.no_grad():
    lens = bench.real_clip_counts(nv, l) if WL in bench.RAGGED else None
    index = inf.build_corpus_index(m, bench.context_batches(0, nv, l, dv, ds, True, True, dev, lens), n_total=nv, l_ref=l)
    qf, qm = bench.synth_queries(nq, dq, dev)
    qvec = inf.stage_query_vectors(m, qf, qm)
    q2c = inf.stage_q2c(index, qvec)
    tw, ti = ops.topk_rows(q2c, 100, alpha=20.0)
    lib = ops._lib.load()
    assert hasattr(lib, "xml_debug_set_q2c_variant"), "needs the debug library: XML_DEBUG=1 bash tvretrieval_amd/csrc/build.sh; XMLHIP_LIB=$PWD/tvretrieval_amd/csrc/libxmlhip_dbg.so"
    vl = inf.ragged_lengths(index)
    kw = dict(vid_len=vl) if vl is not None else {}
    for abl in [int(x) for x in os.environ.get("K7_ABLS", "0,1,2").split(",")]:
        lib.xml_debug_set_q2c_ablation(ctypes.c_int(abl))
        for _ in range(2):
            inf.stage_span_probs(m, index, qvec, ti, **kw)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for s, e in evs:
            s.record(); inf.stage_span_probs(m, index, qvec, ti, **kw); e.record()
        torch.cuda.synchronize()
        print("ablation %d: median %.3f ms" % (abl, sorted(s.elapsed_time(e) for s, e in evs)[2]))
    lib.xml_debug_set_q2c_ablation(ctypes.c_int(0))
