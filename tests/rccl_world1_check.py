"""Run under `python -m torch.distributed.run --nproc-per-node 1` on a GPU box (tests/test_gpu_dist.py does):
pushes every collective of the sharded VCMR pass and of the data-parallel gradient averaging through a REAL
single-rank RCCL group (backend "nccl"), and checks the result against the unsharded search.  This is what one
GPU can verify of the N > 1 path: dtypes (f32 / int32 / bf16), contiguity, API usage, stream ordering."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_golden  # noqa: E402


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    from tvretrieval_amd import dist as xd
    from tvretrieval_amd import inference as inf
    from tvretrieval_amd.model_xml import XML
    from tvretrieval_amd.train import allreduce_gradients
    xd.SKIP_TRIVIAL_COLLECTIVES = False
    d, cfg, sd = load_golden("xml_video_sub_cross_h128")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)       # noqa: E731
    for dtype in (torch.float32, torch.bfloat16):
        m = XML(cfg, compute_dtype=dtype)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        m = m.to(dev).eval()
        with torch.no_grad():
            index = inf.build_corpus_index(m, [(T(d["video_feat"]), T(d["video_mask"]), T(d["sub_feat"]), T(d["sub_mask"]))])
            qf, qm = T(d["query_feat"]), T(d["query_mask"])
            want = inf.vcmr_search(m, index, qf, qm, max_vcmr_video=6, max_before_nms=50)
            got = xd.sharded_vcmr_search(m, index, qf, qm, max_vcmr_video=6, max_before_nms=50)
            own = xd.sharded_vcmr_search(m, index, qf, qm, max_vcmr_video=6, max_before_nms=50, gather_results=False)
            xd.replicate_rerank_features(index)       # owner rerank: all-gather of feat2 / masks, 2-collective pass
            assert index.feat2_all["video"].data_ptr() != index.feat2["video"].data_ptr()
            got2 = xd.sharded_vcmr_search(m, index, qf, qm, max_vcmr_video=6, max_before_nms=50)
            own2 = xd.sharded_vcmr_search(m, index, qf, qm, max_vcmr_video=6, max_before_nms=50, gather_results=False)
        for k in ("top_scores", "top_indices", "flat_scores", "flat_indices"):
            assert torch.equal(got[k], want[k]), (str(dtype), k)
            assert torch.equal(own[k], want[k]), (str(dtype), k, "owner slice")      # world 1: the slice is everything
            assert torch.equal(got2[k], want[k]), (str(dtype), k, "owner rerank")
            assert torch.equal(own2[k], want[k]), (str(dtype), k, "owner rerank, owner slice")

    # ---- the C-ABI collectives (xml_rccl_*) on a 1-rank communicator of our own vs the torch.distributed exchange ----
    ex_c, ex_t = xd.RcclExchange(), xd.TorchExchange()
    assert ex_c.world == 1 and ex_c.rank == 0
    from tvretrieval_amd import ops
    g = torch.Generator(device=dev).manual_seed(3)
    s = torch.randn(37, 2000, device=dev, generator=g)
    loc_s, loc_i = ops.topk_rows(s, 100, alpha=0.0)
    loc_i = (loc_i + 5000).contiguous()
    a = ex_c.topk_by_owner(loc_s, loc_i, 100, 20.0, ops)
    b = ex_t.topk_by_owner(loc_s, loc_i, 100, 20.0, ops)
    torch.cuda.synchronize()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    a = ex_c.allgather_topk(loc_s, loc_i, 100, 20.0, ops)
    b = ex_t.allgather_topk(loc_s, loc_i, 100, 20.0, ops)
    torch.cuda.synchronize()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    rows = torch.randn(11, 2, 64, device=dev, generator=g).to(torch.bfloat16)
    assert torch.equal(ex_c.allgather_rows(rows), ex_t.allgather_rows(rows))
    # the chunked, stream-pipelined owner pass through the C-ABI exchange
    for dtype in (torch.bfloat16,):
        m = XML(cfg, compute_dtype=dtype)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        m = m.to(dev).eval()
        with torch.no_grad():
            index = inf.build_corpus_index(m, [(T(d["video_feat"]), T(d["video_mask"]), T(d["sub_feat"]), T(d["sub_mask"]))])
            xd.replicate_rerank_features(index)
            qf, qm = T(d["query_feat"]), T(d["query_mask"])
            want = inf.vcmr_search(m, index, qf, qm, max_vcmr_video=6, max_before_nms=50)
            for n_chunks in (1, 3):
                got = xd.sharded_vcmr_search(m, index, qf, qm, max_vcmr_video=6, max_before_nms=50, exchange=ex_c,
                                             n_chunks=n_chunks, gather_results=False)
                torch.cuda.synchronize()
                for k in ("top_scores", "top_indices", "flat_scores", "flat_indices"):
                    assert torch.equal(got[k], want[k]), (k, n_chunks)
    # gradient buckets reduced under backward through xml_rccl_allreduce_avg_f32 (1 rank: the average is the identity)
    from tvretrieval_amd.train import BertAdam, GradientReducer, allreduce_gradients, xml_forward_train
    dt, tcfg, _ = load_golden("train_step_video_sub_h128")
    results = []
    for with_reducer in (False, True):
        tm = XML(tcfg)
        tm.load_state_dict({k[len("sd_before/"):]: torch.from_numpy(v.copy()) for k, v in dt.items() if k.startswith("sd_before/")})
        tm = tm.to(dev)
        tm.eval()
        opt = BertAdam(tm.parameters(), lr=1e-3, warmup=-1, t_total=-1, schedule="none")
        if with_reducer:
            red = GradientReducer(opt, bucket_bytes=64 << 10)
            assert red.comm is not None and len(red.buckets) > 3
        batch = dict(query_feat=T(dt["query_feat"]), query_mask=T(dt["query_mask"]), video_feat=T(dt["video_feat"]),
                     video_mask=T(dt["video_mask"]), sub_feat=T(dt["sub_feat"]), sub_mask=T(dt["sub_mask"]),
                     st_ed_indices=T(dt["st_ed_indices"]), neg_ctx_rank=dt["neg_ctx_rank"], neg_q_rank=dt["neg_q_rank"])
        for _ in range(2):       # gradients after forward / backward / all-reduce (second pass: reducer state was reset)
            loss, _ = xml_forward_train(tm, **batch)
            opt.zero_grad()
            loss.backward()
            allreduce_gradients(opt)
        torch.cuda.synchronize()
        results.append(opt.flat_g.clone())
    # the captured step with the reducer's RCCL all-reduces INSIDE the graph (1 rank: the average is the identity):
    # three replays == three eager steps from the same state
    from tvretrieval_amd.train import GraphedTrainStep
    finals = []
    for graphed in (False, True):
        tm = XML(tcfg)
        tm.load_state_dict({k[len("sd_before/"):]: torch.from_numpy(v.copy()) for k, v in dt.items() if k.startswith("sd_before/")})
        tm = tm.to(dev)
        tm.eval()
        opt = BertAdam(tm.parameters(), lr=1e-3, warmup=-1, t_total=-1, schedule="none")
        red = GradientReducer(opt, bucket_bytes=64 << 10)
        assert red.comm is not None
        tb = {k: v for k, v in batch.items() if k not in ("neg_ctx_rank", "neg_q_rank")}
        if graphed:
            step = GraphedTrainStep(tm, opt, tb)
            for _ in range(3):
                step(tb, neg_ctx_rank=dt["neg_ctx_rank"], neg_q_rank=dt["neg_q_rank"])
        else:
            for _ in range(3):
                loss, _ = xml_forward_train(tm, **batch)
                opt.zero_grad()
                loss.backward()
                allreduce_gradients(opt)
                opt.step()
        torch.cuda.synchronize()
        finals.append(opt.flat_p.clone())
    rel = float((finals[0] - finals[1]).abs().max()) / float(finals[0].abs().max())
    assert rel <= 2e-5, rel
    # (weight gradients use split-K f32 atomics: equal to rounding, not bitwise, between any two runs)
    scale = float(results[0].abs().max())
    assert float((results[0] - results[1]).abs().max()) <= 1e-5 * scale, float((results[0] - results[1]).abs().max()) / scale

    class Holder(object):
        pass
    h = Holder()
    h.flat_g = torch.randn(1 << 20, device=dev)
    ref = h.flat_g.clone()
    allreduce_gradients(h, bucket_bytes=1 << 20)      # 4 buckets, ReduceOp.AVG on RCCL
    torch.cuda.synchronize()
    assert torch.allclose(h.flat_g, ref)
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_WORLD1_OK")


if __name__ == "__main__":
    main()
