"""End-to-end eval_epoch at the reference's as-trained shape (bench.WORKLOADS["tvr_val"]: TVR val, 10 895 queries x 2 179
videos, H = 256, max_ctx_l = 100, real clip counts): the wall-clock a user of xml/inference.py:473-531 sees AFTER the corpus
is encoded -- compute_query2ctx_info(tasks = VCMR, SVMR, VR) in batches of eval_query_bsz, K10 records + one D2H per task,
get_submission_top_n, eval_retrieval, temporal NMS at 0.5, eval_retrieval again -- split into its stages, plus the cost of
materialising the reference's nested lists (only needed to write the JSON submission).  Synthetic ground truth: each query's
"correct" video is drawn at random, the span is 2..10 clips.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class SyntheticQueries(object):
    """The reference's eval-dataset contract (xml/start_end_dataset.py:171-343) in "query" mode over synthetic features."""

    def __init__(self, qf, qm, gt_video, n_videos):
        self.qf = qf.cpu().numpy()
        self.lens = qm.sum(1).long().cpu().numpy()
        self.gt_video = gt_video
        self.video2idx = {"v%05d" % i: 3 * i + 7 for i in range(n_videos)}
        self.gt = False

    def set_data_mode(self, mode):
        assert mode == "query"

    def load_gt_vid_name_for_query(self, flag):
        self.gt = flag

    def __len__(self):
        return len(self.qf)

    def __getitem__(self, i):
        meta = dict(desc_id=90000 + i, desc="synthetic query %d" % i, vid_name="v%05d" % self.gt_video[i] if self.gt else None)
        return dict(meta=meta, model_inputs=dict(query_feat=self.qf[i, :self.lens[i]]))


def run(query_bsz=50, nms_thd=0.5, max_before_nms=200, workload="tvr_val", n_queries=None, repeats=2, graph=True):
    import bench
    from tvretrieval_amd import inference as inf
    from tvretrieval_amd.model_xml import XML
    from tvretrieval_amd.results import to_lists
    nq, nv, l, hidden, dv, ds, dq, ctx_mode, _ = bench.WORKLOADS[workload]
    nq = n_queries or nq
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(0)
    model = XML(bench.model_config(hidden, dv, ds, dq, ctx_mode, l), compute_dtype=torch.bfloat16).to(dev).eval()
    lens = bench.real_clip_counts(nv, l) if workload in bench.RAGGED else None
    with torch.no_grad():
        index = inf.build_corpus_index(model, bench.context_batches(0, nv, l, dv, ds, True, ctx_mode == "video_sub", dev, lens),
                                       n_total=nv, l_ref=l)
    qf, qm = bench.synth_queries(nq, dq, dev)
    rng = np.random.default_rng(2018)
    gt_video = rng.integers(0, nv, nq)
    ds_q = SyntheticQueries(qf, qm, gt_video, nv)
    ctx = dict(index=index, video_metas=[dict(vid_name="v%05d" % i) for i in range(nv)])
    clip = 1.5
    st = rng.integers(0, 40, nq)
    gt = [dict(desc_id=90000 + i, desc="", type=["v", "t", "vt"][i % 3], vid_name="v%05d" % gt_video[i],
               ts=[float(st[i] * clip), float((st[i] + rng.integers(2, 11)) * clip)]) for i in range(nq)]
    opt = argparse.Namespace(eval_query_bsz=query_bsz, device=dev, q2c_alpha=20.0, min_pred_l=2, max_pred_l=16,
                             clip_length=clip, debug=False, external_inference_vr_res_path=None, max_ctx_l=l,
                             max_before_nms=max_before_nms, max_vcmr_video=100, nms_thd=nms_thd, dset_name="tvr",
                             graph_search=graph, max_desc_l=int(qm.shape[1]))
    best = None
    for _ in range(repeats + 1):          # first pass = warm-up (workspaces, weight packing, allocator)
        tm = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            sub, met, sub_nms, met_nms = inf.eval_epoch(model, ds_q, opt, tasks=("VCMR", "SVMR", "VR"), ground_truth=gt,
                                                        as_arrays=True, timings=tm, ctx_info=ctx)
        tm["total"] = time.perf_counter() - t0
        if best is None or tm["total"] < best["total"]:
            best = tm
    # the device part alone, for the split: the same batches, results left on the device
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for b in range(0, nq, query_bsz):
            inf.vcmr_search(model, index, qf[b:b + query_bsz], qm[b:b + query_bsz], max_vcmr_video=100,
                            max_before_nms=max_before_nms,
                            svmr_video=torch.from_numpy(gt_video[b:b + query_bsz].astype(np.int32)).to(dev))
    torch.cuda.synchronize()
    search_only = time.perf_counter() - t0
    t0 = time.perf_counter()
    lists = to_lists(sub)
    lists_nms = to_lists(sub_nms)
    t_lists = time.perf_counter() - t0
    n_rows = sum(len(e["predictions"]) for k in ("VCMR", "SVMR", "VR") for e in lists[k])
    n_rows += sum(len(e["predictions"]) for k in ("VCMR", "SVMR") for e in lists_nms[k])
    t0 = time.perf_counter()
    blob = json.dumps(lists)
    t_json = time.perf_counter() - t0
    host_tail = best["top_n"] + best["eval"] + best["nms"] + best["eval_nms"]
    return {"workload": workload, "queries": nq, "videos": nv, "eval_query_bsz": query_bsz, "nms_thd": nms_thd,
            "max_before_nms": max_before_nms, "tasks": ["VCMR", "SVMR", "VR"], "graph_search": bool(graph),
            "total_s": best["total"], "queries_per_s": nq / best["total"],
            "stage_s": {k: round(v, 4) for k, v in best.items()},
            "search_device_only_s": round(search_only, 4),
            "search_host_overhead_s": round(best["search"] - search_only, 4),
            "host_tail_s": round(host_tail, 4), "nms_s": round(best["nms"], 4),
            "lists_on_demand_s": round(t_lists, 4), "list_rows": n_rows, "json_dumps_s": round(t_json, 4),
            "json_mb": round(len(blob) / 1e6, 1),
            "metrics_sample": {"VCMR": met["VCMR"]["0.5-r100"], "SVMR": met["SVMR"]["0.7-r1"], "VR": met["VR"]["r100"],
                               "VCMR_nms": met_nms["VCMR"]["0.5-r100"]},
            "host_threads": os.cpu_count(),
            "what": "eval_epoch after the corpus encode: search = dataset items + pad + K1..K10 per batch of eval_query_bsz + "
                    "ONE D2H per task; host_tail = get_submission_top_n + eval_retrieval + batched NMS + eval_retrieval on "
                    "(Nq, n) arrays; lists_on_demand = the reference's nested lists for all five result sets, built in C"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--bsz", type=int, default=50)
    ap.add_argument("--queries", type=int, default=None)
    ap.add_argument("--workload", default="tvr_val")
    ap.add_argument("--eager", action="store_true", help="opt.graph_search off: every batch as its own chain of launches")
    a = ap.parse_args()
    print(json.dumps(run(a.bsz, workload=a.workload, n_queries=a.queries, graph=not a.eager)))
