#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference, CPU, fp32).

Run in the development container only:   python tools/make_golden.py
The reference never travels; only the arrays written here do.  Each fixture holds the config (json),
the reference state_dict, the inputs and the reference's outputs for the hot-path rows of SURVEY.md 8c:
  (1) encode_context incl. padded rows   (2) encode_query   (3) get_pred_from_raw_query cross=True/False
  (4) the driver tail (top-k videos, flat-sorted moments) through the reference's own
      compute_context_info / compute_query2ctx_info with a duck-typed in-memory dataset
  (5) temporal NMS output   (6) one training step (losses, grads, BertAdam update) with recorded negatives.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.ref_import import import_reference  # noqa: E402
from tvretrieval_amd.easydict_compat import EasyDict  # noqa: E402

OUT_DIR = os.path.join(ROOT, "tests", "golden")


def model_cfg(**kw):
    cfg = dict(
        merge_two_stream=True, cross_att=True, span_predictor_type="conv", encoder_type="transformer",
        add_pe_rnn=False, pe_type="none", visual_input_size=96, sub_input_size=64, query_input_size=64,
        hidden_size=128, stack_conv_predictor_conv_kernel_sizes=-1, conv_kernel_size=5, conv_stride=1,
        max_ctx_l=40, max_desc_l=12, input_drop=0.1, cross_att_drop=0.1, drop=0.1, n_heads=4,
        initializer_range=0.02, ctx_mode="video_sub", margin=0.1, ranking_loss_type="hinge",
        lw_neg_q=1, lw_neg_ctx=1, lw_st_ed=0.01, use_hard_negative=False, hard_pool_size=20,
        use_self_attention=True, no_modular=False)
    cfg.update(kw)
    if "video" not in cfg["ctx_mode"] or "sub" not in cfg["ctx_mode"]:
        cfg["merge_two_stream"] = False      # xml/config.py:256-258
        cfg["cross_att"] = False
    return cfg


def bf16_grid(x):
    """Round to bf16-representable fp32 values: the fixtures then feed the fp32 and the bf16 HIP paths
    the very same numbers (and the zeroed low mantissa bits halve the compressed size)."""
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).to(torch.float32).numpy()


def l2n(x, eps=1e-5):
    return bf16_grid(x / (np.linalg.norm(x, axis=-1, keepdims=True) + eps))


def make_inputs(rng, n, lens, dim, max_l=None):
    max_l = max(lens) if max_l is None else max_l
    feat = np.zeros((n, max_l, dim), dtype=np.float32)
    mask = np.zeros((n, max_l), dtype=np.float32)
    for i, l in enumerate(lens):
        feat[i, :l] = l2n(rng.standard_normal((l, dim)).astype(np.float32))
        mask[i, :l] = 1
    return feat, mask


def perturb_weights(model, seed, scale=1.0):
    """reset_parameters() gives N(0, 0.02) weights, zero biases and unit LN: outputs of different videos
    are then nearly identical (score ties).  Fixtures use a richer but still deterministic init so that
    masks, biases and LN affine terms are all exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("LayerNorm.weight") or name.endswith("layernorm.weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif "predictor" in name:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif p.dim() >= 2:
                fan_in = p.shape[-1]
                p.copy_(scale * torch.randn(p.shape, generator=g) / np.sqrt(fan_in))
            p.copy_(p.to(torch.bfloat16).to(torch.float32))


def sd_arrays(model):
    return {"sd/" + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def gen_model_case(ns, name, cfg, seed, n_v, len_lo, len_hi, n_q, lq_lo, lq_hi, tail):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    model = ns.model_xml.XML(EasyDict(cfg))
    perturb_weights(model, seed + 1)
    model.eval()
    use_video = "video" in cfg["ctx_mode"]
    use_sub = "sub" in cfg["ctx_mode"]
    lens = rng.integers(len_lo, len_hi + 1, size=n_v)
    lens[rng.integers(0, n_v)] = len_hi
    qlens = rng.integers(lq_lo, lq_hi + 1, size=n_q)
    qlens[rng.integers(0, n_q)] = lq_hi
    out = dict(cfg=json.dumps(cfg), ctx_lens=lens, q_lens=qlens)
    vfeat, vmask = make_inputs(rng, n_v, lens, cfg["visual_input_size"])
    sfeat, smask = make_inputs(rng, n_v, lens, cfg["sub_input_size"])
    qfeat, qmask = make_inputs(rng, n_q, qlens, cfg["query_input_size"])
    out.update(video_feat=vfeat, video_mask=vmask, sub_feat=sfeat, sub_mask=smask,
               query_feat=qfeat, query_mask=qmask)
    out.update(sd_arrays(model))
    tv = lambda a: torch.from_numpy(a)
    dummy = torch.zeros(n_v, 2, 2)
    with torch.no_grad():
        # intermediates via hooks (first call of each module = the video/sub/query stream in call order)
        caps = {}

        def cap(key):
            def fn(mod, inp, res):
                caps.setdefault(key, []).append(res.detach().numpy().copy())
            return fn
        hooks = []
        first = "video" if use_video else "sub"
        for mname in [first + "_input_proj", "ctx_pos_embed", "query_encoder", "video_cross_att",
                      "video_cross_layernorm"]:
            if hasattr(model, mname):
                hooks.append(getattr(model, mname).register_forward_hook(cap(mname)))
        for mname in [first + "_encoder1"]:
            if hasattr(model, mname):
                hooks.append(getattr(model, mname).self.register_forward_hook(cap(mname + ".self")))
        v1, v2, s1, s2 = model.encode_context(tv(vfeat) if use_video else dummy, tv(vmask) if use_video else dummy,
                                              tv(sfeat) if use_sub else dummy, tv(smask) if use_sub else dummy)
        vq, sq = model.encode_query(tv(qfeat), tv(qmask))
        for h in hooks:
            h.remove()
        for k, lst in caps.items():
            out["int/%s" % k] = lst[0]
        for k, t in (("vf1", v1), ("vf2", v2), ("sf1", s1), ("sf2", s2)):
            if t is not None:
                out[k] = t.numpy().copy()
        out["video_query"] = vq.numpy().copy()
        out["sub_query"] = sq.numpy().copy()
        ctx_mask = vmask if use_video else smask
        q2c, st, ed = model.get_pred_from_raw_query(
            tv(qfeat), tv(qmask), v1, v2, tv(vmask) if use_video else None,
            s1, s2, tv(smask) if use_sub else None, cross=True)
        out.update(q2c_cross=q2c.numpy().copy(), st_cross=st.numpy().copy(), ed_cross=ed.numpy().copy())
        n_pair = min(n_q, n_v)
        idx = slice(0, n_pair)
        sel = lambda t: None if t is None else t[idx]
        q2c_b, st_b, ed_b = model.get_pred_from_raw_query(
            tv(qfeat)[idx], tv(qmask)[idx], sel(v1), sel(v2), tv(vmask)[idx] if use_video else None,
            sel(s1), sel(s2), tv(smask)[idx] if use_sub else None, cross=False)
        out.update(q2c_pair=q2c_b.numpy().copy(), st_pair=st_b.numpy().copy(), ed_pair=ed_b.numpy().copy())

        # tail, restated inline exactly as xml/inference.py:317-386 executes it (the real driver is
        # replayed end-to-end in the `pipeline_*` fixtures)
        alpha, kvid, min_l, max_l, nbefore = tail
        import torch.nn.functional as F
        w = torch.exp(alpha * q2c)
        stp = F.softmax(st, dim=-1)
        edp = F.softmax(ed, dim=-1)
        top_w, top_i = torch.topk(w, kvid, dim=1, largest=True)
        rows = torch.arange(0, len(stp)).unsqueeze(1)
        prod = torch.einsum("qvm,qv,qvn->qvmn", stp[rows, top_i], top_w, edp[rows, top_i])
        lmask = ns.inference.generate_min_max_length_mask(prod.shape, min_l=min_l, max_l=max_l)
        prod *= torch.from_numpy(lmask)
        fs, fi = torch.sort(prod.reshape(len(prod), -1), dim=1, descending=True)
        out.update(tail_params=np.array([alpha, kvid, min_l, max_l, nbefore], dtype=np.float64),
                   st_probs=stp.numpy().copy(), ed_probs=edp.numpy().copy(),
                   top_scores=top_w.numpy().copy(), top_indices=top_i.numpy().copy(),
                   flat_scores=fs[:, :nbefore].numpy().copy(), flat_indices=fi[:, :nbefore].numpy().copy())
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))
    return model, out


class FakeEvalDataset(torch.utils.data.Dataset):
    """Duck-typed stand-in for StartEndEvalDataset (xml/start_end_dataset.py:171-343): in-memory
    features, same item dicts, same set_data_mode / load_gt_vid_name_for_query switches."""

    def __init__(self, video_items, query_items, max_ctx_len, use_video, use_sub):
        self.video_items = video_items
        self.query_items = query_items
        self.max_ctx_len = max_ctx_len
        self.use_video, self.use_sub = use_video, use_sub
        self.video2idx = {v["vid_name"]: 1000 + 7 * i for i, v in enumerate(video_items)}
        self.query_data = [dict(desc_id=q["desc_id"], desc=q["desc"], vid_name=q["vid_name"]) for q in query_items]
        self.data_mode = "query"
        self.load_gt_video = False

    def set_data_mode(self, mode):
        self.data_mode = mode

    def load_gt_vid_name_for_query(self, flag):
        self.load_gt_video = flag

    def __len__(self):
        return len(self.query_items) if self.data_mode == "query" else len(self.video_items)

    def __getitem__(self, i):
        if self.data_mode == "context":
            v = self.video_items[i]
            mi = dict(video_feat=torch.from_numpy(v["video_feat"]) if self.use_video else torch.zeros(2, 2),
                      sub_feat=torch.from_numpy(v["sub_feat"]) if self.use_sub else torch.zeros(2, 2),
                      tef_feat=torch.zeros(2, 2))
            return dict(meta=dict(vid_name=v["vid_name"], duration=v["duration"]), model_inputs=mi)
        q = self.query_items[i]
        meta = dict(desc_id=q["desc_id"], desc=q["desc"],
                    vid_name=q["vid_name"] if self.load_gt_video else None,
                    ts=None)
        return dict(meta=meta, model_inputs=dict(query_feat=torch.from_numpy(q["query_feat"])))


def _pipeline_world(ns, cfg, seed, n_v, len_lo, len_hi, n_q):
    """Model + in-memory corpus / queries of a driver fixture (same draws, in the same order, for every caller)."""
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    model = ns.model_xml.XML(EasyDict(cfg))
    perturb_weights(model, seed + 1)
    model.eval()
    use_video = "video" in cfg["ctx_mode"]
    use_sub = "sub" in cfg["ctx_mode"]
    lens = rng.integers(len_lo, len_hi + 1, size=n_v)
    lens[0] = len_hi   # global max lives in the first context batch
    videos = []
    for i, l in enumerate(lens):
        videos.append(dict(vid_name="vid_%03d" % i, duration=float(l) * 1.5,
                           video_feat=l2n(rng.standard_normal((l, cfg["visual_input_size"])).astype(np.float32)),
                           sub_feat=l2n(rng.standard_normal((l, cfg["sub_input_size"])).astype(np.float32))))
    queries = []
    for i in range(n_q):
        lq = int(rng.integers(3, cfg["max_desc_l"] + 1))
        queries.append(dict(desc_id=5000 + i, desc="query %d" % i, vid_name="vid_%03d" % int(rng.integers(0, n_v)),
                            query_feat=l2n(rng.standard_normal((lq, cfg["query_input_size"])).astype(np.float32))))
    ds = FakeEvalDataset(videos, queries, cfg["max_ctx_l"], use_video, use_sub)
    return model, ds, videos, queries, lens, rng


def gen_pipeline_case(ns, name, cfg, seed, n_v, len_lo, len_hi, n_q, ctx_bsz, q_bsz, kvid, nbefore, nms_thd):
    """Runs the reference driver: compute_context_info -> compute_query2ctx_info(VCMR,SVMR,VR) -> NMS."""
    import argparse as ap
    model, ds, videos, queries, lens, rng = _pipeline_world(ns, cfg, seed, n_v, len_lo, len_hi, n_q)
    opt = ap.Namespace(eval_context_bsz=ctx_bsz, eval_query_bsz=q_bsz, num_workers=0, pin_memory=False,
                       device=torch.device("cpu"), ctx_mode=cfg["ctx_mode"], external_inference_vr_res_path=None,
                       max_ctx_l=cfg["max_ctx_l"], q2c_alpha=20.0, min_pred_l=2, max_pred_l=16, clip_length=1.5,
                       debug=False, max_before_nms=nbefore, max_vcmr_video=kvid)
    with torch.no_grad():
        ctx_info = ns.inference.compute_context_info(model, ds, opt)
        res = ns.inference.compute_query2ctx_info(model, ds, opt, ctx_info, max_before_nms=nbefore,
                                                  max_n_videos=kvid, tasks=("SVMR", "VCMR", "VR"))
    out = dict(cfg=json.dumps(cfg), ctx_lens=lens,
               opt=json.dumps(dict(eval_context_bsz=ctx_bsz, eval_query_bsz=q_bsz, q2c_alpha=20.0, min_pred_l=2,
                                   max_pred_l=16, clip_length=1.5, max_before_nms=nbefore, max_vcmr_video=kvid,
                                   nms_thd=nms_thd)))
    out.update(sd_arrays(model))
    for i, v in enumerate(videos):
        out["video_feat/%d" % i] = v["video_feat"]
        out["sub_feat/%d" % i] = v["sub_feat"]
    for i, q in enumerate(queries):
        out["query_feat/%d" % i] = q["query_feat"]
    out["query_gt_video"] = np.array([int(q["vid_name"][4:]) for q in queries], dtype=np.int64)
    out["video_idx"] = np.array([ds.video2idx[v["vid_name"]] for v in videos], dtype=np.int64)
    for k in ["video_feat1", "video_feat2", "video_mask", "sub_feat1", "sub_feat2", "sub_mask"]:
        if ctx_info[k] is not None:
            out["ctx/" + k] = ctx_info[k].numpy().copy()
    for task in ["VCMR", "SVMR", "VR"]:
        arr = np.array([[p for p in e["predictions"]] for e in res[task]], dtype=np.float64)
        out["res/" + task] = arr      # (Nq, n, 4): [video_idx, st, ed, score]
    # NMS rows (8f-1): reference post_processing_vcmr_nms / svmr on a deep copy
    import copy
    vc = ns.cal_inference.post_processing_vcmr_nms(copy.deepcopy(res["VCMR"]), nms_thd=nms_thd,
                                                   max_before_nms=nbefore, max_after_nms=100)
    sv = ns.cal_inference.post_processing_svmr_nms(copy.deepcopy(res["SVMR"]), nms_thd=nms_thd,
                                                   max_before_nms=nbefore, max_after_nms=100)
    for i, e in enumerate(vc):
        out["nms/VCMR/%d" % i] = np.array(e["predictions"], dtype=np.float64).reshape(-1, 4)
    for i, e in enumerate(sv):
        out["nms/SVMR/%d" % i] = np.array(e["predictions"], dtype=np.float64).reshape(-1, 4)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_external_vr_case(ns, name, cfg, seed, n_v, len_lo, len_hi, n_q, ctx_bsz, q_bsz, kvid, kext, nbefore):
    """The reference driver with opt.external_inference_vr_res_path set (xml/inference.py:244-249,264-273,349-355):
    another model's VR submission replaces top-k + exp(alpha * s).  The submission is synthetic -- kext > kvid distinct
    videos per query with descending cosine-like scores, so get_submission_top_n's trim to max_n_videos is exercised --
    and is stored in the fixture as arrays; the reference reads it back through its own load_external_vr_res2."""
    import argparse as ap
    import tempfile
    model, ds, videos, queries, lens, rng = _pipeline_world(ns, cfg, seed, n_v, len_lo, len_hi, n_q)
    names = [v["vid_name"] for v in videos]
    ext_vid = np.zeros((n_q, kext), dtype=np.int64)
    ext_score = np.zeros((n_q, kext), dtype=np.float64)
    vr = []
    for i, q in enumerate(queries):
        pick = rng.permutation(n_v)[:kext]
        sc = np.sort(rng.uniform(-0.2, 0.9, kext))[::-1]
        sc = np.round(sc.astype(np.float32).astype(np.float64), 6)
        ext_vid[i] = [ds.video2idx[names[int(j)]] for j in pick]
        ext_score[i] = sc
        vr.append(dict(desc_id=q["desc_id"], desc=q["desc"],
                       predictions=[[int(v), 0, 0, float(x)] for v, x in zip(ext_vid[i], sc)]))
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(dict(video2idx=ds.video2idx, VR=vr), f)
        ext_path = f.name
    opt = ap.Namespace(eval_context_bsz=ctx_bsz, eval_query_bsz=q_bsz, num_workers=0, pin_memory=False,
                       device=torch.device("cpu"), ctx_mode=cfg["ctx_mode"], external_inference_vr_res_path=ext_path,
                       max_ctx_l=cfg["max_ctx_l"], q2c_alpha=20.0, min_pred_l=2, max_pred_l=16, clip_length=1.5,
                       debug=False, max_before_nms=nbefore, max_vcmr_video=kvid)
    try:
        with torch.no_grad():
            ctx_info = ns.inference.compute_context_info(model, ds, opt)
            res = ns.inference.compute_query2ctx_info(model, ds, opt, ctx_info, max_before_nms=nbefore,
                                                      max_n_videos=kvid, tasks=("SVMR", "VCMR", "VR"))
    finally:
        os.unlink(ext_path)
    out = dict(cfg=json.dumps(cfg), ctx_lens=lens,
               opt=json.dumps(dict(eval_context_bsz=ctx_bsz, eval_query_bsz=q_bsz, q2c_alpha=20.0, min_pred_l=2,
                                   max_pred_l=16, clip_length=1.5, max_before_nms=nbefore, max_vcmr_video=kvid)))
    out.update(sd_arrays(model))
    for i, v in enumerate(videos):
        out["video_feat/%d" % i] = v["video_feat"]
        out["sub_feat/%d" % i] = v["sub_feat"]
    for i, q in enumerate(queries):
        out["query_feat/%d" % i] = q["query_feat"]
    out["query_gt_video"] = np.array([int(q["vid_name"][4:]) for q in queries], dtype=np.int64)
    out["video_idx"] = np.array([ds.video2idx[v["vid_name"]] for v in videos], dtype=np.int64)
    out["ext/video_idx"], out["ext/score"] = ext_vid, ext_score
    for task in ["VCMR", "SVMR", "VR"]:
        out["res/" + task] = np.array([[p for p in e["predictions"]] for e in res[task]], dtype=np.float64)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_train_case(ns, name, cfg, seed, bsz, len_lo, len_hi, lw_st_ed_schedule=None):
    """One training step of the reference: XML.forward -> backward -> BertAdam.step, with the CPU
    torch.randint draws of get_neg_scores (xml/model_xml.py:622) recorded in call order."""
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    model = ns.model_xml.XML(EasyDict(cfg))
    perturb_weights(model, seed + 1)
    model.eval()   # dropout off: the HIP training step is compared in eval-mode numerics
    lens = rng.integers(len_lo, len_hi + 1, size=bsz)
    lens[0] = len_hi
    qlens = rng.integers(3, cfg["max_desc_l"] + 1, size=bsz)
    vfeat, vmask = make_inputs(rng, bsz, lens, cfg["visual_input_size"])
    sfeat, smask = make_inputs(rng, bsz, lens, cfg["sub_input_size"])
    qfeat, qmask = make_inputs(rng, bsz, qlens, cfg["query_input_size"])
    st = np.array([rng.integers(0, l) for l in lens])
    ed = np.array([rng.integers(s, l) for s, l in zip(st, lens)])
    st_ed = np.stack([st, ed], axis=1).astype(np.int64)
    out = dict(cfg=json.dumps(cfg), video_feat=vfeat, video_mask=vmask, sub_feat=sfeat, sub_mask=smask,
               query_feat=qfeat, query_mask=qmask, st_ed_indices=st_ed)
    out.update({"sd_before/" + k: v.detach().numpy().copy() for k, v in model.state_dict().items()})
    draws = []
    orig_randint = torch.randint

    def rec_randint(*a, **kw):
        r = orig_randint(*a, **kw)
        draws.append(r.numpy().copy())
        return r
    torch.randint = rec_randint
    try:
        params = list(model.named_parameters())
        no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
        groups = [{"params": [p for n, p in params if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
                  {"params": [p for n, p in params if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
        okw = dict(lr=1e-4, weight_decay=0.01, warmup=0.1, t_total=10, schedule="warmup_linear")
        optim = ns.optimization.BertAdam(groups, **okw)
        tv = torch.from_numpy
        n_steps = 3 if lw_st_ed_schedule is None else len(lw_st_ed_schedule)
        step_losses = []
        for it in range(n_steps):          # the same batch three times: lr multiplier is 0 at step 0, then 1.0, 0.889
            if lw_st_ed_schedule is not None:
                # staged training (train_span_start_epoch, xml/train.py:47-48 -> set_train_st_ed): the span branch -- and
                # with it the *_query_linear / predictor / encoder2 / cross-attention tensors -- joins later; BertAdam
                # skips tensors whose .grad is None and counts steps per tensor (xml/optimization.py:289-291,325-330)
                model.config.lw_st_ed = lw_st_ed_schedule[it]
            loss, ld = model(tv(qfeat), tv(qmask), tv(vfeat), tv(vmask), tv(sfeat), tv(smask), None, None, tv(st_ed))
            # the reference's pinned torch 1.4 zeroes existing .grad tensors in place (never back to None)
            optim.zero_grad(set_to_none=False) if lw_st_ed_schedule is not None else optim.zero_grad()
            loss.backward()
            if it == 0:
                grads = {n: p.grad.detach().numpy().copy() for n, p in params if p.grad is not None}
                no_grad_first = [n for n, p in params if p.grad is None]
                first = (float(loss), dict(ld))
            step_losses.append(float(loss))
            optim.step()
            if it == 0:     # warmup_linear gives multiplier 0 at step 0: the first step only updates the moments
                for k, v in model.state_dict().items():
                    assert np.array_equal(v.detach().numpy(), out["sd_before/" + k]), k
    finally:
        torch.randint = orig_randint
    assert len(draws) == 2 * n_steps
    ld = first[1]
    out.update(neg_ctx_rank=draws[0], neg_q_rank=draws[1], loss=np.float64(first[0]),
               neg_ctx_rank_steps=np.stack(draws[0::2]), neg_q_rank_steps=np.stack(draws[1::2]),
               step_losses=np.array(step_losses, dtype=np.float64),
               loss_st_ed=np.float64(ld["loss_st_ed"]), loss_neg_ctx=np.float64(ld["loss_neg_ctx"]),
               loss_neg_q=np.float64(ld["loss_neg_q"]))
    out.update({"grad/" + k: v for k, v in grads.items()})
    out.update({"sd_after3/" + k: v.detach().numpy().copy() for k, v in model.state_dict().items()})   # = after the last step
    if lw_st_ed_schedule is not None:
        out["lw_st_ed_schedule"] = np.array(lw_st_ed_schedule, dtype=np.float64)
        out["no_grad_at_step0"] = json.dumps(no_grad_first)
        out["final_steps"] = json.dumps({n: int(optim.state[p]["step"]) if len(optim.state[p]) else 0 for n, p in params})
    out["optim"] = json.dumps(dict(b1=0.9, b2=0.999, e=1e-6, max_grad_norm=1.0, **okw))
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_eval_case(ns, name, seed, n_q, n_v, didemo, tasks=("VCMR", "SVMR", "VR"), p_right_vid=0.15, max_pred=121):
    """standalone_eval/eval.py eval_retrieval on a synthetic submission + ground truth (the reference's own
    known-answer input file is a missing large blob)."""
    rng = np.random.default_rng(seed)
    video2idx = {"v%03d" % i: 100 + 3 * i for i in range(n_v)}
    names = list(video2idx)
    gt, sub = [], dict(video2idx=video2idx, VCMR=[], SVMR=[], VR=[])
    for q in range(n_q):
        vn = names[int(rng.integers(0, n_v))]
        st = float(np.round(rng.uniform(0, 60), 2)); ed = st + float(np.round(rng.uniform(1.5, 20), 2))
        if didemo:
            ts = [[st + float(rng.integers(-1, 2)) * 5.0, ed + float(rng.integers(-1, 2)) * 5.0] for _ in range(4)]
        else:
            ts = [st, ed]
        gt.append(dict(desc_id=900 + q, desc="q%d" % q, type=["v", "t", "vt"][int(rng.integers(0, 3))], vid_name=vn,
                       ts=ts, duration=90.0))
        n_pred = int(rng.integers(3, max_pred))

        def moment(correct):
            if correct:
                return [round(st + float(rng.normal(0, 2.0)), 2), round(ed + float(rng.normal(0, 2.0)), 2)]
            a = float(np.round(rng.uniform(0, 70), 2))
            return [a, a + float(np.round(rng.uniform(1.5, 24), 2))]
        vc, sv, vr = [], [], []
        for j in range(n_pred):
            right_vid = rng.random() < p_right_vid
            v = video2idx[vn] if right_vid else video2idx[names[int(rng.integers(0, n_v))]]
            vc.append([v] + moment(right_vid and rng.random() < 0.6) + [float(1.0 / (j + 1))])
            sv.append([video2idx[vn]] + moment(rng.random() < 0.2) + [float(1.0 / (j + 1))])
        perm = rng.permutation(n_v)[:min(n_v, n_pred)]
        vr = [[video2idx[names[int(i)]], 0, 0, float(1.0 / (j + 1))] for j, i in enumerate(perm)]
        for k, lst in (("VCMR", vc), ("SVMR", sv), ("VR", vr)):
            sub[k].append(dict(desc_id=900 + q, desc="q%d" % q, predictions=lst))
    sub = {k: v for k, v in sub.items() if k == "video2idx" or k in tasks}
    metrics = ns.standalone_eval.eval_retrieval(sub, gt, iou_thds=(0.5, 0.7), verbose=False, match_number=True,
                                                use_desc_type=not didemo)
    path = os.path.join(OUT_DIR, name + ".json")
    with open(path, "w") as f:
        json.dump(dict(submission=sub, ground_truth=gt, use_desc_type=not didemo, metrics=metrics), f)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_ingest_case(ns, name, seed, n, dims, max_l, bsz):
    """Feature ingest: what the reference's dataset + collate hand to the model for a list of raw
    per-video feature arrays -- truncate to max_ctx_l (start_end_dataset.py:311,320), l2_normalize_np_array
    (utils/basic_utils.py:82-84), pad_sequences_1d per batch (utils/tensor_utils.py:5-53) -- computed by the reference's
    own functions.  Stored: the raw arrays (flattened + lengths) and, per batch, the padded features and masks with and
    without normalisation."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(3, max_l + 20, n)                       # some longer than max_l: truncation is exercised
    lens[1] = 1
    out = dict(lens=lens.astype(np.int64), max_l=np.int64(max_l), bsz=np.int64(bsz), dims=np.array(dims, dtype=np.int64))
    for tag, dim in zip(("video", "sub"), dims):
        raw = [(rng.standard_normal((int(l), dim)) * rng.uniform(0.05, 3.0)).astype(np.float32) for l in lens]
        raw[2][0] = 0.0                                         # an all-zero clip: x / (0 + eps) stays 0
        out["raw/" + tag] = np.concatenate(raw, 0)
        for norm in (True, False):
            for b in range(0, n, bsz):
                seqs = []
                for a in raw[b:b + bsz]:
                    a = a[:max_l]
                    if norm:
                        a = ns.basic_utils.l2_normalize_np_array(a)
                    seqs.append(torch.from_numpy(np.ascontiguousarray(a)))
                padded, mask = ns.tensor_utils.pad_sequences_1d(seqs, dtype=torch.float32)
                key = "%s/%s/batch%d" % (tag, "norm" if norm else "raw", b // bsz)
                out[key + "/feat"] = padded.numpy()
                out[key + "/mask"] = mask.numpy()
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def main():
    only = sys.argv.pop(1) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else ""
    if only:      # regenerate a subset: python tools/make_golden.py train_step
        g = globals()
        for fn in ("gen_model_case", "gen_pipeline_case", "gen_external_vr_case", "gen_eval_case", "gen_train_case",
                   "gen_ingest_case"):
            orig = g[fn]
            g[fn] = (lambda o: lambda ns, name, *a, **k: o(ns, name, *a, **k) if only in name else None)(orig)
    ap = argparse.ArgumentParser()
    ap.parse_args()
    os.makedirs(OUT_DIR, exist_ok=True)
    ns = import_reference()
    torch.set_num_threads(1)   # deterministic summation order for the committed vectors
    tail = (20.0, 5, 2, 16, 60)
    gen_model_case(ns, "xml_video_sub_cross_h128", model_cfg(), 11, n_v=10, len_lo=9, len_hi=40,
                   n_q=7, lq_lo=3, lq_hi=12, tail=tail)
    gen_model_case(ns, "xml_video_only_h256", model_cfg(ctx_mode="video", hidden_size=256, visual_input_size=128,
                                                         max_ctx_l=24),
                   12, n_v=6, len_lo=6, len_hi=24, n_q=6, lq_lo=3, lq_hi=12, tail=(20.0, 4, 2, 16, 40))
    gen_model_case(ns, "xml_sub_only_h128", model_cfg(ctx_mode="sub", max_ctx_l=24), 13, n_v=6, len_lo=5,
                   len_hi=24, n_q=5, lq_lo=3, lq_hi=12, tail=(20.0, 3, 2, 16, 30))
    gen_model_case(ns, "xml_video_sub_nocross_nomerge_h128",
                   model_cfg(cross_att=False, merge_two_stream=False, max_ctx_l=24), 14, n_v=6, len_lo=5,
                   len_hi=24, n_q=5, lq_lo=3, lq_hi=12, tail=(20.0, 3, 2, 16, 30))
    gen_pipeline_case(ns, "pipeline_video_sub_h128", model_cfg(max_ctx_l=36), 21, n_v=13, len_lo=8, len_hi=36,
                      n_q=9, ctx_bsz=5, q_bsz=4, kvid=6, nbefore=50, nms_thd=0.5)
    gen_pipeline_case(ns, "pipeline_video_only_h128",
                      model_cfg(ctx_mode="video", max_ctx_l=30), 22, n_v=11, len_lo=8, len_hi=30,
                      n_q=7, ctx_bsz=4, q_bsz=3, kvid=5, nbefore=40, nms_thd=0.5)
    gen_external_vr_case(ns, "pipeline_external_vr_h128", model_cfg(max_ctx_l=36), 23, n_v=14, len_lo=8, len_hi=36,
                         n_q=9, ctx_bsz=5, q_bsz=4, kvid=6, kext=8, nbefore=50)
    gen_eval_case(ns, "eval_tvr_style", 41, n_q=60, n_v=25, didemo=False)
    gen_eval_case(ns, "eval_didemo_style", 42, n_q=30, n_v=12, didemo=True)
    # more of the evaluator's input space: tiny and large problems, short prediction lists (fewer than the recall cut-offs),
    # submissions with a subset of the tasks, hardly any / mostly correct videos
    gen_eval_case(ns, "eval_more_tiny", 43, n_q=4, n_v=3, didemo=False, max_pred=8)
    gen_eval_case(ns, "eval_more_large", 44, n_q=80, n_v=70, didemo=False)
    gen_eval_case(ns, "eval_more_vcmr_only", 45, n_q=40, n_v=20, didemo=False, tasks=("VCMR",), p_right_vid=0.02)
    gen_eval_case(ns, "eval_more_svmr_vr_didemo", 46, n_q=35, n_v=9, didemo=True, tasks=("SVMR", "VR"), p_right_vid=0.6)
    gen_train_case(ns, "train_step_video_sub_h128", model_cfg(max_ctx_l=24, lw_st_ed=0.01, visual_input_size=48, sub_input_size=32,
                                                           query_input_size=32), 31, bsz=6,
                   len_lo=6, len_hi=24)
    gen_train_case(ns, "train_step_nocross_lse_h128",
                   model_cfg(max_ctx_l=20, lw_st_ed=0.5, visual_input_size=40, sub_input_size=32, query_input_size=32,
                             cross_att=False, merge_two_stream=False, ranking_loss_type="lse", use_hard_negative=True,
                             hard_pool_size=3), 32, bsz=7, len_lo=5, len_hi=19)
    gen_ingest_case(ns, "ingest_collate", 51, n=11, dims=(48, 32), max_l=40, bsz=4)
    # staged: two steps without the span loss, then three with it (t_total 10, warmup 0.1: the late tensors see multiplier
    # 0 at THEIR step 0 while the early ones are already decaying)
    gen_train_case(ns, "train_step_staged_video_sub_h128",
                   model_cfg(max_ctx_l=24, lw_st_ed=0.0, visual_input_size=48, sub_input_size=32, query_input_size=32), 33,
                   bsz=6, len_lo=6, len_hi=24, lw_st_ed_schedule=[0.0, 0.0, 0.01, 0.01, 0.01])


if __name__ == "__main__":
    main()
