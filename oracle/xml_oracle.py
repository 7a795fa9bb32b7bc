"""CPU oracle for the XML moment-retrieval hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.
The product (`tvretrieval_amd/`) never calls it: the HIP path has no CPU fallback.

What it is: a plain torch-CPU fp32 restatement, written for this repo, of the algorithm of the
reference jayleicn/TVRetrieval XML model and its VCMR/SVMR/VR ranking tail.  It keeps the
reference's *formulation* (same contraction strings, same masking constants, same op order where
fp32 rounding is order dependent) because it doubles as the "reference CPU path" baseline in
bench.py.  Every function cites the reference file:line it follows; `xml/` abbreviates
`baselines/crossmodal_moment_localization/`.

Parity pinning: the reference ships no tests or golden vectors for this path,
so the oracle is pinned against outputs of the reference itself, imported in the development
container by tools/make_golden.py and committed as tests/golden/*.npz
(tests/test_oracle_golden.py replays them).

The model is expressed functionally over a flat `state_dict` (reference key names)
and a config mapping, so fixtures only need to carry arrays.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

NEG_FILL = -1e10          # mask_logits constant, xml/model_xml.py:640-641
ATT_NEG = -10000.0        # additive attention mask, xml/model_components.py:277
LN_EPS = 1e-5             # torch.nn.LayerNorm default used everywhere in the reference


def _t(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x)
    return x


class Weights(object):
    """Flat state_dict view with prefix navigation."""

    def __init__(self, sd, prefix=""):
        self.sd = sd
        self.prefix = prefix

    def sub(self, name):
        return Weights(self.sd, self.prefix + name + ".")

    def __getitem__(self, name):
        return _t(self.sd[self.prefix + name]).float()

    def has(self, name):
        return (self.prefix + name) in self.sd


def mask_logits(target, mask):
    """xml/model_xml.py:640-641 (multiplicative form: masked inputs must be finite)."""
    return target * mask + (1 - mask) * NEG_FILL


def layer_norm(x, w, prefix):
    return F.layer_norm(x, (x.shape[-1],), w[prefix + ".weight"], w[prefix + ".bias"], LN_EPS)


def linear_layer(x, w):
    """LinearLayer.forward, xml/model_components.py:156-163 (eval: dropout is identity)."""
    x = layer_norm(x, w, "LayerNorm")
    x = F.linear(x, w["net.1.weight"], w["net.1.bias"])
    return F.relu(x)


def trainable_pos_enc(x, w):
    """TrainablePositionalEncoding.forward, xml/model_components.py:76-89."""
    seq_l = x.shape[1]
    pos = w["position_embeddings.weight"][:seq_l]
    return layer_norm(x + pos.unsqueeze(0), w, "LayerNorm")


def bert_self_attention(q_states, k_states, v_states, att_mask, w, n_heads):
    """BertSelfAttention.forward, xml/model_components.py:266-303.
    att_mask: (N, Lq or 1, Lk) float 1=valid."""
    n, lq, hsz = q_states.shape
    lk = k_states.shape[1]
    dh = hsz // n_heads
    add_mask = (1 - att_mask.unsqueeze(1)) * ATT_NEG
    q = F.linear(q_states, w["query.weight"], w["query.bias"]).view(n, lq, n_heads, dh).permute(0, 2, 1, 3)
    k = F.linear(k_states, w["key.weight"], w["key.bias"]).view(n, lk, n_heads, dh).permute(0, 2, 1, 3)
    v = F.linear(v_states, w["value.weight"], w["value.bias"]).view(n, lk, n_heads, dh).permute(0, 2, 1, 3)
    scores = torch.matmul(q, k.transpose(-1, -2))
    scores = scores / math.sqrt(dh)
    scores = scores + add_mask
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs, v)
    return ctx.permute(0, 2, 1, 3).contiguous().view(n, lq, hsz)


def bert_self_output(hidden, residual, w):
    """BertSelfOutput.forward, xml/model_components.py:313-317."""
    hidden = F.linear(hidden, w["dense.weight"], w["dense.bias"])
    return layer_norm(hidden + residual, w, "LayerNorm")


def bert_attention(x, att_mask, w, n_heads):
    """BertAttention.forward, xml/model_components.py:207-216 (self-attention + output, no FFN)."""
    a = bert_self_attention(x, x, x, att_mask, w.sub("self"), n_heads)
    return bert_self_output(a, x, w.sub("output"))


class OracleXML(object):
    """Functional mirror of XML (xml/model_xml.py:52-641), transformer encoder, conv span predictor."""

    def __init__(self, config, state_dict):
        self.cfg = dict(config)
        self.w = Weights(state_dict)
        self.use_video = "video" in self.cfg["ctx_mode"]
        self.use_sub = "sub" in self.cfg["ctx_mode"]
        self.n_heads = int(self.cfg["n_heads"])
        assert self.cfg.get("encoder_type", "transformer") == "transformer"
        assert self.cfg.get("span_predictor_type", "conv") == "conv"

    # ---- encoders ------------------------------------------------------------------------
    def encode_input(self, feat, mask, proj_name, enc_name, pos_name):
        """xml/model_xml.py:377-392."""
        x = linear_layer(_t(feat).float(), self.w.sub(proj_name))
        x = trainable_pos_enc(x, self.w.sub(pos_name))
        return bert_attention(x, _t(mask).float().unsqueeze(1), self.w.sub(enc_name), self.n_heads)

    def cross_context_encoder(self, main, main_mask, side, side_mask, name):
        """xml/model_xml.py:357-373."""
        cross_mask = torch.einsum("bm,bn->bmn", main_mask, side_mask)
        cross = bert_self_attention(main, side, side, cross_mask, self.w.sub(name + "_cross_att"), self.n_heads)
        res = layer_norm(cross + main, self.w, name + "_cross_layernorm")
        return bert_attention(res, main_mask.unsqueeze(1), self.w.sub(name + "_encoder2"), self.n_heads)

    def cross_encode_context(self, video_feat, video_mask, sub_feat, sub_mask):
        """xml/model_xml.py:344-355."""
        video_mask = _t(video_mask).float()
        sub_mask = _t(sub_mask).float()
        v1 = self.encode_input(video_feat, video_mask, "video_input_proj", "video_encoder1", "ctx_pos_embed")
        s1 = self.encode_input(sub_feat, sub_mask, "sub_input_proj", "sub_encoder1", "ctx_pos_embed")
        v2 = self.cross_context_encoder(v1, video_mask, s1, sub_mask, "video")
        s2 = self.cross_context_encoder(s1, sub_mask, v1, video_mask, "sub")
        return v1, v2, s1, s2

    def non_cross_encode_context(self, feat, mask, name):
        """xml/model_xml.py:297-329 (three attention layers: enc1 -> feat1; enc2 -> enc3 -> feat2)."""
        mask = _t(mask).float()
        f1 = self.encode_input(feat, mask, name + "_input_proj", name + "_encoder1", "ctx_pos_embed")
        m = mask.unsqueeze(1)
        f2 = bert_attention(f1, m, self.w.sub(name + "_encoder2"), self.n_heads)
        f2 = bert_attention(f2, m, self.w.sub(name + "_encoder3"), self.n_heads)
        return f1, f2

    def encode_context(self, video_feat, video_mask, sub_feat, sub_mask):
        """xml/model_xml.py:331-342."""
        if self.cfg["cross_att"]:
            assert self.use_video and self.use_sub
            return self.cross_encode_context(video_feat, video_mask, sub_feat, sub_mask)
        v1 = v2 = s1 = s2 = None
        if self.use_video:
            v1, v2 = self.non_cross_encode_context(video_feat, video_mask, "video")
        if self.use_sub:
            s1, s2 = self.non_cross_encode_context(sub_feat, sub_mask, "sub")
        return v1, v2, s1, s2

    def encoded_query_tokens(self, query_feat, query_mask):
        return self.encode_input(query_feat, _t(query_mask).float(), "query_input_proj", "query_encoder",
                                 "query_pos_embed")

    def get_modularized_queries(self, encoded_query, query_mask):
        """xml/model_xml.py:410-423 (no_modular=False path)."""
        assert not self.cfg.get("no_modular", False)
        query_mask = _t(query_mask).float()
        sc = F.linear(encoded_query, self.w["modular_vector_mapping.weight"])  # (N, L, 1|2)
        sc = torch.softmax(mask_logits(sc, query_mask.unsqueeze(2)), dim=1)
        mq = torch.einsum("blm,bld->bmd", sc, encoded_query)
        if mq.shape[1] == 2:
            return mq[:, 0], mq[:, 1]
        return mq[:, 0], mq[:, 0]

    def encode_query(self, query_feat, query_mask):
        """xml/model_xml.py:291-295."""
        enc = self.encoded_query_tokens(query_feat, query_mask)
        return self.get_modularized_queries(enc, query_mask)

    # ---- scores --------------------------------------------------------------------------
    def get_video_level_scores(self, modular_query, ctx_feat1, ctx_mask):
        """xml/model_xml.py:436-453: cosine vs every clip, masked, max over clips."""
        q = F.normalize(modular_query, dim=-1)
        c = F.normalize(_t(ctx_feat1).float(), dim=-1)
        s = torch.einsum("md,nld->mln", q, c)
        m = _t(ctx_mask).float().transpose(0, 1).unsqueeze(0)
        s = mask_logits(s, m)
        return torch.max(s, dim=1)[0]

    def _conv(self, sim, wname):
        k = self.w[wname]                      # (1,1,ks)
        return F.conv1d(sim, k, None, int(self.cfg.get("conv_stride", 1)), k.shape[-1] // 2)

    def get_merged_st_ed_prob(self, video_query, video_feat2, sub_query, sub_feat2, ctx_mask, cross):
        """xml/model_xml.py:455-502 with stack_conv disabled (inference forces -1, xml/inference.py:538)."""
        vq = F.linear(video_query, self.w["video_query_linear.weight"], self.w["video_query_linear.bias"])
        sq = F.linear(sub_query, self.w["sub_query_linear.weight"], self.w["sub_query_linear.bias"])
        video_feat2 = _t(video_feat2).float()
        sub_feat2 = _t(sub_feat2).float()
        ctx_mask = _t(ctx_mask).float()
        if cross:
            vs = torch.einsum("md,nld->mnl", vq, video_feat2)
            ss = torch.einsum("md,nld->mnl", sq, sub_feat2)
            sim = (vs + ss) / 2
            nq, nc, l = sim.shape
            sim = sim.view(nq * nc, 1, l)
            st = self._conv(sim, "merged_st_predictor.weight").view(nq, nc, l)
            ed = self._conv(sim, "merged_ed_predictor.weight").view(nq, nc, l)
        else:
            vs = torch.einsum("bd,bld->bl", vq, video_feat2)
            ss = torch.einsum("bd,bld->bl", sq, sub_feat2)
            sim = ((vs + ss) / 2).unsqueeze(1)
            st = self._conv(sim, "merged_st_predictor.weight").squeeze(1)
            ed = self._conv(sim, "merged_ed_predictor.weight").squeeze(1)
        return mask_logits(st, ctx_mask), mask_logits(ed, ctx_mask)

    def get_st_ed_prob(self, modular_query, ctx_feat2, ctx_mask, name, cross):
        """xml/model_xml.py:504-551 (conv span predictor)."""
        q = F.linear(modular_query, self.w[name + "_query_linear.weight"], self.w[name + "_query_linear.bias"])
        ctx_feat2 = _t(ctx_feat2).float()
        ctx_mask = _t(ctx_mask).float()
        if cross:
            sim = torch.einsum("md,nld->mnl", q, ctx_feat2)
            nq, nc, l = sim.shape
            sim = sim.view(nq * nc, 1, l)
            st = self._conv(sim, name + "_st_predictor.weight").view(nq, nc, l)
            ed = self._conv(sim, name + "_ed_predictor.weight").view(nq, nc, l)
            ctx_mask = ctx_mask.unsqueeze(0)
        else:
            sim = torch.einsum("bd,bld->bl", q, ctx_feat2).unsqueeze(1)
            st = self._conv(sim, name + "_st_predictor.weight").squeeze(1)
            ed = self._conv(sim, name + "_ed_predictor.weight").squeeze(1)
        return mask_logits(st, ctx_mask), mask_logits(ed, ctx_mask)

    def get_pred_from_raw_query(self, query_feat, query_mask, video_feat1, video_feat2, video_mask,
                                sub_feat1, sub_feat2, sub_mask, cross=False):
        """xml/model_xml.py:553-586.  Returns (q2ctx, st_logits, ed_logits), un-normalised, masked."""
        vq, sq = self.encode_query(query_feat, query_mask)
        return self.get_pred_from_modular_query(vq, sq, video_feat1, video_feat2, video_mask,
                                                sub_feat1, sub_feat2, sub_mask, cross)

    def get_pred_from_modular_query(self, vq, sq, video_feat1, video_feat2, video_mask,
                                    sub_feat1, sub_feat2, sub_mask, cross=False):
        divisor = int(self.use_sub) + int(self.use_video)
        v_s = self.get_video_level_scores(vq, video_feat1, video_mask) if self.use_video else 0
        s_s = self.get_video_level_scores(sq, sub_feat1, sub_mask) if self.use_sub else 0
        q2ctx = (v_s + s_s) / divisor
        if self.cfg["merge_two_stream"] and self.use_video and self.use_sub:
            st, ed = self.get_merged_st_ed_prob(vq, video_feat2, sq, sub_feat2, video_mask, cross)
        else:
            vst, ved = self.get_st_ed_prob(vq, video_feat2, video_mask, "video", cross) if self.use_video else (0, 0)
            sst, sed = self.get_st_ed_prob(sq, sub_feat2, sub_mask, "sub", cross) if self.use_sub else (0, 0)
            st = (vst + sst) / divisor
            ed = (ved + sed) / divisor
        return q2ctx, st, ed

    # ---- training forward ----------------------------------------------------------------
    def video_level_loss(self, scores, neg_ctx_rank, neg_q_rank):
        """xml/model_xml.py:588-637 with the two torch.randint draws (:622) injected as rank indices."""
        n = len(scores)
        ar = torch.arange(n)
        pos = scores[ar, ar]
        masked = scores.detach().clone()
        masked[ar, ar] = 999

        def neg(sc, sc_masked, ranks):
            _, order = torch.sort(sc_masked, descending=True, dim=1)
            idx = order[ar, _t(ranks).long()]
            return sc[ar, idx]

        neg_ctx = neg(scores, masked, neg_ctx_rank)
        neg_q = neg(scores.transpose(0, 1), masked.transpose(0, 1), neg_q_rank)

        def rank_loss(p, ng):
            if self.cfg.get("ranking_loss_type", "hinge") == "hinge":
                return torch.clamp(self.cfg["margin"] + ng - p, min=0).sum() / len(p)
            return torch.log1p(torch.exp(ng - p)).sum() / len(p)

        return rank_loss(pos, neg_ctx), rank_loss(pos, neg_q)

    def forward_loss(self, query_feat, query_mask, video_feat, video_mask, sub_feat, sub_mask,
                     st_ed_indices, neg_ctx_rank, neg_q_rank):
        """XML.forward, xml/model_xml.py:212-251 (eval-mode numerics: dropout off)."""
        v1, v2, s1, s2 = self.encode_context(video_feat, video_mask, sub_feat, sub_mask)
        q2c, st, ed = self.get_pred_from_raw_query(query_feat, query_mask, v1, v2, video_mask,
                                                   s1, s2, sub_mask, cross=False)
        st_ed_indices = _t(st_ed_indices).long()
        loss_st_ed = 0.0
        if self.cfg["lw_st_ed"] != 0:
            loss_st_ed = F.cross_entropy(st, st_ed_indices[:, 0]) + F.cross_entropy(ed, st_ed_indices[:, 1])
        l_ctx, l_q = self.video_level_loss(q2c, neg_ctx_rank, neg_q_rank)
        loss_st_ed = self.cfg["lw_st_ed"] * loss_st_ed
        l_ctx = self.cfg["lw_neg_ctx"] * l_ctx
        l_q = self.cfg["lw_neg_q"] * l_q
        val = lambda x: float(x.detach()) if torch.is_tensor(x) else float(x)      # noqa: E731
        return loss_st_ed + l_ctx + l_q, dict(loss_st_ed=val(loss_st_ed), loss_neg_ctx=val(l_ctx), loss_neg_q=val(l_q))


# ------------------------------------------------------------------------------------------
# ranking tail (driver code of the reference, xml/inference.py)
# ------------------------------------------------------------------------------------------
def min_max_length_mask(l, min_l, max_l):
    """generate_min_max_length_mask, xml/inference.py:170-192: 1 where min_l <= ed-st < max_l."""
    ones = np.ones((l, l), dtype=np.float32)
    return np.triu(ones, k=min_l) * (1 - np.triu(ones, k=max_l))


def cat_pad_context(tensor_list):
    """cat_tensor, xml/inference.py:71-87: zero-pad each context batch to the global max L, concat."""
    if len(tensor_list) == 0:
        return None
    max_l = max(t.shape[1] for t in tensor_list)
    n = sum(t.shape[0] for t in tensor_list)
    out = tensor_list[0].new_zeros((n, max_l) + tuple(tensor_list[0].shape[2:]))
    r = 0
    for t in tensor_list:
        out[r:r + t.shape[0], :t.shape[1]] = t
        r += t.shape[0]
    return out


def vcmr_tail(q2c, st_logits, ed_logits, q2c_alpha=20.0, max_vcmr_video=100, min_pred_l=2, max_pred_l=16,
              max_before_nms=200, external_top=None):
    """Torch part of compute_query2ctx_info for one query batch, xml/inference.py:317-386.

    q2c (Nq,Nv); st/ed logits (Nq,Nv,L).  Returns dict with the arrays the reference hands to numpy:
    top video scores/indices (Nq,K) and the first `max_before_nms` entries of the flat descending sort
    over (K, L, L).  Multiplication order follows torch.einsum without opt_einsum: (st*w)*ed.
    external_top = (meta indices (Nq,K) int64, scores (Nq,K) f32): the external-VR branch, xml/inference.py:349-355 --
    another model's videos and cosine-like scores replace topk(exp(alpha * q2c)); the weights are exp(alpha * score).
    """
    w = torch.exp(q2c_alpha * q2c)
    st = torch.softmax(st_logits, dim=-1)
    ed = torch.softmax(ed_logits, dim=-1)
    k = min(max_vcmr_video, w.shape[1])
    if external_top is None:
        top_w, top_i = torch.topk(w, k, dim=1, largest=True)
    else:
        top_i = torch.as_tensor(external_top[0]).long()
        top_w = torch.exp(q2c_alpha * torch.as_tensor(external_top[1]).float())
    rows = torch.arange(len(st)).unsqueeze(1)
    st_k = st[rows, top_i]
    ed_k = ed[rows, top_i]
    prod = torch.einsum("qvm,qv,qvn->qvmn", st_k, top_w, ed_k)
    l = prod.shape[-1]
    prod = prod * torch.from_numpy(min_max_length_mask(l, min_pred_l, max_pred_l))
    flat = prod.reshape(len(prod), -1)
    s_sorted, i_sorted = torch.sort(flat, dim=1, descending=True)
    return dict(top_scores=top_w, top_indices=top_i, st_probs=st, ed_probs=ed,
                flat_scores=s_sorted[:, :max_before_nms], flat_indices=i_sorted[:, :max_before_nms],
                ctx_l=l)


def unravel_moments(flat_indices, top_indices, ctx_l, clip_length=1.5):
    """numpy tail, xml/inference.py:423-431.  Returns (video_meta_idx, st_sec, ed_sec)."""
    flat_indices = np.asarray(flat_indices)
    top_indices = np.asarray(top_indices)
    local, st_i, ed_i = np.unravel_index(flat_indices, (top_indices.shape[1], ctx_l, ctx_l))
    vid = np.take_along_axis(top_indices, local, axis=1)
    st_s = st_i.astype(np.float32) * clip_length
    ed_s = ed_i.astype(np.float32) * clip_length + clip_length
    return vid, st_s, ed_s


def svmr_tail(st_probs, ed_probs, min_pred_l=2, max_pred_l=16, max_before_nms=200):
    """get_svmr_res_from_st_ed_probs + top_n_array_2d, xml/inference.py:195-241,
    utils/tensor_utils.py:115-141.  st/ed probs (Nq, L) numpy.  Returns (Nq, n, 3) [st_idx, ed_idx, score]
    (indices before the +1 / clip_length scaling)."""
    st_probs = np.asarray(st_probs, dtype=np.float32)
    ed_probs = np.asarray(ed_probs, dtype=np.float32)
    prod = np.einsum("bm,bn->bmn", st_probs, ed_probs)
    prod = prod * min_max_length_mask(prod.shape[-1], min_pred_l, max_pred_l)[None]
    out = []
    for e in prod:
        order = np.argsort(e, axis=None)
        r, c = np.unravel_index(order, e.shape)
        r = r[::-1][:max_before_nms]
        c = c[::-1][:max_before_nms]
        out.append(np.stack([r, c, e[r, c]], axis=1))
    return np.stack(out, axis=0)


# ------------------------------------------------------------------------------------------
# post-processing ("next" rows)
# ------------------------------------------------------------------------------------------
def temporal_iou(a, b):
    """compute_temporal_iou, utils/temporal_nms.py:5-22 (union = hull, as in the reference)."""
    inter = max(0, min(a[1], b[1]) - max(a[0], b[0]))
    union = max(a[1], b[1]) - min(a[0], b[0])
    return 0 if union == 0 else 1.0 * inter / union


def temporal_nms(predictions, nms_threshold, max_after_nms=100):
    """temporal_non_maximum_suppression, utils/temporal_nms.py:25-74.
    predictions: list of [st, ed, score]; greedy, keeps the best and drops overlaps > threshold."""
    if len(predictions) == 1:
        return predictions
    rest = sorted(predictions, key=lambda x: x[2], reverse=True)
    kept = []
    while len(rest) > 1 and len(kept) < max_after_nms:
        head = rest[0]
        rest = [head] + [p for p in rest[1:] if not temporal_iou(head[:2], p[:2]) > nms_threshold]
        kept.append(rest.pop(0))
    if len(kept) < max_after_nms and len(rest) >= 1:
        kept.append(rest.pop(0))
    return [[p[0], p[1], p[2]] for p in kept]


def vcmr_nms(all_video_predictions, nms_threshold, max_before_nms=1000, max_after_nms=100):
    """filter_vcmr_by_nms, baselines/clip_alignment_with_language/inference.py:189-225."""
    groups = {}
    for p in all_video_predictions[:max_before_nms]:
        groups.setdefault(p[0], []).append(list(p[1:]))
    merged = []
    for vid, preds in groups.items():
        for p in temporal_nms(preds, nms_threshold):
            merged.append([vid] + list(p))
    return sorted(merged, key=lambda x: x[3], reverse=True)[:max_after_nms]


# ------------------------------------------------------------------------------------------
# optimizer (xml/optimization.py)
# ------------------------------------------------------------------------------------------
def warmup_linear(progress, warmup):
    """WarmupLinearSchedule.get_lr_, xml/optimization.py:166-169."""
    if progress < warmup:
        return progress / warmup
    return max((progress - 1.0) / (warmup - 1.0), 0.0)


def bert_adam_step(params, grads, state, step, weight_decay, lr=1e-4, warmup=-1, t_total=-1, b1=0.9, b2=0.999, e=1e-6,
                   max_grad_norm=1.0, **_):
    """BertAdam.step for the warmup_linear schedule, xml/optimization.py:273-338.  params / grads / weight_decay:
    dicts keyed by parameter name; state: dict name -> (m, v), created on first use.  Updates params and grads
    (the per-tensor clip rescales the gradient in place) and returns nothing; `step` is the per-parameter step
    counter BEFORE this call (state['step'], :325-330): one int for all parameters, or a dict name -> int.  Only the
    tensors listed in `params` are touched (`if p.grad is None: continue`, :289-291)."""
    for k, p in params.items():
        sk = step[k] if isinstance(step, dict) else step
        mult = 1.0 if t_total < 0 else warmup_linear(float(sk) / float(t_total), max(warmup, 0.0))
        g = grads[k]
        if k not in state:
            state[k] = (torch.zeros_like(p), torch.zeros_like(p))
        m, v = state[k]
        if max_grad_norm > 0:
            coef = max_grad_norm / (float(g.norm(2)) + 1e-6)     # clip_grad_norm_ on one tensor
            if coef < 1:
                g.mul_(coef)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        update = m / (v.sqrt() + e)
        if weight_decay[k] > 0.0:
            update = update + weight_decay[k] * p
        p.sub_(lr * mult * update)
