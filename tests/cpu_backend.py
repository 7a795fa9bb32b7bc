"""TEST INFRASTRUCTURE: CPU stand-ins (built on the oracle's formulation) for tvretrieval_amd.ops and for the model,
so that the host-side orchestration (corpus sharding, two-phase all-gather merge) can be exercised with world_size-2
`gloo` process groups on a machine without a GPU.  Never imported by the product."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import xml_oracle as O


class CpuOps(object):
    @staticmethod
    def l2norm_rows(x):
        return F.normalize(x.float(), dim=-1)

    @staticmethod
    def q2c_scores(qn, cn, mask, out=None, combine=False):
        s = torch.einsum("md,nld->mln", qn.double(), cn.double()).float()
        s = torch.max(O.mask_logits(s, mask.t().unsqueeze(0)), dim=1)[0]
        if out is None:
            return s
        out.copy_((out + s) * 0.5 if combine else s)
        return out

    @staticmethod
    def q2c_scores_fused(qn, cn, masks, out=None):
        acc = None
        for m in range(len(qn)):
            s = CpuOps.q2c_scores(qn[m], cn[m], masks[m])
            acc = s if acc is None else (acc + s) * 0.5
        return acc

    @staticmethod
    def topk_rows(scores, k, alpha=0.0, idx_in=None):
        n = scores.shape[1]
        pay = idx_in.long() if idx_in is not None else torch.arange(n).repeat(scores.shape[0], 1)
        order = torch.argsort(pay, dim=1, stable=True)
        s1 = torch.gather(scores, 1, order)
        o2 = torch.argsort(s1, dim=1, descending=True, stable=True)[:, :k]
        vals = torch.gather(s1, 1, o2)
        idx = torch.gather(torch.gather(pay, 1, order), 1, o2)
        if alpha != 0.0:
            vals = torch.exp(alpha * vals)
        return vals, idx.int()

    # ---- exact-rank mode (round 3's pipeline: bf16 filter values held in f32 tensors, exact re-score) -----------------
    @staticmethod
    def round_bf16_rows_err(y):
        yb = y.float().to(torch.bfloat16).float()
        return yb, (y.double() - yb.double()).norm(dim=-1).float()

    @staticmethod
    def pack_q2c_corpus(feat1n, mask=None, plan=None, normalize=False):
        return F.normalize(feat1n.float(), dim=-1) if normalize else feat1n

    @staticmethod
    def q2c_rescore(qn, cn, masks, pair_vid):
        nq, kp = pair_vid.shape
        nv = cn[0].shape[0]
        out = torch.full((nq, kp), float("-inf"))
        pv = pair_vid.long()
        ok = (pv >= 0) & (pv < nv)
        for q in range(nq):
            v = pv[q][ok[q]]
            if v.numel() == 0:
                continue
            tot = 0
            for m in range(len(qn)):
                s = torch.einsum("d,nld->nl", qn[m][q].double(), cn[m][v].double()).float()
                tot = tot + O.mask_logits(s, masks[m][v]).max(1)[0]
            out[q, ok[q]] = tot / len(qn)
        return out

    @staticmethod
    def exact_certificate(filter_scores, top_val, eq, ec, slack, alpha, outside):
        c = 1 + 1e-6
        eps = sum(e * c + (c + e) * x for e, x in zip(eq, ec)) / len(eq) + slack
        t_k = top_val[:, -1].clone()
        fail = ((~(filter_scores[:, -1] + eps < t_k)) & bool(outside)).int()
        if alpha != 0.0:
            top_val.copy_(torch.exp(alpha * top_val))
        return fail, eps, t_k - eps, fail.sum().reshape(1).int()

    @staticmethod
    def select_ge_rows(scores, thr, cap=None):
        take = scores >= thr[:, None]
        cnt = take.sum(1).int()
        if cap is None:
            return cnt
        idx = torch.full((scores.shape[0], int(cap)), -1, dtype=torch.int32)
        for r in range(scores.shape[0]):
            cols = torch.nonzero(take[r]).reshape(-1)[:int(cap)]
            idx[r, :cols.numel()] = cols.int()
        return idx, cnt

    @staticmethod
    def convse_rerank(q_lin, feat2, masks, pair_vid, conv_w, l_ref, merged, ksize, softmax=True, zero_skipped=True):
        n_mod = len(q_lin)
        n_conv = 1 if merged else n_mod
        wst = conv_w[:n_conv * ksize].view(n_conv, 1, 1, ksize)
        wed = conv_w[n_conv * ksize:].view(n_conv, 1, 1, ksize)
        nq, kp = pair_vid.shape
        lpad = feat2[0].shape[1]
        st = torch.zeros(nq, kp, lpad)
        ed = torch.zeros(nq, kp, lpad)
        pv = pair_vid.long()
        for q in range(nq):
            ok = pv[q] >= 0
            if not ok.any():
                continue
            v = pv[q][ok]
            sims = [torch.einsum("d,nld->nl", q_lin[m][q].double(), feat2[m][v, :l_ref].double()).float()
                    for m in range(n_mod)]
            conv = lambda x, w: F.conv1d(x.unsqueeze(1), w, padding=ksize // 2).squeeze(1)
            if merged:
                x = (sims[0] + sims[1]) / 2
                a = O.mask_logits(conv(x, wst[0]), masks[0][v, :l_ref])
                b = O.mask_logits(conv(x, wed[0]), masks[0][v, :l_ref])
            else:
                a = sum(O.mask_logits(conv(sims[m], wst[m]), masks[m][v, :l_ref]) for m in range(n_mod)) / n_mod
                b = sum(O.mask_logits(conv(sims[m], wed[m]), masks[m][v, :l_ref]) for m in range(n_mod)) / n_mod
            if softmax:
                a, b = torch.softmax(a, -1), torch.softmax(b, -1)
            st[q, ok, :l_ref] = a
            ed[q, ok, :l_ref] = b
        return st, ed

    @staticmethod
    def moment_topk(st, ed, w, l_ref, min_l, max_l, n_out):
        nq, kp, _ = st.shape
        st, ed = st[..., :l_ref], ed[..., :l_ref]
        if w is None:
            w = torch.ones(nq, kp)
        prod = torch.einsum("qvm,qv,qvn->qvmn", st, w, ed) * torch.from_numpy(O.min_max_length_mask(l_ref, min_l, max_l))
        flat = prod.reshape(nq, -1)
        s, i = torch.sort(flat, dim=1, descending=True, stable=True)
        s, i = s[:, :n_out], i[:, :n_out].int()
        if s.shape[1] < n_out:
            pad = n_out - s.shape[1]
            s = torch.cat([s, torch.zeros(nq, pad)], 1)
            i = torch.cat([i, torch.full((nq, pad), -1, dtype=torch.int32)], 1)
        i = torch.where(s > 0, i, torch.full_like(i, -1))
        return s, i


class CpuModel(object):
    """The slice of the XML host interface that the drivers touch, answered by the oracle."""

    def __init__(self, cfg, sd):
        from tvretrieval_amd.easydict_compat import EasyDict
        self.o = O.OracleXML(cfg, sd)
        self.config = EasyDict(cfg)
        self.use_video, self.use_sub = self.o.use_video, self.o.use_sub
        w = self.o.w
        for m in ("video", "sub"):
            if w.has(m + "_query_linear.weight"):
                setattr(self, m + "_query_linear",
                        (lambda mm: (lambda x: F.linear(x, w[mm + "_query_linear.weight"], w[mm + "_query_linear.bias"])))(m))

    def encode_query(self, qf, qm):
        with torch.no_grad():
            return self.o.encode_query(qf, qm)

    def encode_context(self, vf, vm, sf, sm):
        with torch.no_grad():
            return self.o.encode_context(vf, vm, sf, sm)

    def _conv_weights(self):
        w = self.o.w
        if self.config.merge_two_stream and self.use_video and self.use_sub:
            names_st, names_ed = ["merged_st_predictor"], ["merged_ed_predictor"]
        else:
            mods = [n for n, u in (("video", self.use_video), ("sub", self.use_sub)) if u]
            names_st = [n + "_st_predictor" for n in mods]
            names_ed = [n + "_ed_predictor" for n in mods]
        return torch.cat([w[n + ".weight"].reshape(-1) for n in names_st + names_ed])


class _WallEvent(object):
    """Stand-in for a HIP event on the CPU backend: a perf_counter stamp."""

    def __init__(self):
        self.t = None

    def record(self):
        import time
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class BenchBackend(object):
    """tests/bench_cpu_entry.py: the launcher / sharded orchestration of bench.py on gloo ranks without a
    GPU (tests/test_bench_launcher.py).  Kernels = CpuOps, model = CpuModel over a freshly initialised XML."""
    name, dist_backend = "cpu-stub", "gloo"

    def __init__(self, local_rank):
        self.device = torch.device("cpu")
        self.ops = CpuOps
        torch.set_num_threads(2)

    def make_model(self, cfg, dtype):
        from tvretrieval_amd.model_xml import XML
        sd = {k: v.detach().float() for k, v in XML(cfg).state_dict().items()}
        return CpuModel(cfg, sd)

    def sync(self):
        pass

    def event(self):
        return _WallEvent()

    def init_kwargs(self):
        return {}
