"""XML (Cross-modal Moment Localization) -- host-side mirror of the reference model class whose compute runs in
hand-written HIP kernels (libxmlhip.so, include/xmlhip.h).

Mirrors `XML(nn.Module)` of the reference (baselines/crossmodal_moment_localization/model_xml.py:52-641,
"xml/" below): same constructor (`XML(config)` with the keys of xml_base_config, xml/model_xml.py:19-49), same
method names and argument meaning, same `state_dict()` key names and shapes and the same checkpoint dict
(`{"model", "model_cfg", "epoch"}`, xml/train.py:219-223), so reference checkpoints load unchanged.

What is different by design: the nn.Module tree below only HOLDS parameters (fp32 masters, as in the checkpoint).
No torch op computes anything on the hot path -- each method packs the weights once per (dtype, parameter
version) and dispatches to the C ABI through tvretrieval_amd.ops.  There is no CPU fallback: tensors must be on
the GPU and libxmlhip.so must load.

Scope: transformer encoder + conv span predictor (the reference defaults).  The cnn / gru /
lstm encoder ablations, `cat_linear` predictor, stacked conv predictors and `no_modular` are not implemented and
raise at construction.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .easydict_compat import EasyDict as edict

xml_base_config = edict(
    merge_two_stream=True, cross_att=True, span_predictor_type="conv", encoder_type="transformer",
    add_pe_rnn=False, visual_input_size=2048, query_input_size=768, sub_input_size=768, hidden_size=500,
    conv_kernel_size=5, stack_conv_predictor_conv_kernel_sizes=-1, conv_stride=1, max_ctx_l=100, max_desc_l=30,
    input_drop=0.1, drop=0.1, n_heads=4, ctx_mode="video_sub", margin=0.1, ranking_loss_type="hinge",
    lw_neg_q=1, lw_neg_ctx=1, lw_st_ed=1, use_hard_negative=False, hard_pool_size=20, use_self_attention=True,
    no_modular=False, pe_type="none", initializer_range=0.02,
)


def _round_up(x, m):
    return (x + m - 1) // m * m


class _PackedMixin(object):
    """Caches device-dtype copies of a holder's parameters; re-packs when any parameter changed in place
    (optimizer step, load_state_dict) or moved (.to())."""

    _generation = 0     # bumped by optimizers that update parameters through raw kernels (no _version change)

    @staticmethod
    def bump_generation():
        _PackedMixin._generation += 1

    def _pack_key(self, dtype):
        return (dtype, _PackedMixin._generation) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _act(self, x):
        """x in the activation dtype of the model's compute dtype (XML.set_compute_dtype marks every holder; a holder used on
        its own computes in its input's dtype).  For the direct sub-module forwards: a caller may hand f32 tensors to a bf16
        model like it hands them to the reference's modules."""
        want = ops.act_dtype(getattr(self, "_compute_dtype", None) or x.dtype)
        x = x.contiguous()
        return x if x.dtype == want else ops.convert(x, want)

    def packed(self, dtype):
        # XML.set_compute_dtype(ops.F16S) marks every holder: f32 activations go with split-f16 weights
        if dtype == torch.float32 and getattr(self, "_split16", False):
            dtype = ops.F16S
        key = self._pack_key(dtype)
        if getattr(self, "_pk_key", None) != key:
            with torch.no_grad():
                self._pk = self._build_packed(dtype)
            self._pk_key = key
        return self._pk


def _w(t, dtype):
    t = t.detach().contiguous()
    return t if dtype == torch.float32 else ops.pack_weights(t.float().contiguous(), dtype)


def _f(t):
    return t.detach().float().contiguous()


class LinearLayer(nn.Module, _PackedMixin):
    """Parameter holder with the reference's key names (LayerNorm.*, net.1.*), xml/model_components.py:141-163."""

    def __init__(self, in_hsz, out_hsz, dropout=0.1):
        super().__init__()
        self.LayerNorm = nn.LayerNorm(in_hsz)
        self.net = nn.Sequential(nn.Dropout(dropout), nn.Linear(in_hsz, out_hsz))

    def _build_packed(self, dtype):
        lin = self.net[1]
        w = lin.weight.detach()
        d_in = w.shape[1]
        d_pad = _round_up(d_in, 8)
        if d_pad != d_in:      # TEF inputs (3074 / 770, xml/config.py:251-254): zero K columns, the kernel pads LN(x) alike
            w = torch.nn.functional.pad(w, (0, d_pad - d_in))
        return dict(ln_g=_f(self.LayerNorm.weight), ln_b=_f(self.LayerNorm.bias), w=_w(w, dtype), b=_f(lin.bias))

    def forward(self, x):
        """ReLU(Linear(LayerNorm(x))) (xml/model_components.py:156-163; dropout is the identity in eval mode) on the HIP
        entries xml_add_layernorm + xml_linear.  x (N, L, D_in) f32 (raw features) or the compute dtype; the result is in
        the model's activation dtype.  XML.encode_input does not come through here: it runs this layer fused with the
        positional encoding (xml_linear_ln_relu_pos)."""
        dtype = ops.act_dtype(getattr(self, "_compute_dtype", None) or x.dtype)
        p = self.packed(dtype)
        d_in, d_pad = x.shape[-1], p["w"].shape[1]
        h = ops.add_layernorm(x.contiguous(), None, p["ln_g"], p["ln_b"], out_dtype=dtype)
        if d_pad != d_in:          # TEF widths: the weight's zero columns meet zero activations
            h = torch.nn.functional.pad(h, (0, d_pad - d_in))
        return ops.linear(h.contiguous(), p["w"], p["b"], relu=True)


class TrainablePositionalEncoding(nn.Module, _PackedMixin):
    """Holder: position_embeddings.weight, LayerNorm.*  (xml/model_components.py:67-89)."""

    def __init__(self, max_position_embeddings, hidden_size, dropout=0.1):
        super().__init__()
        self.position_embeddings = nn.Embedding(max_position_embeddings, hidden_size)
        self.LayerNorm = nn.LayerNorm(hidden_size)
        self.dropout = nn.Dropout(dropout)

    def _build_packed(self, dtype):
        # (the table is an ADDEND of the projection's epilogue: activation storage dtype, f32 under ops.F16S)
        return dict(pos=_w(self.position_embeddings.weight, ops.act_dtype(dtype)), ln_g=_f(self.LayerNorm.weight),
                    ln_b=_f(self.LayerNorm.bias))

    def forward(self, input_feat):
        """LayerNorm(input_feat + E[0:L]) (xml/model_components.py:76-89) on xml_add_layernorm; input_feat (N, L, D)."""
        input_feat = self._act(input_feat)
        p = self.packed(input_feat.dtype)
        n, l = input_feat.shape[:2]
        pos = p["pos"][:l].unsqueeze(0).expand(n, -1, -1).contiguous()
        return ops.add_layernorm(input_feat, pos, p["ln_g"], p["ln_b"])


class BertSelfAttention(nn.Module, _PackedMixin):
    """Holder: query/key/value Linear (xml/model_components.py:244-259).  Packed as stacked [q;k;v] and [k;v]."""

    def __init__(self, hidden_size, num_attention_heads, dropout=0.1):
        super().__init__()
        if hidden_size % num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (hidden_size, num_attention_heads))
        self.num_attention_heads = num_attention_heads
        self.query = nn.Linear(hidden_size, hidden_size)
        self.key = nn.Linear(hidden_size, hidden_size)
        self.value = nn.Linear(hidden_size, hidden_size)
        self.dropout = nn.Dropout(dropout)

    def _build_packed(self, dtype):
        q, k, v = self.query, self.key, self.value
        return dict(
            wqkv=_w(torch.cat([q.weight, k.weight, v.weight], 0), dtype),
            bqkv=_f(torch.cat([q.bias, k.bias, v.bias], 0)),
            wq=_w(q.weight, dtype), bq=_f(q.bias),
            wkv=_w(torch.cat([k.weight, v.weight], 0), dtype), bkv=_f(torch.cat([k.bias, v.bias], 0)),
            wk=_w(k.weight, dtype), bk=_f(k.bias), wv=_w(v.weight, dtype), bv=_f(v.bias))

    def forward(self, query_states, key_states, value_states, attention_mask):
        """xml/model_components.py:266-303: three projections (xml_linear) + the attention core (xml_attention_core) ->
        context layer (N, Lq, D).  attention_mask (N, Lq, L) or (N, 1, L) float, 1 = attend.  The kernels take the mask as an
        outer product q_mask x k_mask -- the two forms the reference ever builds (key mask broadcast over the queries;
        cross attention's einsum("bm,bn->bmn"), xml/model_xml.py:357-359); any other mask is rejected."""
        query_states, key_states, value_states = (self._act(t) for t in (query_states, key_states, value_states))
        p = self.packed(query_states.dtype)
        m = attention_mask.float()
        if m.dim() == 2:
            m = m.unsqueeze(1)
        k_mask = m.amax(dim=1).contiguous()                            # (N, L)
        q_mask = None
        if m.shape[1] != 1:
            q_mask = m.amax(dim=2).contiguous()                        # (N, Lq)
            if not torch.equal(q_mask.unsqueeze(2) * k_mask.unsqueeze(1), m):
                raise ValueError("BertSelfAttention: attention_mask must be an outer product of a query mask and a key mask")
        q = ops.linear(query_states.contiguous(), p["wq"], p["bq"])
        k = ops.linear(key_states.contiguous(), p["wk"], p["bk"])
        v = ops.linear(value_states.contiguous(), p["wv"], p["bv"])
        return ops.attention_core(q, k, v, q_mask, k_mask, self.num_attention_heads)


class BertSelfOutput(nn.Module, _PackedMixin):
    def __init__(self, hidden_size, dropout=0.1):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.LayerNorm = nn.LayerNorm(hidden_size)
        self.dropout = nn.Dropout(dropout)

    def _build_packed(self, dtype):
        return dict(wo=_w(self.dense.weight, dtype), bo=_f(self.dense.bias), ln_g=_f(self.LayerNorm.weight),
                    ln_b=_f(self.LayerNorm.bias))

    def forward(self, hidden_states, input_tensor):
        """LayerNorm(dense(hidden_states) + input_tensor) (xml/model_components.py:313-317) on xml_linear + xml_add_layernorm."""
        hidden_states, input_tensor = self._act(hidden_states), self._act(input_tensor)
        p = self.packed(hidden_states.dtype)
        h = ops.linear(hidden_states, p["wo"], p["bo"])
        return ops.add_layernorm(h, input_tensor, p["ln_g"], p["ln_b"])


class BertAttention(nn.Module):
    """Holder: self.{query,key,value}, output.{dense,LayerNorm} (xml/model_components.py:201-216)."""

    def __init__(self, hidden_size, num_attention_heads, dropout=0.1):
        super().__init__()
        self.self = BertSelfAttention(hidden_size, num_attention_heads, dropout)
        self.output = BertSelfOutput(hidden_size, dropout)

    def forward(self, input_tensor, attention_mask, out=None):
        """input (N, L, H) compute dtype; attention_mask (N, 1, L) or (N, L) float 1=valid (key mask).  out: optional
        destination tensor (index build: the layer writes straight into the corpus index)."""
        if attention_mask.dim() == 3:
            attention_mask = attention_mask[:, 0]
        a = self.self.packed(input_tensor.dtype)
        o = self.output.packed(input_tensor.dtype)
        return ops.attention_block(input_tensor.contiguous(), attention_mask.float().contiguous(), a["wqkv"],
                                   a["bqkv"], o["wo"], o["bo"], o["ln_g"], o["ln_b"],
                                   self.self.num_attention_heads, out=out)


class _QueryLinear(nn.Linear, _PackedMixin):
    """video_query_linear / sub_query_linear: nn.Linear parameters, HIP forward."""

    def _build_packed(self, dtype):
        return dict(w=_w(self.weight, dtype), b=_f(self.bias))

    def forward(self, x):
        p = self.packed(x.dtype)
        return ops.linear(x.contiguous(), p["w"], p["b"])


class _SpanConv(nn.Conv1d):
    """{merged,video,sub}_{st,ed}_predictor: nn.Conv1d(1, 1, k, padding=k // 2, bias=False) parameters (same state_dict key,
    `weight` (1, 1, k)); forward on xml_conv1d_rows for callers that hold a similarity tensor (profile_main.py:204-205).  The
    retrieval pass applies the taps inside K7 and never calls this."""

    def forward(self, x):
        assert x.dim() == 3 and x.shape[1] == 1, "span predictor input is (N, 1, L)"
        if torch.is_grad_enabled() and (self.weight.requires_grad or x.requires_grad):
            # the kernel below has no backward: a training graph through this module would silently lose the gradient
            return torch.nn.functional.conv1d(x, self.weight, None, self.stride, self.padding)
        return ops.conv1d_rows(x.float().contiguous(), self.weight.detach().float().reshape(-1).contiguous())


PACK_QUERY_TOKENS = True      # encode_query on the packed valid tokens of large batches (tests / A-B runs: False)
PACK_MIN_ROWS = 16384         # below this many padded token rows the pass is launch-bound and the packing passes do not pay


class XML(nn.Module):
    def __init__(self, config, compute_dtype=torch.float32):
        super().__init__()
        if not isinstance(config, edict):
            config = edict(dict(config))
        self.config = config
        if config.get("encoder_type", "transformer") != "transformer":
            raise NotImplementedError("only encoder_type='transformer' is built (SURVEY.md section 2, #2)")
        if config.get("span_predictor_type", "conv") != "conv":
            raise NotImplementedError("only span_predictor_type='conv' is built")
        if config.get("stack_conv_predictor_conv_kernel_sizes", -1) != -1:
            raise NotImplementedError("stacked conv predictors are disabled at inference by the reference "
                                      "(xml/inference.py:538) and are not built")
        if config.get("no_modular", False):
            raise NotImplementedError("no_modular ablation is not built")
        if config.get("conv_stride", 1) != 1:
            raise NotImplementedError("conv_stride != 1 is not built")
        hsz, nh = config.hidden_size, config.n_heads
        # shapes the kernels implement (include/xmlhip.h) -- rejected here with the reason, not deep inside a launch
        if hsz % nh != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)" % (hsz, nh))
        if hsz % (32 * nh) != 0:
            raise ValueError("hidden_size=%d with n_heads=%d: the HIP attention kernels need a head size that is a multiple "
                             "of 32 (hidden_size %% %d == 0); the reference's training default is 256 (xml/config.py:143), "
                             "xml_base_config's 500 is a placeholder" % (hsz, nh, 32 * nh))
        for key in ("max_ctx_l", "max_desc_l"):
            if config[key] > 128:
                raise ValueError("%s=%d: sequences longer than 128 positions are not built (one attention tile stays "
                                 "on chip; TVR truncates to 100 clips / 30 tokens, xml/config.py:86-88)" % (key, config[key]))
        if config.conv_kernel_size % 2 != 1 or config.conv_kernel_size > 15:
            raise ValueError("conv_kernel_size=%d: the ConvSE kernels take odd sizes up to 15" % config.conv_kernel_size)
        self.query_pos_embed = TrainablePositionalEncoding(config.max_desc_l, hsz, config.input_drop)
        self.ctx_pos_embed = TrainablePositionalEncoding(config.max_ctx_l, hsz, config.input_drop)
        self.query_input_proj = LinearLayer(config.query_input_size, hsz, config.input_drop)
        self.query_encoder = BertAttention(hsz, nh, config.drop)

        def conv():
            k = config.conv_kernel_size
            return _SpanConv(1, 1, k, stride=1, padding=k // 2, bias=False)

        self.use_video = "video" in config.ctx_mode
        self.use_sub = "sub" in config.ctx_mode
        for name, use, in_size in (("video", self.use_video, config.visual_input_size),
                                   ("sub", self.use_sub, config.sub_input_size)):
            if not use:
                continue
            setattr(self, name + "_input_proj", LinearLayer(in_size, hsz, config.input_drop))
            setattr(self, name + "_encoder1", BertAttention(hsz, nh, config.drop))
            setattr(self, name + "_encoder2", BertAttention(hsz, nh, config.drop))
            if config.cross_att:
                setattr(self, name + "_cross_att", BertSelfAttention(hsz, nh, config.drop))
                setattr(self, name + "_cross_layernorm", nn.LayerNorm(hsz))
            else:
                setattr(self, name + "_encoder3", BertAttention(hsz, nh, config.drop))
            setattr(self, name + "_query_linear", _QueryLinear(hsz, hsz))
            if not config.merge_two_stream:
                setattr(self, name + "_st_predictor", conv())
                setattr(self, name + "_ed_predictor", conv())
        self.modular_vector_mapping = nn.Linear(hsz, int(self.use_sub) + int(self.use_video), bias=False)
        if config.merge_two_stream:
            self.merged_st_predictor = conv()
            self.merged_ed_predictor = conv()
        self.reset_parameters()
        self.set_compute_dtype(compute_dtype)

    # ---- parameter management ------------------------------------------------------------------------
    def reset_parameters(self):
        """Same initial distribution as the reference (xml/model_xml.py:185-201)."""
        def re_init(module):
            if isinstance(module, (nn.Linear, nn.Embedding)):
                module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
            elif isinstance(module, nn.LayerNorm):
                module.bias.data.zero_()
                module.weight.data.fill_(1.0)
            elif isinstance(module, nn.Conv1d):
                module.reset_parameters()
            if isinstance(module, nn.Linear) and module.bias is not None:
                module.bias.data.zero_()
        self.apply(re_init)

    def set_compute_dtype(self, dtype):
        """torch.float32 (exact-f32 MFMA: the parity configuration), torch.bfloat16, or ops.F16S: f32 activations with every
        projection on the 16-bit MFMA pipe as a split-f16 product (f32-grade results, include/xmlhip.h "Exact-rank mode on
        the 16-bit pipe") -- the exact-rank mode's model."""
        assert dtype in (torch.float32, torch.bfloat16) or dtype is ops.F16S
        if dtype is ops.F16S and self.config.hidden_size % 32:
            raise ValueError("compute_dtype=ops.F16S needs hidden_size %% 32 == 0 (split rows are stored per 32 elements)")
        self.compute_dtype = dtype
        for m in self.modules():
            if isinstance(m, _PackedMixin):
                m._split16 = dtype is ops.F16S
                m._compute_dtype = dtype
        return self

    @property
    def act_dtype(self):
        """storage dtype of the activations (and of the tensors the public methods return)"""
        return ops.act_dtype(self.compute_dtype)

    def set_hard_negative(self, use_hard_negative, hard_pool_size):
        self.config.use_hard_negative = use_hard_negative
        self.config.hard_pool_size = hard_pool_size

    def set_train_st_ed(self, lw_st_ed):
        self.config.lw_st_ed = lw_st_ed

    def _conv_weights(self):
        """flat f32 [st filters..., ed filters...] in modality order (video, sub) or merged."""
        if self.config.merge_two_stream and self.use_video and self.use_sub:
            st, ed = [self.merged_st_predictor], [self.merged_ed_predictor]
        else:
            names = [n for n, u in (("video", self.use_video), ("sub", self.use_sub)) if u]
            st = [getattr(self, n + "_st_predictor") for n in names]
            ed = [getattr(self, n + "_ed_predictor") for n in names]
        # (cached like the packed projection weights: re-built when a tap changed in place or moved -- a 50-query batch is
        # ~20 short kernels, and the concatenation was one of them plus two launch gaps)
        key = (_PackedMixin._generation,) + tuple((m.weight.data_ptr(), m.weight._version) for m in st + ed)
        hit = self.__dict__.get("_conv_w_cache")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, torch.cat([m.weight.detach().float().reshape(-1) for m in st + ed]).contiguous())
            self.__dict__["_conv_w_cache"] = hit
        return hit[1]

    def _ln_params(self, ln):
        return _f(ln.weight), _f(ln.bias)

    # ---- encoders ------------------------------------------------------------------------------------
    def encode_input(self, feat, mask, input_proj_layer, encoder_layer, pos_embed_layer, out=None):
        """xml/model_xml.py:377-392.  feat (N, L, D_in) f32 (or compute dtype), mask (N, L) float."""
        dt = self.compute_dtype
        p, e = input_proj_layer.packed(dt), pos_embed_layer.packed(dt)
        if feat.shape[1] > e["pos"].shape[0]:
            raise IndexError("sequence length %d exceeds the positional table (%d)" % (feat.shape[1], e["pos"].shape[0]))
        if feat.dtype not in (torch.float32, ops.act_dtype(dt)):
            feat = feat.float()
        x = ops.linear_ln_relu_pos(feat.contiguous(), p["ln_g"], p["ln_b"], p["w"], p["b"], e["pos"], e["ln_g"],
                                   e["ln_b"])
        return encoder_layer(x, mask) if out is None else encoder_layer(x, mask, out=out)

    def cross_context_encoder(self, main_context_feat, main_context_mask, side_context_feat, side_context_mask,
                              cross_att_layer, norm_layer, self_att_layer, out=None):
        """xml/model_xml.py:357-373."""
        c = cross_att_layer.packed(main_context_feat.dtype)
        g, b = self._ln_params(norm_layer)
        res = ops.cross_attention(main_context_feat, main_context_mask.float().contiguous(), side_context_feat,
                                  side_context_mask.float().contiguous(), c["wq"], c["bq"], c["wkv"], c["bkv"], g, b,
                                  cross_att_layer.num_attention_heads)
        return self_att_layer(res, main_context_mask) if out is None else self_att_layer(res, main_context_mask, out=out)

    def cross_encode_context(self, video_feat, video_mask, sub_feat, sub_mask, outs=(None, None, None, None)):
        """xml/model_xml.py:344-355."""
        ev = self.encode_input(video_feat, video_mask, self.video_input_proj, self.video_encoder1, self.ctx_pos_embed,
                               out=outs[0])
        es = self.encode_input(sub_feat, sub_mask, self.sub_input_proj, self.sub_encoder1, self.ctx_pos_embed, out=outs[2])
        xv = self.cross_context_encoder(ev, video_mask, es, sub_mask, self.video_cross_att,
                                        self.video_cross_layernorm, self.video_encoder2, out=outs[1])
        xs = self.cross_context_encoder(es, sub_mask, ev, video_mask, self.sub_cross_att,
                                        self.sub_cross_layernorm, self.sub_encoder2, out=outs[3])
        return ev, xv, es, xs

    def non_cross_encode_context(self, context_feat, context_mask, module_name="video", outs=(None, None)):
        """xml/model_xml.py:297-329: encoder1 -> feat1 ; encoder2 -> encoder3 -> feat2."""
        f1 = self.encode_input(context_feat, context_mask, getattr(self, module_name + "_input_proj"),
                               getattr(self, module_name + "_encoder1"), self.ctx_pos_embed, out=outs[0])
        f2 = getattr(self, module_name + "_encoder2")(f1, context_mask)
        enc3 = getattr(self, module_name + "_encoder3")
        f2 = enc3(f2, context_mask) if outs[1] is None else enc3(f2, context_mask, out=outs[1])
        return f1, f2

    def encode_context(self, video_feat, video_mask, sub_feat, sub_mask, outs=(None, None, None, None)):
        """xml/model_xml.py:331-342.  Unused modalities return None.  outs: optional destinations for (video feat1, video
        feat2, sub feat1, sub feat2) -- contiguous (N, L, H) tensors of the compute dtype, e.g. row ranges of a preallocated
        corpus index; the last layer of each branch then writes there directly."""
        if self.config.cross_att:
            assert self.use_video and self.use_sub
            return self.cross_encode_context(video_feat, video_mask, sub_feat, sub_mask, outs)
        v1 = v2 = s1 = s2 = None
        if self.use_video:
            v1, v2 = self.non_cross_encode_context(video_feat, video_mask, "video", outs[0:2])
        if self.use_sub:
            s1, s2 = self.non_cross_encode_context(sub_feat, sub_mask, "sub", outs[2:4])
        return v1, v2, s1, s2

    def get_modularized_queries(self, encoded_query, query_mask, return_modular_att=False):
        """xml/model_xml.py:399-423."""
        if return_modular_att:
            raise NotImplementedError("visualisation outputs are out of scope")
        wm = _f(self.modular_vector_mapping.weight)
        out = ops.modular_pool(encoded_query.contiguous(), query_mask.float().contiguous(), wm)
        return (out[0], out[1]) if out.shape[0] == 2 else (out[0], out[0])

    def encode_query(self, query_feat, query_mask, n_valid_tokens=None):
        """xml/model_xml.py:291-295.
        n_valid_tokens (host int, not a reference argument): query_mask.sum() when the caller built the masks on the host
        and every row is a non-empty prefix of ones -- the packed encoder then needs no read-back (ops.pack_plan)."""
        # (not while a HIP graph is being captured: the packing plan needs a host read-back and the packed launch shapes
        # depend on the number of valid tokens of THIS batch -- a graph would bake the warm-up batch's in)
        if PACK_QUERY_TOKENS and query_feat.is_cuda and query_feat.shape[0] * query_feat.shape[1] >= PACK_MIN_ROWS \
                and query_feat.shape[1] <= 32 and self.config.hidden_size <= 1024 \
                and not torch.cuda.is_current_stream_capturing():
            packed = self._encode_query_packed(query_feat, query_mask, n_valid_tokens)
            if packed is not None:
                return packed
        enc = self.encode_input(query_feat, query_mask, self.query_input_proj, self.query_encoder,
                                self.query_pos_embed)
        return self.get_modularized_queries(enc, query_mask)

    def _encode_query_packed(self, query_feat, query_mask, n_valid_tokens=None):
        """encode_query without the padding rows.  The reference pads every query to the batch maximum (30 tokens on TVR,
        17.5 valid on average) and runs the projections, the attention and the LayerNorms on all of them; here the valid
        tokens of the batch are packed back to back (include/xmlhip.h "PACKED variable-length sequences"): 42 % fewer rows
        through K1-K4 at the TVR length distribution.  Same values per query -- a padded key adds exp(-10000 + s - max) = +0
        to the attention softmax and the modular pooling gives padded tokens weight exp(-1e10 - max) = 0.
        Returns None (caller keeps the padded path) unless every mask row is a non-empty prefix of ones."""
        dt = self.compute_dtype
        n, lq, d_in = query_feat.shape
        p, e = self.query_input_proj.packed(dt), self.query_pos_embed.packed(dt)
        if lq > e["pos"].shape[0]:
            raise IndexError("sequence length %d exceeds the positional table (%d)" % (lq, e["pos"].shape[0]))
        if d_in % 8 or d_in > 4096:
            return None
        cu, src_row, rows = ops.pack_plan(query_mask.float().contiguous(), n_valid_tokens)   # (one 4-byte read-back without it)
        if rows < 0:
            return None
        feat = query_feat if query_feat.dtype in (torch.float32, ops.act_dtype(dt)) else query_feat.float()
        # K1+K2: token i of the packed batch reads row src_row[i] of the padded one, positional row src_row[i] % lq
        x = ops.linear_ln_relu_pos_packed(feat.reshape(n * lq, d_in).contiguous(), src_row, rows, lq, p["ln_g"], p["ln_b"],
                                          p["w"], p["b"], e["pos"], e["ln_g"], e["ln_b"])
        a, o = self.query_encoder.self.packed(dt), self.query_encoder.output.packed(dt)
        max_len = int(lq)
        x = ops.attention_block_varlen(x, cu, n, max_len, a["wqkv"], a["bqkv"], o["wo"], o["bo"], o["ln_g"], o["ln_b"],
                                       self.query_encoder.self.num_attention_heads)
        out = ops.modular_pool_varlen(x, cu, n, max_len, _f(self.modular_vector_mapping.weight))
        return (out[0], out[1]) if out.shape[0] == 2 else (out[0], out[0])

    # ---- scores --------------------------------------------------------------------------------------
    @staticmethod
    def pad_context(feat, lpad):
        """(N, L, H) -> (N, lpad, H), zero rows beyond L (device memory plumbing only)."""
        n, l = feat.shape[:2]
        if l == lpad:
            return feat.contiguous()
        out = feat.new_zeros((n, lpad) + tuple(feat.shape[2:]))
        out[:, :l] = feat
        return out

    def get_video_level_scores(self, modularied_query, context_feat1, context_mask):
        """xml/model_xml.py:436-453 -> (Nq, Nv) f32."""
        lpad = _round_up(context_feat1.shape[1], 16)
        qn = ops.l2norm_rows(modularied_query.contiguous())
        cn = ops.l2norm_rows(self.pad_context(context_feat1, lpad))
        return ops.q2c_scores(qn, cn, self.pad_context(context_mask.float(), lpad))

    def _span_logits(self, queries, feats2, masks, names, cross, softmax=False, pair_vid=None):
        """Shared K7 dispatch.  queries/feats2/masks: per-modality lists; returns (st, ed) (Nq, K, L) f32."""
        l_ref = feats2[0].shape[1]
        lpad = _round_up(l_ref, 16)
        merged = bool(self.config.merge_two_stream and self.use_video and self.use_sub and len(names) == 2)
        q_lin = [getattr(self, n + "_query_linear")(q.contiguous()) for n, q in zip(names, queries)]
        f2 = [self.pad_context(f, lpad) for f in feats2]
        mk = [self.pad_context(m.float(), lpad) for m in masks]
        nq, nv = q_lin[0].shape[0], f2[0].shape[0]
        if pair_vid is None:
            if cross:
                pair_vid = torch.arange(nv, dtype=torch.int32, device=q_lin[0].device).repeat(nq, 1).contiguous()
            else:
                assert nq == nv, "cross=False pairs query i with video i"
                pair_vid = torch.arange(nv, dtype=torch.int32, device=q_lin[0].device).unsqueeze(1).contiguous()
        if merged:
            conv_w = torch.cat([self.merged_st_predictor.weight.detach().float().reshape(-1),
                                self.merged_ed_predictor.weight.detach().float().reshape(-1)]).contiguous()
        else:
            conv_w = torch.cat([getattr(self, n + "_st_predictor").weight.detach().float().reshape(-1) for n in names] +
                               [getattr(self, n + "_ed_predictor").weight.detach().float().reshape(-1) for n in names]
                               ).contiguous()
        st, ed = ops.convse_rerank(q_lin, f2, mk, pair_vid, conv_w, l_ref, merged, self.config.conv_kernel_size,
                                   softmax=softmax)
        st, ed = st[..., :l_ref], ed[..., :l_ref]
        if not cross and pair_vid.shape[1] == 1:
            st, ed = st[:, 0], ed[:, 0]
        return st, ed

    def get_merged_st_ed_prob(self, video_query, video_feat, sub_query, sub_feat, context_mask, cross=False,
                              return_similaity=False):
        """xml/model_xml.py:455-502 (masked logits)."""
        assert self.use_video and self.use_sub and not return_similaity
        return self._span_logits([video_query, sub_query], [video_feat, sub_feat], [context_mask, context_mask],
                                 ["video", "sub"], cross)

    def get_st_ed_prob(self, modularied_query, context_feat2, context_mask, module_name="video", cross=False):
        """xml/model_xml.py:504-551 (masked logits, single stream)."""
        return self._span_logits([modularied_query], [context_feat2], [context_mask], [module_name], cross)

    def get_pred_from_raw_query(self, query_feat, query_mask, video_feat1, video_feat2, video_mask,
                                sub_feat1, sub_feat2, sub_mask, cross=False):
        """xml/model_xml.py:553-586 -> (q2ctx (Nq,Nv), st logits, ed logits); cross=True: (Nq,Nv,L)."""
        video_query, sub_query = self.encode_query(query_feat, query_mask)
        return self.get_pred_from_modular_query(video_query, sub_query, video_feat1, video_feat2, video_mask,
                                                sub_feat1, sub_feat2, sub_mask, cross)

    def get_pred_from_modular_query(self, video_query, sub_query, video_feat1, video_feat2, video_mask,
                                    sub_feat1, sub_feat2, sub_mask, cross=False):
        q2c = None
        for use, q, f1, m in ((self.use_video, video_query, video_feat1, video_mask),
                              (self.use_sub, sub_query, sub_feat1, sub_mask)):
            if not use:
                continue
            lpad = _round_up(f1.shape[1], 16)
            qn = ops.l2norm_rows(q.contiguous())
            cn = ops.l2norm_rows(self.pad_context(f1, lpad))
            mk = self.pad_context(m.float(), lpad)
            if q2c is None:
                q2c = ops.q2c_scores(qn, cn, mk)
            else:
                ops.q2c_scores(qn, cn, mk, out=q2c, combine=True)   # (video + sub) / 2
        names = [n for n, u in (("video", self.use_video), ("sub", self.use_sub)) if u]
        qs = dict(video=video_query, sub=sub_query)
        fs = dict(video=video_feat2, sub=sub_feat2)
        ms = dict(video=video_mask, sub=sub_mask)
        st, ed = self._span_logits([qs[n] for n in names], [fs[n] for n in names], [ms[n] for n in names], names,
                                   cross)
        return q2c, st, ed

    def forward(self, query_feat, query_mask, video_feat, video_mask, sub_feat, sub_mask, tef_feat, tef_mask,
                st_ed_indices, neg_ctx_rank=None, neg_q_rank=None):
        """XML.forward (xml/model_xml.py:212-251) -> (loss, loss dict).

        This is the graph of tvretrieval_amd.train: HIP forward kernels recorded as autograd nodes with hand-written
        HIP backward (`loss.backward()` fills the f32 `.grad` of every parameter); under torch.no_grad() the same
        kernels just compute the loss values.  Dropout is applied in `model.train()` mode only.  The
        in-batch negatives of get_neg_scores (xml/model_xml.py:608-624) are drawn with torch.randint on the CPU
        generator in the reference's order unless rank indices are injected."""
        from .train import xml_forward_train
        return xml_forward_train(self, query_feat, query_mask, video_feat, video_mask, sub_feat, sub_mask,
                                 st_ed_indices, neg_ctx_rank, neg_q_rank)

    def get_video_level_loss(self, query_context_scores, neg_ctx_rank=None, neg_q_rank=None):
        """xml/model_xml.py:588-637 on the (N, N) in-batch score matrix -> (loss_neg_ctx, loss_neg_q), unweighted
        (xml_rank_loss kernel; differentiable when the scores carry a graph)."""
        from .autograd import RankLossFn
        from .train import draw_negative_ranks
        cfg = self.config
        if cfg.ranking_loss_type not in ("hinge", "lse"):
            raise NotImplementedError("Only support 'hinge' and 'lse'")
        n = len(query_context_scores)
        if neg_ctx_rank is None or neg_q_rank is None:
            neg_ctx_rank, neg_q_rank = draw_negative_ranks(self, n)
        dev = query_context_scores.device
        to_dev = lambda r: torch.as_tensor(r).to(device=dev, dtype=torch.int32).contiguous()   # noqa: E731
        losses = RankLossFn.apply(query_context_scores.float().contiguous(), to_dev(neg_ctx_rank), to_dev(neg_q_rank),
                                  float(cfg.margin), cfg.ranking_loss_type == "lse")
        return losses[0], losses[1]


def mask_logits(target, mask):
    """Kept for API parity (xml/model_xml.py:640-641); the kernels apply it in their epilogues."""
    return target * mask + (1 - mask) * (-1e10)
