"""Transformer variant of the phoneme-to-articulation network on MI355X -- inference (forward / generate).

Mirrors reference ``phoneme_to_articulation/transformer/models.py``: ``ArtSpeechTransformer`` (:280-474) with the
standard post-norm encoder (:309-318) and the custom multi-channel decoder (``MultiChannelTransformerDecoderLayer``
:165-277 = A self-channel blocks + A*(A-1) channel-interaction blocks + A cross-attention blocks per layer, each a
``ChannelProcessingLayer`` :37-100).  Constructor signature, ``forward`` / ``generate`` contracts and ``state_dict``
keys are the reference's.  The quirks listed in SURVEY appendix A.7 are reproduced (shared LayerNorm for src and tgt,
residual on the projected query, ReLU pre-projections, causal memory mask in ``forward``, zeros at padded encoder
positions when run without grad = the reference's nested-tensor fast path).

MI355X design: all A*(A+1) channel blocks of a decoder layer run as GROUPED strided-batched fp32-MFMA GEMMs
(``as_gemm_f32`` with per-batch offset tables) on activations kept in (frame, channel, feature) layout, so the whole
layer is ~25 launches instead of the reference's ~2 000 small kernels; LayerNorm affines are folded into the q/k/v
projections (one affine-free x_hat per channel serves every block that reads it); attention probabilities are a
masked-softmax kernel between two grouped GEMMs.  Host orchestration is Python (as in the reference); every device
operation goes through the C ABI.  Training (backward) of this variant is not built yet.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from ... import _lib
from ..encoder_decoder.models import HEAD_HIDDEN, _build_views, _numel

FF_DIM = 2048  # nn.TransformerEncoderLayer's default dim_feedforward (the reference does not override it, :309-313)
_BLOCK_FIELDS = (("query.0.weight", "dd"), ("query.0.bias", "d"), ("key.0.weight", "dd"), ("key.0.bias", "d"),
                 ("value.0.weight", "dd"), ("value.0.bias", "d"), ("multihead_attn.in_proj_weight", "3dd"),
                 ("multihead_attn.in_proj_bias", "3d"), ("multihead_attn.out_proj.weight", "dd"),
                 ("multihead_attn.out_proj.bias", "d"), ("layer_norm.weight", "d"), ("layer_norm.bias", "d"))


def _up(n, m=64):
    return (n + m - 1) // m * m


class _Layout:
    """state_dict key -> (offset, shape) in one flat fp32 buffer; per-layer block tensors are stacked over the
    A*(A+1) channel blocks of the layer in the order: self blocks c, interaction blocks (c, j), input blocks c."""

    def __init__(self, V, A, d, L, nf, head_lay, head_dims):
        self.views, self.off = {}, 0
        self.stacks = []  # per decoder layer: {field: offset of the [NB][...] stack}
        NB = A * (A + 1)
        size = {"dd": (d, d), "d": (d,), "3dd": (3 * d, d), "3d": (3 * d,)}

        def take(key, shape):
            o = self.off
            self.views[key] = (o, tuple(shape))
            self.off += _up(_numel(shape))
            return o

        take("src_embedding.weight", (V, d))
        take("tgt_embedding.0.weight", (nf,)), take("tgt_embedding.0.bias", (nf,))
        take("tgt_embedding.1.weight", (d, nf)), take("tgt_embedding.1.bias", (d,))
        for l in range(L):
            e = f"encoder.layers.{l}."
            take(e + "self_attn.in_proj_weight", (3 * d, d)), take(e + "self_attn.in_proj_bias", (3 * d,))
            take(e + "self_attn.out_proj.weight", (d, d)), take(e + "self_attn.out_proj.bias", (d,))
            take(e + "linear1.weight", (FF_DIM, d)), take(e + "linear1.bias", (FF_DIM,))
            take(e + "linear2.weight", (d, FF_DIM)), take(e + "linear2.bias", (d,))
            for n in ("norm1", "norm2"):
                take(e + n + ".weight", (d,)), take(e + n + ".bias", (d,))
        for l in range(L):
            p = f"decoder.layers.{l}."
            names = ([f"{p}chan_processing_layers.{c}." for c in range(A)]
                     + [f"{p}chan_interaction_layers.{c}.interactions.{j}." for c in range(A) for j in range(A - 1)]
                     + [f"{p}chan_input_layers.{c}." for c in range(A)])
            stack = {}
            for field, kind in _BLOCK_FIELDS:
                shape = size[kind]
                base = self.off
                stack[field] = base
                n = _numel(shape)
                for b, name in enumerate(names):
                    self.views[name + field] = (base + b * n, shape)
                self.off += _up(NB * n)
            # per-channel fuse of the interaction outputs, stacked over channels
            for field, shape in (("linear.0.weight", ((A - 1) * d,)), ("linear.0.bias", ((A - 1) * d,)),
                                 ("linear.1.weight", (d, (A - 1) * d)), ("linear.1.bias", (d,))):
                base = self.off
                stack["inter." + field] = base
                n = _numel(shape)
                for c in range(A):
                    self.views[f"{p}chan_interaction_layers.{c}.{field}"] = (base + c * n, shape)
                self.off += _up(A * n)
            stack["ff_ln_w"], stack["ff_ln_b"] = take(p + "feed_forward.0.weight", (d,)), take(p + "feed_forward.0.bias", (d,))
            stack["ff_w"], stack["ff_b"] = take(p + "feed_forward.1.weight", (d, d)), take(p + "feed_forward.1.bias", (d,))
            stack["ln_w"], stack["ln_b"] = take(p + "layer_norm.weight", (d,)), take(p + "layer_norm.bias", (d,))
            self.stacks.append(stack)
        take("linear.0.weight", (A * d,)), take("linear.0.bias", (A * d,))
        take("linear.1.weight", (d, A * d)), take("linear.1.bias", (d,))
        self.head_base = self.off
        for k, (o, shape) in _build_views(head_dims, head_lay).items():
            if k.startswith("predictors."):
                self.views[k] = (self.head_base + o, shape)
        self.off += _up(head_lay.total)
        self.total = self.off


def _default_init(key, shape):
    """PyTorch's default initialisers by tensor role (not draw-for-draw identical to the reference's
    construction order: seed-for-seed initial parity is not provided for this variant)."""
    t = torch.empty(shape)
    if key.endswith("layer_norm.weight") or ".norm" in key and key.endswith("weight") or key in ("tgt_embedding.0.weight", "linear.0.weight") \
            or key.endswith("feed_forward.0.weight") or (".linear.0.weight" in key and "interaction" in key) \
            or (key.startswith("predictors.") and (".linear.0.weight" in key or ".linear.3.weight" in key or ".linear.6.weight" in key)):
        return t.fill_(1.0)
    if t.dim() == 1:
        if "layer_norm.bias" in key or ".norm" in key or key in ("tgt_embedding.0.bias", "linear.0.bias") \
                or key.endswith("feed_forward.0.bias") or (".linear.0.bias" in key and "interaction" in key) or "in_proj_bias" in key \
                or "out_proj.bias" in key or (key.startswith("predictors.") and (".linear.0.bias" in key or ".linear.3.bias" in key or ".linear.6.bias" in key)):
            return t.zero_()
        return t.uniform_(-0.05, 0.05)
    if key == "src_embedding.weight":
        return t.normal_()
    if "in_proj_weight" in key:
        return nn.init.xavier_uniform_(t)
    bound = 1.0 / math.sqrt(shape[-1])
    return t.uniform_(-bound, bound)


class ArtSpeechTransformer(nn.Module):
    def __init__(self, vocab_size: int, num_articulators: int, embed_dim: int = 64, num_heads: int = 4, num_layers: int = 4,
                 num_feat: int = 100, dropout: float = 0.):
        super().__init__()
        if embed_dim % num_heads or (embed_dim // num_heads) % 4 or num_feat % 2:
            raise NotImplementedError("artspeech_amd transformer needs head_dim = embed_dim / num_heads to be a multiple of 4 "
                                      "(16-byte aligned head slices) and an even num_feat")
        self.embed_dim, self.num_heads, self.num_layers = embed_dim, num_heads, num_layers
        self.num_articulators, self.num_feat, self.vocab_size = num_articulators, num_feat, vocab_size
        self.dropout = float(dropout)
        self.head_dims = _lib.Dims(1, num_articulators, 1, embed_dim, num_feat // 2, 1)
        self.head_lay = _lib.layout(self.head_dims)
        self.lay = _Layout(vocab_size, num_articulators, embed_dim, num_layers, num_feat, self.head_lay, self.head_dims)
        flat = torch.zeros(self.lay.total)
        for k, (o, shape) in self.lay.views.items():
            flat[o:o + _numel(shape)] = _default_init(k, shape).reshape(-1)
        # the reference's decoder layers are deep copies of one layer (identical initial weights, :326-329)
        for l in range(1, num_layers):
            for k, (o, shape) in self.lay.views.items():
                if k.startswith(f"decoder.layers.{l}."):
                    o0 = self.lay.views[k.replace(f"decoder.layers.{l}.", "decoder.layers.0.", 1)][0]
                    flat[o:o + _numel(shape)] = flat[o0:o0 + _numel(shape)]
        self.flat = nn.Parameter(flat)
        position = torch.arange(5000).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, embed_dim, 2) * (-math.log(10000.0) / embed_dim))
        pe = torch.zeros(1, 5000, embed_dim)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)  # exposed as "pos_encoding.pe" like the reference's persistent buffer
        self.register_buffer("start", torch.zeros(1, 1, num_articulators, num_feat), persistent=False)
        self._cache = {}

    # ------------------------------------------------------------------ state_dict contract
    def named_views(self):
        return {k: self.flat.detach()[o:o + _numel(s)].view(s) for k, (o, s) in self.lay.views.items()}

    @property
    def total_parameters(self):
        return sum(_numel(s) for _, s in self.lay.views.values())

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for k, v in self.named_views().items():
            destination[prefix + k] = v if keep_vars else v.clone()
        destination[prefix + "pos_encoding.pe"] = self.pe if keep_vars else self.pe.clone()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        views = self.named_views()
        views["pos_encoding.pe"] = self.pe
        for k, dst in views.items():
            key = prefix + k
            if key not in state_dict:
                missing_keys.append(key)
            elif tuple(state_dict[key].shape) != tuple(dst.shape):
                error_msgs.append(f"size mismatch for {key}: copying a param with shape {tuple(state_dict[key].shape)} from "
                                  f"checkpoint, the shape in current model is {tuple(dst.shape)}.")
            else:
                with torch.no_grad():
                    dst.copy_(state_dict[key])
        if strict:
            unexpected_keys.extend(k for k in state_dict if k.startswith(prefix) and k[len(prefix):] not in views)

    # ------------------------------------------------------------------ device helpers
    def _P(self, off):
        return self.flat.data_ptr() + 4 * off

    def _gemm(self, **kw):
        g = _lib.Gemm()
        g.batch = 1
        for k, v in kw.items():
            setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
        _lib.check(_lib.lib().as_gemm_f32(C.byref(g), _lib.stream_ptr()), "as_gemm_f32")

    def _offsets(self, values):
        return torch.tensor(values, dtype=torch.int64, device=self.flat.device)

    def _scratch(self, B, T, Tm):
        """Scratch buffers, allocated once for the largest (B, T, Tm) seen (generate() grows T step by step)."""
        cap = self._cache.get("cap")
        if cap is not None and cap[0] >= B and cap[1] >= T and cap[2] >= Tm:
            return self._cache["scratch"]
        if cap is not None:
            B, T, Tm = max(B, cap[0]), max(T, cap[1]), max(Tm, cap[2])
        dev, d, A, h = self.flat.device, self.embed_dim, self.num_articulators, self.num_heads
        R, NBmax = B * T, max(A * (A - 1), A)
        Rk = B * max(T, Tm)
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
        self._cache["scratch"] = dict(
            x=f(R, A, d), xhat=f(R, A, d), proc=f(R, A, d), inter_cat=f(R, A, max(A - 1, 1) * d), inter=f(R, A, d), inp=f(R, A, d),
            kb=f(NBmax, Rk, d), vb=f(NBmax, Rk, d), q2=f(NBmax, R, d), k2=f(NBmax, Rk, d), v2=f(NBmax, Rk, d), ctx=f(NBmax, R, d),
            scores=f(NBmax * B * h, T, max(T, Tm)), wq=f(A * (A + 1), d, d), bq=f(A * (A + 1), d), wk=f(A * (A + 1), d, d),
            bk=f(A * (A + 1), d), wv=f(A * (A + 1), d, d), bv=f(A * (A + 1), d), memhat=f(B * Tm, d),
            cat_hat=f(R, A, max(A - 1, 1) * d), wlin=f(A, d, max(A - 1, 1) * d), blin=f(A, d), ffw=f(d, d), ffb=f(d), lnx=f(R, A, d),
            feat_hat=f(R, A * d), wfin=f(d, A * d), bfin=f(d), feat=f(R, d), temb_hat=f(R * A, self.num_feat),
            wemb=f(d, self.num_feat), bemb=f(d),
            head_ws=f(_lib.lib().as_head_workspace_floats(C.byref(self.head_dims), R)),
            enc_qkv=f(B * Tm, 3 * d), enc_scores=f(B * h, Tm, Tm), enc_ctx=f(B * Tm, d), enc_tmp=f(B * Tm, d),
            enc_ff=f(B * Tm, FF_DIM), enc_x=f(B * Tm, d),
        )
        self._cache["cap"] = (B, T, Tm)
        return self._cache["scratch"]

    def _work(self, B, T, Tm):
        """Scratch (shared) + the offset tables of this exact (B, T, Tm).  T: target length, Tm: source length."""
        scratch = self._scratch(B, T, Tm)
        key = (B, T, Tm)
        if key in self._cache:
            buf = dict(scratch)
            buf.update(self._cache[key])
            return buf
        d, A, h = self.embed_dim, self.num_articulators, self.num_heads
        R, dh = B * T, d // h
        buf = {}
        buf["pe_tgt"] = self.pe[0, :T].repeat(1, A).contiguous()  # (T, A*d): the same position row for every channel
        # ---- offset tables (in floats) for the three block groups
        Ad = A * d
        groups = {}
        blocks = {"proc": [(c, c, c) for c in range(A)],                       # (block slot, tgt channel, src channel)
                  "inter": [(c * (A - 1) + j, i, c) for c in range(A) for j, i in enumerate([i for i in range(A) if i != c])],
                  "input": [(c, c, None) for c in range(A)]}
        wbase = {"proc": 0, "inter": A, "input": A + A * (A - 1)}
        for name, blks in blocks.items():
            nb = len(blks)
            Tk = Tm if name == "input" else T
            Rk = B * Tk
            if name == "inter":   # q is written into the concat buffer: row stride A*(A-1)*d, slot (c, j)
                out_off = [c * (A - 1) * d + j * d for c in range(A) for j in range(A - 1)]
                out_ld = A * (A - 1) * d
            else:
                out_off = [c * d for c in range(A)]
                out_ld = Ad
            g = dict(nb=nb, Tk=Tk, out_ld=out_ld,
                     q_src=self._offsets([t * d for _, t, _ in blks]),                       # into xhat-like [R][A][d]
                     kv_src=self._offsets([0 if s is None else s * d for _, _, s in blks]),
                     w=self._offsets([(wbase[name] + i) * d * d for i in range(nb)]),
                     b=self._offsets([(wbase[name] + i) * d for i in range(nb)]),
                     w3=self._offsets([(wbase[name] + i) * 3 * d * d for i in range(nb)]),
                     b3=self._offsets([(wbase[name] + i) * 3 * d for i in range(nb)]),
                     out=self._offsets(out_off),
                     blk_q=self._offsets([i * R * d for i in range(nb)]), blk_k=self._offsets([i * Rk * d for i in range(nb)]),
                     # attention level: z = (block, b, head)
                     zq=self._offsets([i * R * d + b * T * d + hh * dh for i in range(nb) for b in range(B) for hh in range(h)]),
                     zk=self._offsets([i * Rk * d + b * Tk * d + hh * dh for i in range(nb) for b in range(B) for hh in range(h)]),
                     zs=self._offsets([z * T * Tk for z in range(nb * B * h)]))
            groups[name] = g
        buf["groups"] = groups
        # encoder attention offsets: z = (b, head) into the packed [R][3d] projection
        buf["enc"] = dict(q=self._offsets([b * Tm * 3 * d + hh * dh for b in range(B) for hh in range(h)]),
                          c=self._offsets([b * Tm * d + hh * dh for b in range(B) for hh in range(h)]),
                          s=self._offsets([z * Tm * Tm for z in range(B * h)]))
        self._cache[key] = buf
        out = dict(scratch)
        out.update(buf)
        return out

    # ------------------------------------------------------------------ encoder
    def _encode(self, src, src_key_padding_mask, zero_padded):
        L = _lib.lib()
        st = _lib.stream_ptr()
        B, Tm = src.shape
        d, h = self.embed_dim, self.num_heads
        dh = d // h
        bufs = self._work(B, Tm, Tm)
        e = dict(bufs["enc"], qkv=bufs["enc_qkv"], scores=bufs["enc_scores"], ctx=bufs["enc_ctx"], tmp=bufs["enc_tmp"],
                 ff=bufs["enc_ff"], x=bufs["enc_x"])
        V = self.lay.views
        x = e["x"]
        pe = self.pe[0, :Tm].contiguous()
        _lib.check(L.as_embed_posenc(_lib.ptr(src), src.stride(0), self._P(V["src_embedding.weight"][0]), _lib.ptr(pe), _lib.ptr(x),
                                     B * Tm, Tm, d, st), "as_embed_posenc")
        kpm = src_key_padding_mask
        for l in range(self.num_layers):
            p = f"encoder.layers.{l}."
            o = lambda k: self._P(V[p + k][0])  # noqa: E731
            self._gemm(A=x, B=o("self_attn.in_proj_weight"), C=e["qkv"], bias=o("self_attn.in_proj_bias"), M=B * Tm, N=3 * d, K=d,
                       a_i=d, a_k=1, b_j=d, b_k=1, ldc=3 * d)
            qkv = e["qkv"].data_ptr()
            self._gemm(A=qkv, B=qkv + 4 * d, C=e["scores"], M=Tm, N=Tm, K=dh, a_i=3 * d, a_k=1, b_j=3 * d, b_k=1, ldc=Tm,
                       batch=B * h, a_off=e["q"], b_off=e["q"], c_off=e["s"])
            _lib.check(L.as_attn_softmax(_lib.ptr(e["scores"]), B * h, Tm, Tm, h, B, 1.0 / math.sqrt(dh), None, _lib.ptr(kpm), st),
                       "as_attn_softmax")
            self._gemm(A=e["scores"], B=qkv + 8 * d, C=e["ctx"], M=Tm, N=dh, K=Tm, a_i=Tm, a_k=1, b_j=1, b_k=3 * d, ldc=d,
                       batch=B * h, a_off=e["s"], b_off=e["q"], c_off=e["c"])
            self._gemm(A=e["ctx"], B=o("self_attn.out_proj.weight"), C=e["tmp"], bias=o("self_attn.out_proj.bias"), M=B * Tm, N=d, K=d,
                       a_i=d, a_k=1, b_j=d, b_k=1, ldc=d)
            _lib.check(L.as_layernorm_fwd(_lib.ptr(x), _lib.ptr(e["tmp"]), o("norm1.weight"), o("norm1.bias"), _lib.ptr(x), None, None,
                                          B * Tm, d, 0, st), "as_layernorm_fwd")
            self._gemm(A=x, B=o("linear1.weight"), C=e["ff"], bias=o("linear1.bias"), M=B * Tm, N=FF_DIM, K=d, a_i=d, a_k=1, b_j=d,
                       b_k=1, ldc=FF_DIM, act=1)
            self._gemm(A=e["ff"], B=o("linear2.weight"), C=e["tmp"], bias=o("linear2.bias"), M=B * Tm, N=d, K=FF_DIM, a_i=FF_DIM, a_k=1,
                       b_j=FF_DIM, b_k=1, ldc=d)
            _lib.check(L.as_layernorm_fwd(_lib.ptr(x), _lib.ptr(e["tmp"]), o("norm2.weight"), o("norm2.bias"), _lib.ptr(x), None, None,
                                          B * Tm, d, 0, st), "as_layernorm_fwd")
        if zero_padded and kpm is not None:
            # nn.TransformerEncoder's nested-tensor fast path (no grad): padded source positions come back as zeros
            valid = (~torch.isinf(kpm)).to(torch.float32).reshape(-1).contiguous()
            _lib.check(L.as_row_scale(_lib.ptr(x), _lib.ptr(valid), _lib.ptr(x), B * Tm, d, st), "as_row_scale")
        return x[:B * Tm].clone()

    # ------------------------------------------------------------------ decoder
    def _fold_blocks(self, buf, stack):
        """W' = W.diag(gamma), b' = b + W.beta for the q/k/v pre-projections of every block of the layer."""
        L, st = _lib.lib(), _lib.stream_ptr()
        NB, d = self.num_articulators * (self.num_articulators + 1), self.embed_dim
        g, b = self._P(stack["layer_norm.weight"]), self._P(stack["layer_norm.bias"])
        for name, wdst, bdst in (("query", "wq", "bq"), ("key", "wk", "bk"), ("value", "wv", "bv")):
            _lib.check(L.as_fold_ln(self._P(stack[f"{name}.0.weight"]), g, b, self._P(stack[f"{name}.0.bias"]), _lib.ptr(buf[wdst]),
                                    _lib.ptr(buf[bdst]), NB, d, d, st), "as_fold_ln")

    def _group(self, buf, stack, name, tgt_hat, src_hat, src_ld, out, attn_mask, kpm, B, T):
        """One group of channel blocks (ChannelProcessingLayer :70-100) as grouped GEMMs."""
        L, st = _lib.lib(), _lib.stream_ptr()
        g = buf["groups"][name]
        d, h, A = self.embed_dim, self.num_heads, self.num_articulators
        dh, nb, Tk = d // h, g["nb"], g["Tk"]
        R, Rk, Ad = B * T, B * Tk, A * d
        # q = relu(x_hat_tgt W_q'^T + b_q') -> straight into the block's output slot (the residual of :98)
        self._gemm(A=tgt_hat, B=buf["wq"], C=out, bias=buf["bq"], M=R, N=d, K=d, a_i=Ad, a_k=1, b_j=d, b_k=1, ldc=g["out_ld"], act=1,
                   batch=nb, a_off=g["q_src"], b_off=g["w"], c_off=g["out"], bias_off=g["b"])
        for wsrc, bsrc, dst in (("wk", "bk", "kb"), ("wv", "bv", "vb")):
            self._gemm(A=src_hat, B=buf[wsrc], C=buf[dst], bias=buf[bsrc], M=Rk, N=d, K=d, a_i=src_ld, a_k=1, b_j=d, b_k=1, ldc=d, act=1,
                       batch=nb, a_off=g["kv_src"], b_off=g["w"], c_off=g["blk_k"], bias_off=g["b"])
        # nn.MultiheadAttention in-projections (rows [0:d], [d:2d], [2d:3d] of in_proj_weight)
        ipw, ipb = self._P(stack["multihead_attn.in_proj_weight"]), self._P(stack["multihead_attn.in_proj_bias"])
        self._gemm(A=out, B=ipw, C=buf["q2"], bias=ipb, M=R, N=d, K=d, a_i=g["out_ld"], a_k=1, b_j=d, b_k=1, ldc=d, batch=nb,
                   a_off=g["out"], b_off=g["w3"], c_off=g["blk_q"], bias_off=g["b3"])
        for src, dst, part in (("kb", "k2", 1), ("vb", "v2", 2)):
            self._gemm(A=buf[src], B=ipw + 4 * part * d * d, C=buf[dst], bias=ipb + 4 * part * d, M=Rk, N=d, K=d, a_i=d, a_k=1, b_j=d,
                       b_k=1, ldc=d, batch=nb, a_off=g["blk_k"], b_off=g["w3"], c_off=g["blk_k"], bias_off=g["b3"])
        Z = nb * B * h
        self._gemm(A=buf["q2"], B=buf["k2"], C=buf["scores"], M=T, N=Tk, K=dh, a_i=d, a_k=1, b_j=d, b_k=1, ldc=Tk, batch=Z,
                   a_off=g["zq"], b_off=g["zk"], c_off=g["zs"])
        _lib.check(L.as_attn_softmax(_lib.ptr(buf["scores"]), Z, T, Tk, h, B, 1.0 / math.sqrt(dh),
                                     _lib.ptr(attn_mask) if attn_mask is not None else None,
                                     _lib.ptr(kpm) if kpm is not None else None, st), "as_attn_softmax")
        self._gemm(A=buf["scores"], B=buf["v2"], C=buf["ctx"], M=T, N=dh, K=Tk, a_i=Tk, a_k=1, b_j=1, b_k=d, ldc=d, batch=Z,
                   a_off=g["zs"], b_off=g["zk"], c_off=g["zq"])
        # out-projection accumulated onto q:  out = q + (ctx W_o^T + b_o)
        self._gemm(A=buf["ctx"], B=self._P(stack["multihead_attn.out_proj.weight"]), C=out, bias=self._P(stack["multihead_attn.out_proj.bias"]),
                   M=R, N=d, K=d, a_i=d, a_k=1, b_j=d, b_k=1, ldc=g["out_ld"], accumulate=1, batch=nb, a_off=g["blk_q"], b_off=g["w"],
                   c_off=g["out"], bias_off=g["b"])

    def _decoder_layer(self, buf, l, mem_hat, tgt_mask, memory_mask, tgt_kpm, mem_kpm, B, T):
        L, st = _lib.lib(), _lib.stream_ptr()
        A, d = self.num_articulators, self.embed_dim
        R, stack = B * T, self.lay.stacks[l]
        ln = lambda x, xhat, rows, D: _lib.check(L.as_layernorm_fwd(_lib.ptr(x), None, None, None, None, _lib.ptr(xhat), None, rows, D, 0, st))  # noqa: E731
        self._fold_blocks(buf, stack)
        # self-channel blocks (:234-242): src = tgt = channel c
        ln(buf["x"], buf["xhat"], R * A, d)
        self._group(buf, stack, "proc", buf["xhat"], buf["xhat"], A * d, buf["proc"], tgt_mask, tgt_kpm, B, T)
        # channel interactions (:246-261): block (c, j): src = channel c, tgt = the j-th other channel
        ln(buf["proc"], buf["xhat"], R * A, d)
        self._group(buf, stack, "inter", buf["xhat"], buf["xhat"], A * d, buf["inter_cat"], tgt_mask, tgt_kpm, B, T)
        K10 = (A - 1) * d
        ln(buf["inter_cat"], buf["cat_hat"], R * A, K10)
        _lib.check(L.as_fold_ln(self._P(stack["inter.linear.1.weight"]), self._P(stack["inter.linear.0.weight"]),
                                self._P(stack["inter.linear.0.bias"]), self._P(stack["inter.linear.1.bias"]), _lib.ptr(buf["wlin"]),
                                _lib.ptr(buf["blin"]), A, d, K10, st), "as_fold_ln")
        self._gemm(A=buf["cat_hat"], B=buf["wlin"], C=buf["inter"], bias=buf["blin"], M=R, N=d, K=K10, a_i=A * K10, a_k=1, b_j=K10, b_k=1,
                   ldc=A * d, act=1, batch=A, a_batch=K10, b_batch=d * K10, c_batch=d, bias_batch=d)
        # cross attention to the encoder memory (:263-271): src = memory, tgt = fused channel
        ln(buf["inter"], buf["xhat"], R * A, d)
        self._group(buf, stack, "input", buf["xhat"], mem_hat, d, buf["inp"], memory_mask, mem_kpm, B, T)
        # x = LN(inp); out = x + relu(Linear(LN_ff(x)))   (:273-275)
        _lib.check(L.as_layernorm_fwd(_lib.ptr(buf["inp"]), None, self._P(stack["ln_w"]), self._P(stack["ln_b"]), _lib.ptr(buf["x"]), None,
                                      None, R * A, d, 0, st), "as_layernorm_fwd")
        ln(buf["x"], buf["lnx"], R * A, d)
        _lib.check(L.as_fold_ln(self._P(stack["ff_w"]), self._P(stack["ff_ln_w"]), self._P(stack["ff_ln_b"]), self._P(stack["ff_b"]),
                                _lib.ptr(buf["ffw"]), _lib.ptr(buf["ffb"]), 1, d, d, st), "as_fold_ln")
        self._gemm(A=buf["lnx"], B=buf["ffw"], C=buf["x"], bias=buf["ffb"], M=R * A, N=d, K=d, a_i=d, a_k=1, b_j=d, b_k=1, ldc=d, act=1,
                   accumulate=1)

    def _generate_one_step(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                           memory_key_padding_mask=None):
        """(bs, seq_len, num_channels, num_feat) -> (bs, seq_len, num_channels, 2, num_feat / 2)  (reference :430-474)."""
        L, st = _lib.lib(), _lib.stream_ptr()
        B, T, A, nf = tgt.shape
        d, V = self.embed_dim, self.lay.views
        Tm = memory.shape[0] // B
        buf = self._work(B, T, Tm)
        R = B * T
        tgt = tgt.contiguous().float()
        # tgt_embedding: LN(nf) -> Linear(nf, d) -> ReLU, then + positional encoding per channel (:441-454)
        _lib.check(L.as_layernorm_fwd(_lib.ptr(tgt), None, None, None, None, _lib.ptr(buf["temb_hat"]), None, R * A, nf, 0, st))
        _lib.check(L.as_fold_ln(self._P(V["tgt_embedding.1.weight"][0]), self._P(V["tgt_embedding.0.weight"][0]),
                                self._P(V["tgt_embedding.0.bias"][0]), self._P(V["tgt_embedding.1.bias"][0]), _lib.ptr(buf["wemb"]),
                                _lib.ptr(buf["bemb"]), 1, d, nf, st), "as_fold_ln")
        self._gemm(A=buf["temb_hat"], B=buf["wemb"], C=buf["x"], bias=buf["bemb"], M=R * A, N=d, K=nf, a_i=nf, a_k=1, b_j=nf, b_k=1, ldc=d, act=1)
        _lib.check(L.as_embed_posenc(None, 0, None, _lib.ptr(buf["pe_tgt"]), _lib.ptr(buf["x"]), R, T, A * d, st), "as_embed_posenc")
        # memory is shared by every cross-attention block: one affine-free normalisation
        _lib.check(L.as_layernorm_fwd(_lib.ptr(memory), None, None, None, None, _lib.ptr(buf["memhat"]), None, B * Tm, d, 0, st))
        for l in range(self.num_layers):
            self._decoder_layer(buf, l, buf["memhat"], tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask, B, T)
        # (B, T, A*d) features -> LN -> Linear -> ReLU -> heads -> sigmoid (:466-472)
        _lib.check(L.as_layernorm_fwd(_lib.ptr(buf["x"]), None, None, None, None, _lib.ptr(buf["feat_hat"]), None, R, A * d, 0, st))
        _lib.check(L.as_fold_ln(self._P(V["linear.1.weight"][0]), self._P(V["linear.0.weight"][0]), self._P(V["linear.0.bias"][0]),
                                self._P(V["linear.1.bias"][0]), _lib.ptr(buf["wfin"]), _lib.ptr(buf["bfin"]), 1, d, A * d, st), "as_fold_ln")
        self._gemm(A=buf["feat_hat"], B=buf["wfin"], C=buf["feat"], bias=buf["bfin"], M=R, N=d, K=A * d, a_i=A * d, a_k=1, b_j=A * d, b_k=1,
                   ldc=d, act=1)
        out = torch.empty((B, T, A, 2, nf // 2), dtype=torch.float32, device=tgt.device)
        _lib.check(L.as_head_fwd(C.byref(self.head_dims), C.byref(self.head_lay), self._P(self.lay.head_base), _lib.ptr(buf["feat"]), R,
                                 _lib.ptr(out), _lib.ptr(buf["head_ws"]), 0, st), "as_head_fwd")
        return out

    # ------------------------------------------------------------------ public API
    def _check(self, src):
        _lib.require_gpu(src, "src")
        _lib.require_gpu(self.flat, "model parameters")
        if self.training and self.dropout > 0.0:
            raise NotImplementedError("ArtSpeechTransformer: dropout > 0 in training mode is not built")

    def forward(self, src, tgt, src_attn_mask=None, tgt_attn_mask=None, memory_mask=None, src_key_padding_mask=None,
                tgt_key_padding_mask=None, memory_key_padding_mask=None):
        """src (bs, seq_len) int64, tgt (bs, seq_len, num_channels, num_feat); float masks as built by
        ``pad_sequence_transformer_collate_fn``.  As in the reference (:380-387) the cross-attention mask is
        ``src_attn_mask`` and ``memory_mask`` / ``memory_key_padding_mask`` arguments are not used by the decoder call
        of ``forward``.  Inference only: the result carries no autograd graph."""
        self._check(src)
        zero_padded = self._zero_padded(torch.is_grad_enabled())
        with torch.no_grad():
            f = lambda m: None if m is None else m.contiguous().float()  # noqa: E731
            memory = self._encode(src.long().contiguous(), f(src_key_padding_mask), zero_padded=zero_padded)
            return self._generate_one_step(tgt, memory, tgt_mask=f(tgt_attn_mask), memory_mask=f(src_attn_mask),
                                           tgt_key_padding_mask=f(tgt_key_padding_mask), memory_key_padding_mask=None)

    _grad_mode_hint = None

    def _zero_padded(self, grad_enabled):
        """The reference's encoder returns zeros at padded positions exactly when nn.TransformerEncoder takes its
        nested-tensor fast path: eval() mode and no gradient tracking (torch.no_grad(), or frozen parameters).
        `set_encoder_grad_mode` overrides the detection (tests)."""
        if self._grad_mode_hint is not None:
            return not self._grad_mode_hint
        return (not self.training) and not (grad_enabled and self.flat.requires_grad)

    def set_encoder_grad_mode(self, grad_mode):
        self._grad_mode_hint = grad_mode

    def generate(self, src, src_key_padding_mask):
        """Autoregressive decoding exactly as the reference (:391-427): the encoder once, then seq_len full
        re-decodes of the growing prefix without target masks.  Returns (bs, seq_len, num_channels, 2, num_samples)."""
        self._check(src)
        with torch.no_grad():
            B, T = src.shape
            kpm = src_key_padding_mask.contiguous().float()
            memory = self._encode(src.long().contiguous(), kpm, zero_padded=True)
            tgt = self.start.repeat(B, 1, 1, 1)
            for _ in range(T):
                nxt = self._generate_one_step(tgt, memory, memory_key_padding_mask=kpm)
                tgt = torch.cat([tgt, nxt[:, -1:].reshape(B, 1, self.num_articulators, self.num_feat)], dim=1)
            return tgt.reshape(B, T + 1, self.num_articulators, 2, self.num_feat // 2)[:, 1:]
