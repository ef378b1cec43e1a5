"""Transformer variant of the phoneme-to-articulation network on MI355X (forward, backward, generate).

Mirrors reference ``phoneme_to_articulation/transformer/models.py``: ``ArtSpeechTransformer`` (:280-474) with the
standard post-norm encoder (:309-318) and the custom multi-channel decoder (``MultiChannelTransformerDecoderLayer``
:165-277 = A self-channel blocks + A*(A-1) channel-interaction blocks + A cross-attention blocks per layer, each a
``ChannelProcessingLayer`` :37-100).  Constructor signature, ``forward`` / ``generate`` contracts and ``state_dict``
keys are the reference's.  The quirks of SURVEY appendix A.7 are reproduced: ONE LayerNorm shared by src and tgt
inside a block, the residual adds the *projected* query, ReLU after the q/k/v pre-projections, the cross-attention of
``forward`` is causally masked, the encoder layer keeps the library defaults (post-norm, ReLU, dim_feedforward 2048,
dropout 0.1 in training mode) and -- when the reference would take nn.TransformerEncoder's nested-tensor fast path
(eval() without gradient tracking) -- padded source positions of the memory are zeros.

MI355X design: the A*(A+1) channel blocks of a decoder layer run as GROUPED fp32-MFMA GEMMs (``as_gemm_f32`` with
per-batch offset tables) on block-major tensors, i.e. ~30 launches per layer instead of the reference's ~2 000 small
kernels; every LayerNorm that feeds a Linear is applied affine-free once per channel and its gamma/beta are folded
into the consumers' weights; the attention core is one fused kernel per group.  Each of the three block groups of a
layer (self, interaction, cross) is ONE autograd node with a hand-written backward (``ops.ChannelBlocks``): residuals,
ReLU masks and the per-channel gradient sums ride in GEMM epilogues / segmented reductions instead of separate passes
over the [blocks, rows, d] activations.  The other building blocks are autograd Functions whose forward and backward
both run on the C ABI (``ops.py``); autograd only wires them.  Parameters are stored stacked over the blocks of a group
(one tensor per field, group and layer: a group's gradient is then a whole tensor, not a slice).
"""
import math
import os

import torch
import torch.utils.checkpoint
import torch.nn as nn
import torch.nn.functional as F

from ... import _lib
from ..encoder_decoder.models import _build_views, _numel, _reference_init
from .ops import (Attention, ChannelBlocks, FoldLN, GroupedLinear, Heads, LayerNormAffine, Normalize, NormalizeRes,
                  channel_blocks_forward)

FF_DIM = 2048      # nn.TransformerEncoderLayer's default dim_feedforward (not overridden by the reference, :309-313)
ENC_DROPOUT = 0.1  # ... and its default dropout: the model's `dropout` argument does not reach the encoder

_BLOCK = (("query.0.weight", "q_w"), ("query.0.bias", "q_b"), ("key.0.weight", "k_w"), ("key.0.bias", "k_b"),
          ("value.0.weight", "v_w"), ("value.0.bias", "v_b"), ("multihead_attn.in_proj_weight", "in_w"),
          ("multihead_attn.in_proj_bias", "in_b"), ("multihead_attn.out_proj.weight", "o_w"),
          ("multihead_attn.out_proj.bias", "o_b"), ("layer_norm.weight", "ln_w"), ("layer_norm.bias", "ln_b"))


def _uniform(shape, fan_in):
    bound = 1.0 / math.sqrt(fan_in)
    return torch.empty(shape).uniform_(-bound, bound)


def _reference_order_state(vocab_size, A, d, heads, L, nf):
    """Default PyTorch initialisation drawn in the reference's construction order (transformer/models.py:291-343), so that
    the same torch seed yields the same initial weights.  The torch modules built here only consume the generator the way
    the reference's constructor does and hand over their tensors (reference key -> tensor); nn.TransformerEncoder /
    nn.TransformerDecoder deep-copy ONE constructed layer, so all layers of a stack start identical (:309-329)."""
    sd = {}

    def take(prefix, module):
        for k, v in module.state_dict().items():
            sd[prefix + k] = v.detach()

    def channel_block(prefix):  # ChannelProcessingLayer.__init__ (:46-69)
        for name in ("query", "key", "value"):
            take(f"{prefix}{name}.0.", nn.Linear(d, d))
        take(prefix + "multihead_attn.", nn.MultiheadAttention(embed_dim=d, num_heads=heads, dropout=0.0, batch_first=True))
        take(prefix + "layer_norm.", nn.LayerNorm(d))

    take("src_embedding.", nn.Embedding(vocab_size, d))
    take("tgt_embedding.0.", nn.LayerNorm(nf))
    take("tgt_embedding.1.", nn.Linear(nf, d))
    enc = nn.TransformerEncoderLayer(d_model=d, nhead=heads, batch_first=True)
    for l in range(L):
        take(f"encoder.layers.{l}.", enc)
    first = len(sd)
    pre = "decoder.layers.0."
    for c in range(A):
        channel_block(f"{pre}chan_processing_layers.{c}.")
    for c in range(A):  # ChannelInteractionsLayer.__init__ (:114-131)
        for j in range(A - 1):
            channel_block(f"{pre}chan_interaction_layers.{c}.interactions.{j}.")
        take(f"{pre}chan_interaction_layers.{c}.linear.0.", nn.LayerNorm((A - 1) * d))
        take(f"{pre}chan_interaction_layers.{c}.linear.1.", nn.Linear((A - 1) * d, d))
    for c in range(A):
        channel_block(f"{pre}chan_input_layers.{c}.")
    take(pre + "feed_forward.0.", nn.LayerNorm(d))
    take(pre + "feed_forward.1.", nn.Linear(d, d))
    take(pre + "layer_norm.", nn.LayerNorm(d))
    layer0 = list(sd.items())[first:]
    for l in range(1, L):
        for k, v in layer0:
            sd[f"decoder.layers.{l}." + k[len(pre):]] = v
    take("linear.0.", nn.LayerNorm(A * d))
    take("linear.1.", nn.Linear(A * d, d))
    for a in range(A):
        for k, v in _reference_init(None, 1, None, None, nf // 2, False, in_features=d).items():
            sd[f"predictors.{a}.{k}"] = v
    return sd


# generate(): reuse the step-invariant memory-side K/V and restrict the last layer to the newest frame (both exact);
# ARTSPEECH_GENERATE_PLAIN=1 (or setting this to False) re-decodes everything like the reference, for A/B timing.
GENERATE_SAVINGS = os.environ.get("ARTSPEECH_GENERATE_PLAIN") is None


class ArtSpeechTransformer(nn.Module):
    def __init__(self, vocab_size: int, num_articulators: int, embed_dim: int = 64, num_heads: int = 4, num_layers: int = 4,
                 num_feat: int = 100, dropout: float = 0.):
        super().__init__()
        if embed_dim % num_heads or num_feat % 2:
            raise ValueError("embed_dim must be divisible by num_heads and num_feat must be even (x and y halves)")
        if embed_dim % 4 or num_articulators < 2:
            raise NotImplementedError("artspeech_amd transformer needs embed_dim to be a multiple of 4 (16-byte rows for the "
                                      "grouped GEMM epilogues) and at least 2 articulators")
        A, d, L, nf = num_articulators, embed_dim, num_layers, num_feat
        self.embed_dim, self.num_heads, self.num_layers = d, num_heads, L
        self.num_articulators, self.num_feat, self.vocab_size = A, nf, vocab_size
        self.dropout = float(dropout)
        self.head_dims = _lib.Dims(1, A, 1, d, nf // 2, 1)
        self.head_lay = _lib.layout(self.head_dims)
        K10 = (A - 1) * d
        self._group_sizes = {"proc": A, "inter": A * (A - 1), "input": A}
        P = {}
        # storage only: the values come from _reference_order_state() below (seed-for-seed with the reference)
        _uniform = lambda shape, fan: torch.empty(shape)  # noqa: E731
        P["src_emb"] = torch.empty(vocab_size, d)
        P["tgt_ln_w"], P["tgt_ln_b"] = torch.ones(nf), torch.zeros(nf)
        P["tgt_w"], P["tgt_b"] = _uniform((d, nf), nf), _uniform((d,), nf)
        for l in range(L):
            e = f"enc{l}_"
            P[e + "in_w"], P[e + "in_b"] = torch.empty(3 * d, d), torch.zeros(3 * d)
            P[e + "o_w"], P[e + "o_b"] = _uniform((d, d), d), torch.zeros(d)
            P[e + "l1_w"], P[e + "l1_b"] = _uniform((FF_DIM, d), d), _uniform((FF_DIM,), d)
            P[e + "l2_w"], P[e + "l2_b"] = _uniform((d, FF_DIM), FF_DIM), _uniform((d,), FF_DIM)
            for n in ("n1", "n2"):
                P[e + n + "_w"], P[e + n + "_b"] = torch.ones(d), torch.zeros(d)
        layer0 = {}
        for grp, NB in self._group_sizes.items():
            for name, shape in (("q_w", (NB, d, d)), ("q_b", (NB, d)), ("k_w", (NB, d, d)), ("k_b", (NB, d)), ("v_w", (NB, d, d)),
                                ("v_b", (NB, d)), ("o_w", (NB, d, d)), ("o_b", (NB, d)), ("in_w", (NB, 3 * d, d)), ("in_b", (NB, 3 * d)),
                                ("ln_w", (NB, d)), ("ln_b", (NB, d))):
                layer0[f"{grp}_{name}"] = torch.empty(shape)
        layer0["il_ln_w"], layer0["il_ln_b"] = torch.ones(A, K10), torch.zeros(A, K10)
        layer0["il_w"], layer0["il_b"] = _uniform((A, d, K10), K10), _uniform((A, d), K10)
        layer0["ff_ln_w"], layer0["ff_ln_b"] = torch.ones(d), torch.zeros(d)
        layer0["ff_w"], layer0["ff_b"] = _uniform((d, d), d), _uniform((d,), d)
        layer0["ln2_w"], layer0["ln2_b"] = torch.ones(d), torch.zeros(d)
        for l in range(L):  # the reference's decoder layers are deep copies of one layer (identical initial weights, :326-329)
            for k, v in layer0.items():
                P[f"dec{l}_{k}"] = v.clone()
        P["fin_ln_w"], P["fin_ln_b"] = torch.ones(A * d), torch.zeros(A * d)
        P["fin_w"], P["fin_b"] = _uniform((d, A * d), A * d), _uniform((d,), A * d)
        head = torch.zeros(self.head_lay.total)
        self._head_views = {k: v for k, v in _build_views(self.head_dims, self.head_lay).items() if k.startswith("predictors.")}
        for k, (o, shape) in self._head_views.items():
            if ".linear.0." in k or ".linear.3." in k or ".linear.6." in k:
                head[o:o + _numel(shape)] = 1.0 if k.endswith("weight") else 0.0
            else:
                fan = shape[-1] if len(shape) > 1 else (d if ".linear.1." in k else 256)
                head[o:o + _numel(shape)] = _uniform(shape, fan).reshape(-1)
        P["head_flat"] = head
        self.P = nn.ParameterDict({k: nn.Parameter(v) for k, v in P.items()})
        self._fold_cache = None  # {key: folded (W, b)} while generate() runs

        position = torch.arange(5000).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d, 2) * (-math.log(10000.0) / d))
        pe = torch.zeros(1, 5000, d)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)  # exposed as "pos_encoding.pe" like the reference's persistent buffer
        self.register_buffer("start", torch.zeros(1, 1, A, nf), persistent=False)
        self._map = self._build_key_map()
        init = _reference_order_state(vocab_size, A, d, num_heads, L, nf)
        init["pos_encoding.pe"] = self.pe
        self.load_state_dict(init)
        self._grad_mode_hint = None
        # block groups of a decoder layer: (tgt channel per block, src channel per block)
        inter_pairs = [(c, i) for c in range(A) for i in range(A) if i != c]
        self._groups = {"proc": (tuple(range(A)), tuple(range(A))),
                        "inter": (tuple(i for _, i in inter_pairs), tuple(c for c, _ in inter_pairs)),
                        "input": (tuple(range(A)), (0,) * A)}

    # ------------------------------------------------------------------ state_dict contract (the reference's keys)
    def _build_key_map(self):
        A, L = self.num_articulators, self.num_layers
        m = {"src_embedding.weight": ("src_emb", None), "tgt_embedding.0.weight": ("tgt_ln_w", None),
             "tgt_embedding.0.bias": ("tgt_ln_b", None), "tgt_embedding.1.weight": ("tgt_w", None),
             "tgt_embedding.1.bias": ("tgt_b", None), "linear.0.weight": ("fin_ln_w", None), "linear.0.bias": ("fin_ln_b", None),
             "linear.1.weight": ("fin_w", None), "linear.1.bias": ("fin_b", None)}
        for l in range(L):
            e, n = f"encoder.layers.{l}.", f"enc{l}_"
            for ref, mine in (("self_attn.in_proj_weight", "in_w"), ("self_attn.in_proj_bias", "in_b"),
                              ("self_attn.out_proj.weight", "o_w"), ("self_attn.out_proj.bias", "o_b"), ("linear1.weight", "l1_w"),
                              ("linear1.bias", "l1_b"), ("linear2.weight", "l2_w"), ("linear2.bias", "l2_b"), ("norm1.weight", "n1_w"),
                              ("norm1.bias", "n1_b"), ("norm2.weight", "n2_w"), ("norm2.bias", "n2_b")):
                m[e + ref] = (n + mine, None)
            p, n = f"decoder.layers.{l}.", f"dec{l}_"
            names = {"proc": [f"{p}chan_processing_layers.{c}." for c in range(A)],
                     "inter": [f"{p}chan_interaction_layers.{c}.interactions.{j}." for c in range(A) for j in range(A - 1)],
                     "input": [f"{p}chan_input_layers.{c}." for c in range(A)]}
            for grp, blocks in names.items():
                for b, name in enumerate(blocks):
                    for ref, mine in _BLOCK:
                        m[name + ref] = (f"{n}{grp}_{mine}", b)
            for c in range(A):
                q = f"{p}chan_interaction_layers.{c}.linear."
                m[q + "0.weight"], m[q + "0.bias"] = (n + "il_ln_w", c), (n + "il_ln_b", c)
                m[q + "1.weight"], m[q + "1.bias"] = (n + "il_w", c), (n + "il_b", c)
            m[p + "feed_forward.0.weight"], m[p + "feed_forward.0.bias"] = (n + "ff_ln_w", None), (n + "ff_ln_b", None)
            m[p + "feed_forward.1.weight"], m[p + "feed_forward.1.bias"] = (n + "ff_w", None), (n + "ff_b", None)
            m[p + "layer_norm.weight"], m[p + "layer_norm.bias"] = (n + "ln2_w", None), (n + "ln2_b", None)
        return m

    def named_views(self):
        """reference state_dict key -> view (shares storage with the stacked parameters)."""
        out = {}
        for key, (name, idx) in self._map.items():
            t = self.P[name].detach()
            out[key] = t if idx is None else t[idx]
        hf = self.P["head_flat"].detach()
        for k, (o, shape) in self._head_views.items():
            out[k] = hf[o:o + _numel(shape)].view(shape)
        return out

    def named_grad_views(self):
        out = {}
        for key, (name, idx) in self._map.items():
            g = self.P[name].grad
            if g is not None:
                out[key] = g if idx is None else g[idx]
        g = self.P["head_flat"].grad
        if g is not None:
            for k, (o, shape) in self._head_views.items():
                out[k] = g[o:o + _numel(shape)].view(shape)
        return out

    @property
    def total_parameters(self):
        return sum(v.numel() for v in self.named_views().values())

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for k, v in self.named_views().items():
            destination[prefix + k] = v if keep_vars else v.clone()
        destination[prefix + "pos_encoding.pe"] = self.pe if keep_vars else self.pe.clone()

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        # the stacked ParameterDict is an implementation detail: expose the reference's keys only
        destination = {} if destination is None else destination
        self._save_to_state_dict(destination, prefix, keep_vars)
        return destination

    def load_state_dict(self, state_dict, strict=True, assign=False):
        views = self.named_views()
        views["pos_encoding.pe"] = self.pe
        missing = [k for k in views if k not in state_dict]
        unexpected = [k for k in state_dict if k not in views]
        errors = []
        for k, dst in views.items():
            if k in state_dict:
                if tuple(state_dict[k].shape) != tuple(dst.shape):
                    errors.append(f"size mismatch for {k}: copying a param with shape {tuple(state_dict[k].shape)} from "
                                  f"checkpoint, the shape in current model is {tuple(dst.shape)}.")
                else:
                    with torch.no_grad():
                        dst.copy_(state_dict[k])
        if strict and unexpected:
            errors.insert(0, "Unexpected key(s) in state_dict: " + ", ".join(f'"{k}"' for k in unexpected) + ". ")
        if strict and missing:
            errors.insert(0, "Missing key(s) in state_dict: " + ", ".join(f'"{k}"' for k in missing) + ". ")
        if errors:
            raise RuntimeError("Error(s) in loading state_dict for ArtSpeechTransformer:\n\t" + "\n\t".join(errors))
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ------------------------------------------------------------------ encoder
    def _encode(self, src, src_key_padding_mask, zero_padded):
        """Embedding + positional encoding + L post-norm encoder layers (:368-378).  Returns memory [B*Tm, d]."""
        P, d, h = self.P, self.embed_dim, self.num_heads
        B, Tm = src.shape
        train = self.training
        x = F.embedding(src, P["src_emb"]) + self.pe[:, :Tm]
        x = F.dropout(x, self.dropout, train).reshape(B * Tm, d)
        kpm = src_key_padding_mask
        for l in range(self.num_layers):
            e = f"enc{l}_"
            qkv = GroupedLinear.apply(x[None], P[e + "in_w"].view(3, d, d), P[e + "in_b"].view(3, d), (0, 0, 0), False)
            ctx = Attention.apply(qkv[0:1], qkv[1:2], qkv[2:3], None, kpm, B, h)
            o = GroupedLinear.apply(ctx, P[e + "o_w"][None], P[e + "o_b"][None], (0,), False)[0]
            x = LayerNormAffine.apply(x, F.dropout(o, ENC_DROPOUT, train), P[e + "n1_w"], P[e + "n1_b"])
            h1 = GroupedLinear.apply(x[None], P[e + "l1_w"][None], P[e + "l1_b"][None], (0,), True)
            h2 = GroupedLinear.apply(F.dropout(h1, ENC_DROPOUT, train), P[e + "l2_w"][None], P[e + "l2_b"][None], (0,), False)[0]
            x = LayerNormAffine.apply(x, F.dropout(h2, ENC_DROPOUT, train), P[e + "n2_w"], P[e + "n2_b"])
        if zero_padded and kpm is not None:
            # nn.TransformerEncoder's nested-tensor fast path (eval, no grad): padded source positions come back as zeros
            x = x * (~torch.isinf(kpm)).to(x.dtype).reshape(-1, 1)
        return x

    # ------------------------------------------------------------------ decoder
    def _fold(self, W, gamma, beta, b):
        """FoldLN with a per-generate() cache: the folded weights depend on the parameters only, and generate() asks for
        the same ones at every one of its seq_len decoder passes."""
        cache = self._fold_cache
        if cache is None:
            return FoldLN.apply(W, gamma, beta, b)
        key = (W.data_ptr(), W.shape, gamma.data_ptr(), b.data_ptr())
        if key not in cache:
            cache[key] = FoldLN.apply(W, gamma, beta, b)
        return cache[key]

    def _memory_kv(self, l, mem_hat):
        """Source side of the cross-attention blocks of layer l (k / v pre-projections + MHA in-projections of the memory,
        :47-60 and the in_proj of :62-67): depends on the encoder output only, so generate() computes it once per layer
        instead of once per generated frame."""
        P, d = self.P, self.embed_dim
        _, src_idx = self._groups["input"]
        n = f"dec{l}_input_"
        ident = tuple(range(self._group_sizes["input"]))
        ln_w, ln_b = P[n + "ln_w"], P[n + "ln_b"]
        wk, bk = self._fold(P[n + "k_w"], ln_w, ln_b, P[n + "k_b"])
        wv, bv = self._fold(P[n + "v_w"], ln_w, ln_b, P[n + "v_b"])
        k = GroupedLinear.apply(mem_hat, wk, bk, src_idx, True)
        v = GroupedLinear.apply(mem_hat, wv, bv, src_idx, True)
        in_w, in_b = P[n + "in_w"], P[n + "in_b"]
        return (GroupedLinear.apply(k, in_w[:, d:2 * d], in_b[:, d:2 * d], ident, False),
                GroupedLinear.apply(v, in_w[:, 2 * d:], in_b[:, 2 * d:], ident, False))

    def _blocks(self, l, group, xhat_tgt, xhat_src, attn_mask, kpm, B, kv=None, cat=None):
        """One group of ChannelProcessingLayers (:70-100) on affine-free normalised inputs -> (out, q): the out-projections
        [G, R, d] -- with cat = (A, per) concatenated over the `per` blocks of each channel, [A, R, per * d] (:133-162) -- and
        the projected queries, the residual (:98) that the LayerNorm consuming the pair adds."""
        P = self.P
        tgt_idx, src_idx = self._groups[group]
        n = f"dec{l}_{group}_"
        if kv is None:
            return ChannelBlocks.apply(xhat_tgt, xhat_src, P[n + "q_w"], P[n + "q_b"], P[n + "k_w"], P[n + "k_b"], P[n + "v_w"],
                                       P[n + "v_b"], P[n + "in_w"], P[n + "in_b"], P[n + "o_w"], P[n + "o_b"], P[n + "ln_w"],
                                       P[n + "ln_b"], attn_mask, kpm, (tgt_idx, src_idx, B, self.num_heads, cat, torch.is_grad_enabled()))
        # generate(): the memory side (k2, v2) was projected once per call (_memory_kv); inference only, same arithmetic
        assert not torch.is_grad_enabled()
        return channel_blocks_forward(xhat_tgt, None, P[n + "q_w"], P[n + "q_b"], None, None, None, None, P[n + "in_w"], P[n + "in_b"],
                                      P[n + "o_w"], P[n + "o_b"], P[n + "ln_w"], P[n + "ln_b"], attn_mask, kpm,
                                      (tgt_idx, src_idx, B, self.num_heads, cat), False, kv2=kv)[:2]

    def _decoder_layer(self, l, x, mem_hat, tgt_mask, memory_mask, tgt_kpm, mem_kpm, B, mem_kv=None, last_only=False):
        """MultiChannelTransformerDecoderLayer.forward (:216-277) on channel-major x [A, R, d].  last_only (generate()'s last
        layer): only the newest frame's output is consumed, so every query side is restricted to that frame; the self blocks
        and the key / value side of the interaction blocks still see the whole prefix.  Returns [A, B, d] then."""
        P, A, d = self.P, self.num_articulators, self.embed_dim
        R = x.shape[1]
        n = f"dec{l}_"
        xhat = Normalize.apply(x)
        phat = NormalizeRes.apply(*self._blocks(l, "proc", xhat, xhat, tgt_mask, tgt_kpm, B), None)   # [A, R, d]
        if last_only:
            assert tgt_mask is None and tgt_kpm is None
            phat_q = phat.view(A, B, R // B, d)[:, :, -1].contiguous()                        # [A, B, d]: one query row per utterance
            R = B
        else:
            phat_q = phat
        cat = self._blocks(l, "inter", phat_q, phat, tgt_mask, tgt_kpm, B, cat=(A, A - 1))   # [A, R, (A-1) d]: concat over the others
        wl, bl = self._fold(P[n + "il_w"], P[n + "il_ln_w"], P[n + "il_ln_b"], P[n + "il_b"])
        inter = GroupedLinear.apply(NormalizeRes.apply(*cat, (A, A - 1)), wl, bl, tuple(range(A)), True)      # [A, R, d]
        inp = self._blocks(l, "input", Normalize.apply(inter), mem_hat, memory_mask, mem_kpm, B, kv=mem_kv)
        y = LayerNormAffine.apply(*inp, P[n + "ln2_w"], P[n + "ln2_b"])
        wf, bf = self._fold(P[n + "ff_w"][None], P[n + "ff_ln_w"][None], P[n + "ff_ln_b"][None], P[n + "ff_b"][None])
        ff = GroupedLinear.apply(Normalize.apply(y).view(1, A * R, d), wf, bf, (0,), True).view(A, R, d)
        return y + ff

    def _generate_one_step(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                           memory_key_padding_mask=None, memory_kv=None, last_only=False):
        """(bs, seq_len, num_channels, num_feat) -> (bs, seq_len, num_channels, 2, num_feat / 2)  (reference :430-474)."""
        P, A, d, nf = self.P, self.num_articulators, self.embed_dim, self.num_feat
        B, T = tgt.shape[:2]
        R = B * T
        train = self.training
        that = Normalize.apply(tgt.reshape(R * A, nf).float())
        w, b = self._fold(P["tgt_w"][None], P["tgt_ln_w"][None], P["tgt_ln_b"][None], P["tgt_b"][None])
        emb = GroupedLinear.apply(that[None], w, b, (0,), True).view(B, T, A, d)
        x = F.dropout(emb + self.pe[0, :T].view(1, T, 1, d), self.dropout, train)            # positional encoding per channel
        x = x.permute(2, 0, 1, 3).reshape(A, R, d).contiguous()   # channel-major (with T = 1 the reshape alone is a strided view)
        mem_hat = Normalize.apply(memory)[None] if memory_kv is None else None                 # shared by every cross block
        # ARTSPEECH_CHECKPOINT_LAYERS=1 (or model.checkpoint_layers = True): keep only each decoder layer's INPUT for the
        # backward and recompute the layer's forward there (every op is an autograd Function, so torch.utils.checkpoint
        # applies as is): about one sixth of the activation memory for one extra forward of the decoder
        ckpt = (getattr(self, "checkpoint_layers", False) or os.environ.get("ARTSPEECH_CHECKPOINT_LAYERS") == "1") and \
            torch.is_grad_enabled() and memory_kv is None and not last_only
        for l in range(self.num_layers):
            if ckpt:
                def run(x_, mem_, l=l):
                    return self._decoder_layer(l, x_, mem_, tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask, B)
                x = torch.utils.checkpoint.checkpoint(run, x, mem_hat, use_reentrant=False)
                continue
            x = self._decoder_layer(l, x, mem_hat, tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask, B,
                                    mem_kv=None if memory_kv is None else memory_kv[l],
                                    last_only=last_only and l == self.num_layers - 1)
        if last_only:  # x is [A, B, d]: the newest frame only
            T, R = 1, B
        feat = F.dropout(x.permute(1, 0, 2).reshape(R, A * d).contiguous(), self.dropout, train)
        w, b = self._fold(P["fin_w"][None], P["fin_ln_w"][None], P["fin_ln_b"][None], P["fin_b"][None])
        feat = GroupedLinear.apply(Normalize.apply(feat)[None], w, b, (0,), True)[0]
        out = Heads.apply(feat, P["head_flat"], self.head_dims, self.head_lay)
        return out.view(B, T, A, 2, nf // 2)

    # ------------------------------------------------------------------ public API
    def _check(self, src):
        _lib.require_gpu(src, "src")
        _lib.require_gpu(self.P["src_emb"], "model parameters")

    def _zero_padded(self):
        """True exactly when the reference's nn.TransformerEncoder would take its nested-tensor fast path: eval() mode
        and no gradient tracking (torch.no_grad(), or frozen parameters).  `set_encoder_grad_mode` overrides (tests)."""
        if self._grad_mode_hint is not None:
            return not self._grad_mode_hint
        return (not self.training) and not (torch.is_grad_enabled() and self.P["src_emb"].requires_grad)

    def set_encoder_grad_mode(self, grad_mode):
        self._grad_mode_hint = grad_mode

    def forward(self, src, tgt, src_attn_mask=None, tgt_attn_mask=None, memory_mask=None, src_key_padding_mask=None,
                tgt_key_padding_mask=None, memory_key_padding_mask=None):
        """src (bs, seq_len) int64, tgt (bs, seq_len, num_channels, num_feat); float masks as built by
        ``pad_sequence_transformer_collate_fn``.  As in the reference (:380-387) the cross-attention mask is
        ``src_attn_mask``; the ``memory_mask`` / ``memory_key_padding_mask`` arguments are not used by ``forward``."""
        self._check(src)
        f = lambda m: None if m is None else m.contiguous().float()  # noqa: E731
        memory = self._encode(src.long(), f(src_key_padding_mask), self._zero_padded())
        return self._generate_one_step(tgt, memory, tgt_mask=f(tgt_attn_mask), memory_mask=f(src_attn_mask),
                                       tgt_key_padding_mask=f(tgt_key_padding_mask), memory_key_padding_mask=None)

    def generate(self, src, src_key_padding_mask):
        """Autoregressive decoding exactly as the reference (:391-427): the encoder once, then seq_len full
        re-decodes of the growing prefix without target masks.  Returns (bs, seq_len, num_channels, 2, num_samples)."""
        self._check(src)
        with torch.no_grad():
            B, T = src.shape
            kpm = src_key_padding_mask.contiguous().float()
            memory = self._encode(src.long(), kpm, self._zero_padded())
            tgt = self.start.repeat(B, 1, 1, 1)
            # the only step-invariant part of the (unmasked, hence non-causal) re-decoding: the memory side of the cross blocks
            memory_kv = None
            self._fold_cache = {} if GENERATE_SAVINGS else None  # folded LayerNorm weights: once per call, not per frame
            try:
                if GENERATE_SAVINGS:
                    mem_hat = Normalize.apply(memory)[None]
                    memory_kv = [self._memory_kv(l, mem_hat) for l in range(self.num_layers)]
                for _ in range(T):
                    nxt = self._generate_one_step(tgt, memory, memory_key_padding_mask=kpm, memory_kv=memory_kv,
                                                  last_only=GENERATE_SAVINGS)
                    tgt = torch.cat([tgt, nxt[:, -1:].reshape(B, 1, self.num_articulators, self.num_feat)], dim=1)
            finally:
                self._fold_cache = None
            return tgt.reshape(B, T + 1, self.num_articulators, 2, self.num_feat // 2)[:, 1:]
