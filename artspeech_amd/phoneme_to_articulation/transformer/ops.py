"""Differentiable building blocks of the transformer variant: ``torch.autograd.Function`` wrappers whose forward AND
backward run on the C ABI (grouped fp32-MFMA GEMMs, masked softmax, LayerNorm kernels).  PyTorch's autograd engine only
wires them together (and adds gradients where a tensor has several consumers).

Layout convention: per-block tensors are block-major ``[G][rows][features]`` (contiguous); a grouped linear reads
channel ``src[g]`` of a channel-major input ``[C][rows][K]``.
"""
import ctypes as C
import math
import os
import weakref

import torch

from ... import _lib

_TABLES = {}
_SLAB = {}

# Precision of the forward linears of the modules built on these ops (as_gemm.precision): "f32" = exact fp32 MFMA,
# "bf16x6" / "bf16x3" = fp32 operands split on the fly into 3 / 2 bf16 pieces on the bf16 MFMA, fp32 accumulation.
# "lib" = the library's matrix arithmetic (as_set_matrix_arith: split fp32 on the bf16 matrix instruction unless the process chose
# the exact fp32 one) for the forward linears, at any size.  Measured at configs[3] (d=256, L=6, B=32, T=200): forward + backward
# 186.8 -> 172.7 ms; the full-width model's contours against the REFERENCE fixture then sit at 1.19 x the 1e-4 bound (0.89 x with
# "f32": two correct fp32 roundings of a 6-layer network differ by about that much), so the default stays "f32": by default only
# the backward's GEMMs (input and weight gradients, GRAD_PRECISION below) use the split arithmetic.
PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16x6": 2, "lib": 3}
GEMM_PRECISION = PRECISIONS[os.environ.get("ARTSPEECH_GEMM_PRECISION", "f32")]


# Precision of the backward's GEMMs -- input gradients (dx = dz W, with their residual / ReLU-mask / segmented-reduction operands)
# and weight gradients (dW = dz^T x with the bias gradient as fused column sums): the library's matrix arithmetic by default
# (configs[3]: 218 ms all-fp32 -> 186.8 ms).  The parity tests hold gradients to a yardstick relative to their own magnitude and the split
# product is at least as accurate as the fp32 instruction's (tests/test_gpu_parity.py::test_split_matrix_arithmetic_error_vs_fp64),
# so nothing a forward value is compared with depends on this.  ARTSPEECH_GRAD_PRECISION=f32 keeps them on the fp32 instruction.
GRAD_PRECISION = PRECISIONS[os.environ.get("ARTSPEECH_GRAD_PRECISION", "lib")]


def set_gemm_precision(name, grad=None):
    global GEMM_PRECISION, GRAD_PRECISION
    GEMM_PRECISION = PRECISIONS[name]
    if grad is not None:
        GRAD_PRECISION = PRECISIONS[grad]


def _gemm(**kw):
    g = _lib.Gemm()
    g.batch = 1
    g.precision = GEMM_PRECISION
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    _lib.check(_lib.lib().as_gemm_f32(C.byref(g), _lib.stream_ptr()), "as_gemm_f32")


def _table(dev, key, build):
    k = (dev, key)
    if k not in _TABLES:
        _TABLES[k] = torch.tensor(build(), dtype=torch.int64, device=dev)
    return _TABLES[k]


def _table32(dev, key, build):
    k = (dev, key)
    if k not in _TABLES:
        _TABLES[k] = torch.tensor(build(), dtype=torch.int32, device=dev)
    return _TABLES[k]


def _slab(dev):
    if dev not in _SLAB:
        _SLAB[dev] = torch.empty(20 << 20, dtype=torch.float32, device=dev)  # split-K slabs / stream-K pieces (80 MB)
    return _SLAB[dev]


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class FoldLN(torch.autograd.Function):
    """(W [G,R,K], gamma [G,K], beta [G,K], b [G,R]) -> (W.diag(gamma), b + W.beta): Linear(LayerNorm(x)) == x_hat Wf^T + bf."""

    @staticmethod
    def forward(ctx, W, gamma, beta, b):
        W, gamma, beta, b = _c(W), _c(gamma), _c(beta), _c(b)
        G, R, K = W.shape
        Wf, bf = torch.empty_like(W), torch.empty_like(b)
        _lib.check(_lib.lib().as_fold_ln(_lib.ptr(W), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(b), _lib.ptr(Wf), _lib.ptr(bf), G, R, K,
                                         _lib.stream_ptr()), "as_fold_ln")
        ctx.save_for_backward(W, gamma, beta)
        return Wf, bf

    @staticmethod
    def backward(ctx, dWf, dbf):
        W, gamma, beta = ctx.saved_tensors
        G, R, K = W.shape
        dWf, dbf = _c(dWf), _c(dbf)
        dW, dgamma, dbeta = torch.empty_like(W), torch.empty_like(gamma), torch.empty_like(beta)
        _lib.check(_lib.lib().as_unfold_ln(_lib.ptr(dWf), _lib.ptr(dbf), _lib.ptr(W), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(dW),
                                           _lib.ptr(dgamma), _lib.ptr(dbeta), G, R, K, _lib.stream_ptr()), "as_unfold_ln")
        return dW, dgamma, dbeta, dbf


class Normalize(torch.autograd.Function):
    """Affine-free LayerNorm over the last dim (eps 1e-5): x -> x_hat."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        D = x.shape[-1]
        rows = x.numel() // D
        xhat = torch.empty_like(x)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().as_layernorm_fwd(_lib.ptr(x), None, None, None, None, _lib.ptr(xhat), _lib.ptr(rstd), rows, D, 0,
                                               _lib.stream_ptr()), "as_layernorm_fwd")
        ctx.save_for_backward(xhat, rstd)
        return xhat

    @staticmethod
    def backward(ctx, dxhat):
        xhat, rstd = ctx.saved_tensors
        D = xhat.shape[-1]
        dxhat = _c(dxhat)
        dx = torch.empty_like(xhat)
        _lib.check(_lib.lib().as_layernorm_bwd(_lib.ptr(dxhat), _lib.ptr(xhat), _lib.ptr(rstd), None, _lib.ptr(dx), xhat.numel() // D, D,
                                               _lib.stream_ptr()), "as_layernorm_bwd")
        return dx


class NormalizeRes(torch.autograd.Function):
    """x_hat = affine-free LayerNorm(x + res) -- the LayerNorm that consumes a block group's output adds the group's residual
    (the projected queries, transformer/models.py:98), in the reference's order (sum + bias) + q.  cat = None: res has x's
    layout.  cat = (A, per): x is the concatenated [A, R, per * d] output of the group and res the block-major
    [A, per, R, d] queries (ChannelBlocks)."""

    @staticmethod
    def forward(ctx, x, res, cat):
        x, res = _c(x), _c(res)
        D = x.shape[-1]
        rows = x.numel() // D
        xhat = torch.empty_like(x)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        if cat is None:
            _lib.check(_lib.lib().as_layernorm_fwd(_lib.ptr(x), _lib.ptr(res), None, None, None, _lib.ptr(xhat), _lib.ptr(rstd), rows, D, 0,
                                                   _lib.stream_ptr()), "as_layernorm_fwd")
        else:
            A_, per = cat
            _lib.check(_lib.lib().as_layernorm_fwd_blockres(_lib.ptr(x), _lib.ptr(res), _lib.ptr(xhat), _lib.ptr(rstd), A_, rows // A_, per,
                                                            D // per, _lib.stream_ptr()), "as_layernorm_fwd_blockres")
        ctx.save_for_backward(xhat, rstd)
        ctx.cat = cat
        return xhat

    @staticmethod
    def backward(ctx, dxhat):
        xhat, rstd = ctx.saved_tensors
        D = xhat.shape[-1]
        dxhat = _c(dxhat)
        dx = torch.empty_like(xhat)
        _lib.check(_lib.lib().as_layernorm_bwd(_lib.ptr(dxhat), _lib.ptr(xhat), _lib.ptr(rstd), None, _lib.ptr(dx), xhat.numel() // D, D,
                                               _lib.stream_ptr()), "as_layernorm_bwd")
        if ctx.cat is None:
            return dx, dx, None
        # the residual's gradient is the same tensor seen block-major: a strided [A, per, R, d] view, no copy (ChannelBlocks
        # recognises it and reads the gradient in place)
        A_, per = ctx.cat
        return dx, dx.view(A_, xhat.shape[1], per, D // per).permute(0, 2, 1, 3), None


class LayerNormAffine(torch.autograd.Function):
    """y = LayerNorm(x + res) * gamma + beta (res optional) -- the post-norm sites whose output is used directly."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta):
        x = _c(x)
        res = _c(res) if res is not None else None
        D = x.shape[-1]
        rows = x.numel() // D
        y, xhat = torch.empty_like(x), torch.empty_like(x)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().as_layernorm_fwd(_lib.ptr(x), _lib.ptr(res), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y), _lib.ptr(xhat),
                                               _lib.ptr(rstd), rows, D, 0, _lib.stream_ptr()), "as_layernorm_fwd")
        ctx.save_for_backward(xhat, rstd, gamma)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, rstd, gamma = ctx.saved_tensors
        D = xhat.shape[-1]
        dy = _c(dy)
        flat_dy, flat_xh = dy.reshape(-1, D), xhat.reshape(-1, D)
        dgamma, dbeta = (flat_dy * flat_xh).sum(0), flat_dy.sum(0)  # two small column reductions (glue)
        dxhat = dy * gamma
        dx = torch.empty_like(xhat)
        _lib.check(_lib.lib().as_layernorm_bwd(_lib.ptr(dxhat), _lib.ptr(xhat), _lib.ptr(rstd), None, _lib.ptr(dx), xhat.numel() // D, D,
                                               _lib.stream_ptr()), "as_layernorm_bwd")
        return dx, (dx if ctx.has_res else None), dgamma, dbeta


class GroupedLinear(torch.autograd.Function):
    """out[g] = act(x[src[g]] W[g]^T + b[g]):  x [C, R, K], W [G, N, K], b [G, N] -> out [G, R, N]; act = ReLU if relu."""

    @staticmethod
    def forward(ctx, x, W, b, src, relu):
        x, W, b = _c(x), _c(W), _c(b)
        Cc, R, K = x.shape
        G, N, _ = W.shape
        src = tuple(int(s) for s in src)
        identity = Cc == G and src == tuple(range(G))
        out = torch.empty((G, R, N), dtype=torch.float32, device=x.device)
        kw = dict(A=x, B=W, C=out, bias=b, M=R, N=N, K=K, a_i=K, a_k=1, b_j=K, b_k=1, ldc=N, batch=G, b_batch=N * K, c_batch=R * N,
                  bias_batch=N, act=1 if relu else 0)
        if identity:
            kw["a_batch"] = R * K
        else:
            kw["a_off"] = _table(x.device, ("src", src, R * K), lambda: [s * R * K for s in src])
        _gemm(**kw)
        ctx.save_for_backward(x, W, out if relu else None)
        ctx.meta = (src, identity, relu)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, W, out = ctx.saved_tensors
        src, identity, relu = ctx.meta
        Cc, R, K = x.shape
        G, N, _ = W.shape
        L, st = _lib.lib(), _lib.stream_ptr()
        dz = _c(dout)
        if relu:
            dzr = torch.empty_like(dz)
            _lib.check(L.as_relu_bwd(_lib.ptr(dz), _lib.ptr(out), _lib.ptr(dzr), dz.numel(), st), "as_relu_bwd")
            dz = dzr
        dW = db = dx = None
        src_off = None if identity else _table(x.device, ("src", src, R * K), lambda: [s * R * K for s in src])
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dW = torch.empty_like(W)
            db = torch.empty((G, N), dtype=torch.float32, device=x.device)
            kw = dict(A=dz, B=x, C=dW, M=N, N=K, K=R, a_i=1, a_k=N, b_j=1, b_k=K, ldc=K, batch=G, a_batch=R * N, c_batch=N * K,
                      colsum=db, colsum_batch=N, precision=GRAD_PRECISION)
            if identity:
                slab = _slab(x.device)
                kw.update(b_batch=R * K, splitk_ws=slab, splitk_ws_floats=slab.numel())
            else:
                kw["b_off"] = src_off
            _gemm(**kw)
        if ctx.needs_input_grad[0]:
            part = torch.empty((G, R, K), dtype=torch.float32, device=x.device)
            _gemm(A=dz, B=W, C=part, M=R, N=K, K=N, a_i=N, a_k=1, b_j=1, b_k=K, ldc=K, batch=G, a_batch=R * N, b_batch=N * K,
                  c_batch=R * K, precision=GRAD_PRECISION)
            if identity:
                dx = part
            else:
                dx = torch.empty_like(x)
                srct = torch.tensor(src, dtype=torch.int32, device=x.device)
                _lib.check(L.as_group_reduce(_lib.ptr(part), _lib.ptr(srct), G, Cc, R * K, _lib.ptr(dx), st), "as_group_reduce")
        return dx, dW, db, None, None


FUSED_DS = os.environ.get("ARTSPEECH_UNFUSED_DS") is None  # ablation: dP GEMM + as_attn_softmax_bwd_t instead of as_attention_bwd_ds
FUSED_ATTENTION = os.environ.get("ARTSPEECH_UNFUSED_ATTENTION") is None  # ablation switch (tools/bench_attention.py)
_MASK_T = {}  # id(mask tensor) -> (weak reference to it, (version, Tk, T), key-major copy)


def _key_major_mask(attn_mask, Tk, T):
    """(B, T, Tk) additive mask -> (B, Tk32, T): key-major, keys zero-padded to a multiple of 32 (as_attention_fwd).
    The same mask tensor serves every attention call of a forward pass, so the transpose is kept -- keyed on the tensor
    OBJECT (weak reference + version counter), never on its address: a mask that has been freed takes its entry with it,
    and a new mask that happens to get the same storage address (or the same id) can never hit a stale entry."""
    k = id(attn_mask)
    tag = (attn_mask._version, Tk, T)
    hit = _MASK_T.get(k)
    if hit is not None and hit[0]() is attn_mask and hit[1] == tag:
        return hit[2]
    B = attn_mask.shape[0]
    mt = torch.zeros((B, (Tk + 31) // 32 * 32, T), dtype=torch.float32, device=attn_mask.device)
    mt[:, :Tk] = attn_mask.to(torch.float32).transpose(1, 2)
    # causal: every key AFTER the query is masked with -inf for every utterance -> P[q][k] == 0 exactly for k > q (one check per
    # mask tensor; the backward's GEMMs over P^T / dS^T then skip the k-tiles that hold nothing but those zeros, as_gemm.k_tri)
    upper = torch.triu(torch.ones(T, Tk, dtype=torch.bool, device=attn_mask.device), diagonal=1)
    mt.causal = bool(torch.isneginf(attn_mask.to(torch.float32))[:, upper].all().item()) if bool(upper.any()) else False
    _MASK_T[k] = (weakref.ref(attn_mask, lambda _r, k=k: _MASK_T.pop(k, None)), tag, mt)
    return mt


def _z_tables(dev, G, B, T, Tk, d, heads):
    R, Rk, dh, Z = B * T, B * Tk, d // heads, G * B * heads
    zq = _table(dev, ("zq", G, B, T, d, heads), lambda: [g * R * d + b * T * d + h * dh for g in range(G) for b in range(B) for h in range(heads)])
    zk = _table(dev, ("zq", G, B, Tk, d, heads), lambda: [g * Rk * d + b * Tk * d + h * dh for g in range(G) for b in range(B) for h in range(heads)])
    zs = _table(dev, ("zs", Z, T, Tk), lambda: [z * T * Tk for z in range(Z)])
    return zq, zk, zs


def attention_forward(Q, K, V, attn_mask, kpm, B, heads, training):
    """Multi-head attention core on projected tensors (nn.MultiheadAttention semantics, float additive masks):
    Q [G, B*T, d], K/V [G, B*Tk, d] -> ctx [G, B*T, d];  P = softmax(Q_h K_h^T / sqrt(dh) + attn_mask[b] + kpm[b]).
    Returns (ctx, tensors to keep for attention_backward or None, scale, causal) -- causal: the mask is -inf above the diagonal
    for every utterance, i.e. the probabilities there are exact zeros (what attention_backward's GEMMs then skip)."""
    G, R, d = Q.shape
    Rk = K.shape[1]
    T, Tk, dh = R // B, Rk // B, d // heads
    Z = G * B * heads
    dev = Q.device
    L = _lib.lib()
    scale = 1.0 / math.sqrt(dh)
    km = _c(kpm) if kpm is not None else None
    if FUSED_ATTENTION and L.as_attention_supported(T, Tk, d, heads):
        # scores stay in registers (as_attention_fwd); training additionally keeps the probabilities, key-major
        out = torch.empty_like(Q)
        # key-major probabilities with rows padded to a multiple of 32 floats: a strip's 128-byte segment of a row is then one
        # cache line for the forward's stores, the dS kernel and the backward GEMMs (at T = 200 the natural 800-byte pitch
        # makes every segment straddle two)
        Tp = (T + 31) // 32 * 32
        Pt = torch.empty((Z, Tk, Tp), dtype=torch.float32, device=dev) if training else None
        mt = _key_major_mask(attn_mask, Tk, T) if attn_mask is not None else None
        causal = mt is not None and bool(getattr(mt, "causal", False))
        fwd = L.as_attention_fwd_causal if causal else L.as_attention_fwd
        _lib.check(fwd(_lib.ptr(Q), _lib.ptr(K), _lib.ptr(V), _lib.ptr(mt), _lib.ptr(km), _lib.ptr(out), None, _lib.ptr(Pt), G, B, heads,
                       T, Tk, d, scale, _lib.stream_ptr()), "as_attention_fwd")
        return out, ((Q, K, V, Pt, out) if training else None), scale, causal
    zq, zk, zs = _z_tables(dev, G, B, T, Tk, d, heads)
    P = torch.empty((Z, T, Tk), dtype=torch.float32, device=dev)
    _gemm(A=Q, B=K, C=P, M=T, N=Tk, K=dh, a_i=d, a_k=1, b_j=d, b_k=1, ldc=Tk, batch=Z, a_off=zq, b_off=zk, c_off=zs)
    am = _c(attn_mask) if attn_mask is not None else None
    _lib.check(L.as_attn_softmax(_lib.ptr(P), Z, T, Tk, heads, B, scale, _lib.ptr(am), _lib.ptr(km), _lib.stream_ptr()), "as_attn_softmax")
    out = torch.empty_like(Q)
    _gemm(A=P, B=V, C=out, M=T, N=dh, K=Tk, a_i=Tk, a_k=1, b_j=1, b_k=d, ldc=d, batch=Z, a_off=zs, b_off=zk, c_off=zq)
    return out, ((Q, K, V, P) if training else None), scale, False


def attention_backward(saved, B, heads, scale, dctx, dQ=None, dK=None, dV=None, causal=False):
    """(dQ, dK, dV) of attention_forward; the three may be handed in (contiguous slices of a caller's buffer).  causal: the
    forward's additive mask was -inf above the diagonal for every utterance (attention_forward found out)."""
    Q, K, V = saved[:3]
    G, R, d = Q.shape
    Rk = K.shape[1]
    T, Tk, dh = R // B, Rk // B, d // heads
    Z = G * B * heads
    dev = Q.device
    zq, zk, zs = _z_tables(dev, G, B, T, Tk, d, heads)
    dctx = _c(dctx)
    dQ = torch.empty_like(Q) if dQ is None else dQ
    dK = torch.empty_like(K) if dK is None else dK
    dV = torch.empty_like(V) if dV is None else dV
    if len(saved) == 5:
        # backward of the fused forward: the same five products as the unfused path, on the KEY-major probabilities
        # P^T [Z][Tk][Tp] that as_attention_fwd left (only the operand strides differ), D = rowsum(dctx * ctx) instead of a
        # second pass over the scores
        Pt, out = saved[3:]
        dPt = torch.empty_like(Pt)
        Tp = Pt.shape[2]   # row pitch of the key-major tensors: T rounded up to 32 (attention_forward)
        zp = _table(dev, ("zp", Z, Tk, Tp), lambda: [z * Tk * Tp for z in range(Z)])
        # causal mask: P^T[key][q] and dS^T[key][q] are exact zeros for q < key -- the three products below skip those k-tiles
        lower = dict(k_tri=1, precision=0) if causal else {}   # reduction over q, rows = keys: zero for q < key
        upper = dict(k_tri=2, precision=0) if causal else {}   # reduction over keys, rows = q: zero for key > q
        # dV = P^T dctx
        _gemm(A=Pt, B=dctx, C=dV, M=Tk, N=dh, K=T, a_i=Tp, a_k=1, b_j=1, b_k=d, ldc=d, batch=Z, a_off=zp, b_off=zq, c_off=zk, **lower)
        if FUSED_DS:
            # dS^T = P^T o (V dctx^T - D) * scale in one kernel: dP is never formed
            ds_fn = _lib.lib().as_attention_bwd_ds_causal if causal else _lib.lib().as_attention_bwd_ds
            _lib.check(ds_fn(_lib.ptr(V), _lib.ptr(dctx), _lib.ptr(out), _lib.ptr(Pt), _lib.ptr(dPt), G, B, heads, T, Tk, d, scale,
                             _lib.stream_ptr()), "as_attention_bwd_ds")
        else:
            _gemm(A=V, B=dctx, C=dPt, M=Tk, N=T, K=dh, a_i=d, a_k=1, b_j=d, b_k=1, ldc=Tp, batch=Z, a_off=zk, b_off=zq, c_off=zp)
            dsum = torch.empty((Z, T), dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().as_attn_softmax_bwd_t(_lib.ptr(Pt), _lib.ptr(dPt), _lib.ptr(out), _lib.ptr(dctx), _lib.ptr(dsum), G, B,
                                                        heads, T, Tk, d, scale, _lib.stream_ptr()), "as_attn_softmax_bwd_t")
        # dQ = dS K (dS read through its transpose) ; dK = dS^T Q
        _gemm(A=dPt, B=K, C=dQ, M=T, N=dh, K=Tk, a_i=1, a_k=Tp, b_j=1, b_k=d, ldc=d, batch=Z, a_off=zp, b_off=zk, c_off=zq, **upper)
        _gemm(A=dPt, B=Q, C=dK, M=Tk, N=dh, K=T, a_i=Tp, a_k=1, b_j=1, b_k=d, ldc=d, batch=Z, a_off=zp, b_off=zq, c_off=zk, **lower)
        return dQ, dK, dV
    P = saved[3]
    dP = torch.empty_like(P)
    # dP = dctx V^T ; dV = P^T dctx
    _gemm(A=dctx, B=V, C=dP, M=T, N=Tk, K=dh, a_i=d, a_k=1, b_j=d, b_k=1, ldc=Tk, batch=Z, a_off=zq, b_off=zk, c_off=zs)
    _gemm(A=P, B=dctx, C=dV, M=Tk, N=dh, K=T, a_i=1, a_k=Tk, b_j=1, b_k=d, ldc=d, batch=Z, a_off=zs, b_off=zq, c_off=zk)
    _lib.check(_lib.lib().as_attn_softmax_bwd(_lib.ptr(P), _lib.ptr(dP), Z, T, Tk, scale, _lib.stream_ptr()), "as_attn_softmax_bwd")
    # dQ = dS K ; dK = dS^T Q
    _gemm(A=dP, B=K, C=dQ, M=T, N=dh, K=Tk, a_i=Tk, a_k=1, b_j=1, b_k=d, ldc=d, batch=Z, a_off=zs, b_off=zk, c_off=zq)
    _gemm(A=dP, B=Q, C=dK, M=Tk, N=dh, K=T, a_i=1, a_k=Tk, b_j=1, b_k=d, ldc=d, batch=Z, a_off=zs, b_off=zq, c_off=zk)
    return dQ, dK, dV


class Attention(torch.autograd.Function):
    """attention_forward / attention_backward as an autograd node (the encoder layers; the decoder's blocks go through
    ChannelBlocks)."""

    @staticmethod
    def forward(ctx, Q, K, V, attn_mask, kpm, B, heads):
        out, saved, scale, causal = attention_forward(_c(Q), _c(K), _c(V), attn_mask, kpm, B, heads, any(ctx.needs_input_grad[:3]))
        if saved is not None:
            ctx.save_for_backward(*saved)
        ctx.meta = (B, heads, scale)
        ctx.causal = causal
        return out

    @staticmethod
    def backward(ctx, dctx):
        saved = ctx.saved_tensors   # read ONCE (torch.utils.checkpoint's unpack hooks allow a single access)
        return (*attention_backward(saved, *ctx.meta, dctx, causal=ctx.causal), None, None, None, None)


def _ptr(t, off_floats=0):
    return t.data_ptr() + 4 * off_floats


def channel_blocks_forward(xt, xs, q_w, q_b, k_w, k_b, v_w, v_b, in_w, in_b, o_w, o_b, ln_w, ln_b, attn_mask, kpm, cfg, training,
                           kv2=None):
    """Forward of ChannelBlocks (see there).  kv2 = (k2, v2) [G, Rs, d] each: the in-projected key / value side computed
    elsewhere (generate(): the memory side of the cross-attention blocks, once per call) -- inference only; xs is unused
    then.  Returns (out, q, tensors for the backward or None, meta): out without the residual, q the projected queries (the
    residual a consuming LayerNorm adds), block-major [G, R, d] or, with cat, [A, per, R, d]."""
    tgt, src, B, heads, cat = cfg
    xt = _c(xt)
    q_w, q_b, in_w, in_b, o_w, o_b, ln_w, ln_b = (_c(p) for p in (q_w, q_b, in_w, in_b, o_w, o_b, ln_w, ln_b))
    G, d = q_w.shape[0], q_w.shape[-1]
    R = xt.shape[1]
    dev = xt.device
    L, st = _lib.lib(), _lib.stream_ptr()
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
    assert kv2 is None or not training
    if kv2 is None:
        xs, k_w, k_b, v_w, v_b = (_c(p) for p in (xs, k_w, k_b, v_w, v_b))
        Rs = xs.shape[1]
        folds = ((q_w, q_b), (k_w, k_b), (v_w, v_b))
    else:
        Rs = kv2[0].shape[1]
        folds = ((q_w, q_b),)
    # LayerNorm affine folded into the three pre-projections
    W3, b3 = new(3, G, d, d), new(3, G, d)
    for j, (w, b) in enumerate(folds):
        _lib.check(L.as_fold_ln(_lib.ptr(w), _lib.ptr(ln_w), _lib.ptr(ln_b), _lib.ptr(b), _ptr(W3, j * G * d * d), _ptr(b3, j * G * d), G,
                                d, d, st), "as_fold_ln")
    same = R == Rs and kv2 is None
    pre = new(3, G, R, d) if same else None          # q, k, v (after the ReLU), one buffer when the row counts agree
    q = pre[0] if same else new(G, R, d)
    t_off = _table(dev, ("src", tgt, R * d), lambda: [s * R * d for s in tgt])
    # training: the ReLUs leave their bit images (1 bit per element) for the backward's masks
    ncb = (d + 31) // 32
    qbits = torch.empty((G, R, ncb), dtype=torch.int32, device=dev) if training else None
    kvbits = torch.empty((2 * G, Rs, ncb), dtype=torch.int32, device=dev) if training else None
    lib_or_exact = 3 if GEMM_PRECISION == 3 else 0   # (the on-the-fly split kernels of as_gemm.precision 1 / 2 have no bit-image epilogue)
    bq = dict(relu_bits=qbits, relu_bits_batch=R * ncb, precision=lib_or_exact) if training else {}
    bkv = dict(relu_bits=kvbits, relu_bits_batch=Rs * ncb, precision=lib_or_exact) if training else {}
    _gemm(A=xt, B=W3, C=q, bias=b3, M=R, N=d, K=d, a_i=d, a_k=1, b_j=d, b_k=1, ldc=d, batch=G, a_off=t_off, b_batch=d * d,
          c_batch=R * d, bias_batch=d, act=1, **bq)
    # MHA in-projection: slice j of the stacked [G, 3d, d] weight, read in place
    w_off = _table(dev, ("inw", G, d), lambda: [g * 3 * d * d + j * d * d for j in range(3) for g in range(G)])
    bi_off = _table(dev, ("inb", G, d), lambda: [g * 3 * d + j * d for j in range(3) for g in range(G)])
    p2 = new(3, G, R, d) if same else None
    q2 = p2[0] if same else new(G, R, d)
    kv = None
    if kv2 is None:
        kv = pre[1:] if same else new(2, G, Rs, d)
        s_off2 = _table(dev, ("src2", src, Rs * d), lambda: [s * Rs * d for s in src] * 2)
        _gemm(A=xs, B=_ptr(W3, G * d * d), C=kv, bias=_ptr(b3, G * d), M=Rs, N=d, K=d, a_i=d, a_k=1, b_j=d, b_k=1, ldc=d, batch=2 * G,
              a_off=s_off2, b_batch=d * d, c_batch=Rs * d, bias_batch=d, act=1, **bkv)
        kv2 = p2[1:] if same else new(2, G, Rs, d)
    if same:
        _gemm(A=pre, B=in_w, C=p2, bias=in_b, M=R, N=d, K=d, a_i=d, a_k=1, b_j=d, b_k=1, ldc=d, batch=3 * G, a_batch=R * d,
              b_off=w_off, c_batch=R * d, bias_off=bi_off)
    else:
        _gemm(A=q, B=in_w, C=q2, bias=in_b, M=R, N=d, K=d, a_i=d, a_k=1, b_j=d, b_k=1, ldc=d, batch=G, a_batch=R * d,
              b_batch=3 * d * d, c_batch=R * d, bias_batch=3 * d)
        if kv is not None:
            _gemm(A=kv, B=in_w, C=kv2, bias=in_b, M=Rs, N=d, K=d, a_i=d, a_k=1, b_j=d, b_k=1, ldc=d, batch=2 * G, a_batch=Rs * d,
                  b_off=w_off[G:], c_batch=Rs * d, bias_off=bi_off[G:])
    att, att_saved, scale, causal = attention_forward(q2, _c(kv2[0]), _c(kv2[1]), attn_mask, kpm, B, heads, training)
    # out-projection, straight into its final layout.  The residual `q + out` (:98) is NOT added here: every group's output
    # goes into a LayerNorm only, which adds q as it reads (NormalizeRes / LayerNormAffine) -- no pass of its own, and the
    # reference's order (sum + bias) + q.  (As the accumulators' initial value it costs the GEMM 110 us per 110-block launch
    # and widens the forward's rounding noise: the full-width contours left the 1e-4 band; fetched by the GEMM's epilogue 450 us.)
    if cat is not None:
        A_, per = cat
        assert A_ * per == G
        out = new(A_, R, per * d)
        ldo = per * d
        c_off = _table(dev, ("cat", A_, per, R, d), lambda: [c * R * per * d + j * d for c in range(A_) for j in range(per)])
        lay = dict(c_off=c_off)
        q_out = q.view(A_, per, R, d)
    else:
        out = new(G, R, d)
        ldo = d
        lay = dict(c_batch=R * d)
        q_out = q
    _gemm(A=att, B=o_w, C=out, bias=o_b, M=R, N=d, K=d, a_i=d, a_k=1, b_j=d, b_k=1, ldc=ldo, batch=G, a_batch=R * d, b_batch=d * d,
          bias_batch=d, **lay)
    if not training:
        return out, q_out, None, None
    return (out, q_out, (xt, xs, q_w, k_w, v_w, ln_w, ln_b, W3, in_w, o_w, q, kv, att, qbits, kvbits, *att_saved),
            (tgt, src, B, heads, cat, scale, ldo, causal))


class ChannelBlocks(torch.autograd.Function):
    """One GROUP of ChannelProcessingLayers (reference transformer/models.py:37-100; the A self blocks, the A*(A-1)
    interaction blocks or the A cross-attention blocks of a decoder layer, :165-277) as ONE autograd node with a
    hand-written backward:

        q = relu(LN(x_tgt) Wq^T + bq), k / v likewise from x_src      (the LayerNorm is shared: :70-76; its affine is folded
        q2, k2, v2 = in_proj(q, k, v); ctx = attention(q2, k2, v2)      into the three weights, x_* arrive normalised)
        out = ctx Wo^T + bo ;  returns (out, q)                        (the residual `q + out`, :98, is added by the LayerNorm
                                                                        that consumes the pair: NormalizeRes / LayerNormAffine)

    xt [Ct, R, d] / xs [Cs, Rs, d] are channel-major, block g reads channels tgt[g] / src[g].  cat = (A, per): the output
    is written as [A, R, per * d] -- block g = c * per + j lands in columns [j*d, (j+1)*d) of channel c, which is the
    concatenation over the other channels that ChannelInteractionsLayer builds (:133-162), q then is [A, per, R, d] -- else
    both are block-major [G, R, d].

    What the fusion buys over one autograd node per Linear (measured): no pass for the residual add (it rides
    in the next LayerNorm's read); in the backward the residual's gradient starts the accumulators of the
    in-projection's input-gradient GEMM, whose epilogue applies the ReLU masks from 1-bit images the forward GEMMs left
    (no pass over the [G, R, d] activations in between), the k and v
    sides run as one batch of 2G, the in-projection reads and writes the stacked [G, 3d, d] weight / gradient in place, and
    the per-channel input gradients are sums over the blocks of a channel INSIDE one segmented GEMM (as_gemm.k_seg)."""

    @staticmethod
    def forward(ctx, xt, xs, q_w, q_b, k_w, k_b, v_w, v_b, in_w, in_b, o_w, o_b, ln_w, ln_b, attn_mask, kpm, cfg):
        # cfg[5]: the caller's torch.is_grad_enabled() -- inside forward() autograd is always off, and the parameters keep
        # requires_grad under no_grad(), so needs_input_grad alone cannot tell an inference pass (which then would write the
        # attention probabilities and the ReLU bit images for nothing)
        training = bool(cfg[5]) and any(ctx.needs_input_grad)
        cfg = cfg[:5]
        out, q_out, saved, meta = channel_blocks_forward(xt, xs, q_w, q_b, k_w, k_b, v_w, v_b, in_w, in_b, o_w, o_b, ln_w, ln_b, attn_mask,
                                                         kpm, cfg, training)
        if training:
            ctx.save_for_backward(*saved)
            ctx.meta = meta
        return out, q_out

    @staticmethod
    def backward(ctx, dout, dq_res):
        saved = ctx.saved_tensors   # read ONCE (torch.utils.checkpoint's unpack hooks allow a single access)
        xt, xs, q_w, k_w, v_w, ln_w, ln_b, W3, in_w, o_w, q, kv, att, qbits, kvbits = saved[:15]
        att_saved = saved[15:]
        tgt, src, B, heads, cat, scale, ldo, causal = ctx.meta
        G, d = q_w.shape[0], q_w.shape[-1]
        ncb = (d + 31) // 32
        (Ct, R, _), (Cs, Rs, _) = xt.shape, xs.shape
        dev = xt.device
        L, st = _lib.lib(), _lib.stream_ptr()
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
        slab = _slab(dev)
        ws = dict(splitk_ws=slab, splitk_ws_floats=slab.numel())
        dout = _c(dout)
        if cat is not None:
            A_, per = cat
            c_off = _table(dev, ("cat", A_, per, R, d), lambda: [c * R * per * d + j * d for c in range(A_) for j in range(per)])
            a_lay = dict(a_off=c_off)
            in_place = (A_, per, R, d), (R * per * d, d, per * d, 1)
        else:
            a_lay = dict(a_batch=R * d)
            in_place = (G, R, d), (R * d, d, 1)
        # the gradient over the residual (q is the group's second output): normally the consuming LayerNorm's dx seen in q's
        # layout, i.e. `dout`'s own storage -- read in place as the initial value of the query gradient's accumulators
        if dq_res is None:
            r_lay = {}
        elif dq_res.data_ptr() == dout.data_ptr() and (tuple(dq_res.shape), tuple(dq_res.stride())) == in_place:
            r_lay = dict(res=dout, res_ld=ldo, res_off=c_off) if cat is not None else dict(res=dout, res_ld=d, res_batch=R * d)
        else:
            dq_res = dq_res.contiguous().view(G, R, d)
            r_lay = dict(res=dq_res, res_ld=d, res_batch=R * d)
        same = R == Rs
        # ---- out-projection: d ctx = dout Wo ; dWo = dout^T ctx (+ its bias gradient as the column sums of dout)
        datt = new(G, R, d)
        _gemm(A=dout, B=o_w, C=datt, M=R, N=d, K=d, a_i=ldo, a_k=1, b_j=1, b_k=d, ldc=d, batch=G, b_batch=d * d, c_batch=R * d,
              precision=GRAD_PRECISION, **a_lay)
        do_w, do_b = new(G, d, d), new(G, d)
        _gemm(A=dout, B=att, C=do_w, M=d, N=d, K=R, a_i=1, a_k=ldo, b_j=1, b_k=d, ldc=d, batch=G, b_batch=R * d, c_batch=d * d,
              colsum=do_b, colsum_batch=d, precision=GRAD_PRECISION, **a_lay, **ws)
        # ---- attention core
        dp2 = new(3, G, R, d) if same else None
        dq2 = dp2[0] if same else new(G, R, d)
        dkv2 = dp2[1:] if same else new(2, G, Rs, d)
        attention_backward(att_saved, B, heads, scale, datt, dQ=dq2, dK=dkv2[0], dV=dkv2[1], causal=causal)
        del datt
        # ---- in-projection: weight gradients into the stacked [G, 3d, d] tensor; input gradients through the ReLU masks of
        # q / k / v, the query's together with the gradient that arrives over the residual (`dout` itself)
        din_w, din_b = new(G, 3 * d, d), new(G, 3 * d)
        for j, (dz, y, rows) in enumerate(((dq2, q, R), (dkv2[0], kv[0], Rs), (dkv2[1], kv[1], Rs))):
            _gemm(A=dz, B=y, C=_ptr(din_w, j * d * d), M=d, N=d, K=rows, a_i=1, a_k=d, b_j=1, b_k=d, ldc=d, batch=G, a_batch=rows * d,
                  b_batch=rows * d, c_batch=3 * d * d, colsum=_ptr(din_b, j * d), colsum_batch=3 * d, precision=GRAD_PRECISION, **ws)
        dq, dkv = new(G, R, d), new(2, G, Rs, d)
        w_off = _table(dev, ("inw", G, d), lambda: [g * 3 * d * d + j * d * d for j in range(3) for g in range(G)])
        _gemm(A=dq2, B=in_w, C=dq, M=R, N=d, K=d, a_i=d, a_k=1, b_j=1, b_k=d, ldc=d, batch=G, a_batch=R * d, b_batch=3 * d * d,
              c_batch=R * d, mask_bits=qbits, mask_batch=R * ncb, precision=GRAD_PRECISION, **r_lay)
        _gemm(A=dkv2, B=in_w, C=dkv, M=Rs, N=d, K=d, a_i=d, a_k=1, b_j=1, b_k=d, ldc=d, batch=2 * G, a_batch=Rs * d, b_off=w_off[G:],
              c_batch=Rs * d, mask_bits=kvbits, mask_batch=Rs * ncb, precision=GRAD_PRECISION)
        del dp2, dq2, dkv2
        # ---- pre-projections: gradients of the folded weights, unfolded onto (W, gamma, beta)
        t_off = _table(dev, ("src", tgt, R * d), lambda: [s * R * d for s in tgt])
        s_off2 = _table(dev, ("src2", src, Rs * d), lambda: [s * Rs * d for s in src] * 2)
        dWf, dbf = new(3, G, d, d), new(3, G, d)
        _gemm(A=dq, B=xt, C=dWf, M=d, N=d, K=R, a_i=1, a_k=d, b_j=1, b_k=d, ldc=d, batch=G, a_batch=R * d, b_off=t_off, c_batch=d * d,
              colsum=dbf, colsum_batch=d, precision=GRAD_PRECISION, **ws)
        _gemm(A=dkv, B=xs, C=_ptr(dWf, G * d * d), M=d, N=d, K=Rs, a_i=1, a_k=d, b_j=1, b_k=d, ldc=d, batch=2 * G, a_batch=Rs * d,
              b_off=s_off2, c_batch=d * d, colsum=_ptr(dbf, G * d), colsum_batch=d, precision=GRAD_PRECISION, **ws)
        dW3, dg3, db3 = new(3, G, d, d), new(3, G, d), new(3, G, d)
        for j, w in enumerate((q_w, k_w, v_w)):
            _lib.check(L.as_unfold_ln(_ptr(dWf, j * G * d * d), _ptr(dbf, j * G * d), _lib.ptr(w), _lib.ptr(ln_w), _lib.ptr(ln_b),
                                      _ptr(dW3, j * G * d * d), _ptr(dg3, j * G * d), _ptr(db3, j * G * d), G, d, d, st), "as_unfold_ln")
        dln_w, dln_b = dg3.sum(0), db3.sum(0)   # the three projections share the block's LayerNorm
        # ---- input gradients per channel: sums over the blocks that read the channel
        dxt = _channel_sums(dq.view(G, R, d), W3[0], tgt, Ct, R, d) if ctx.needs_input_grad[0] else None
        dxs = _channel_sums(dkv.view(2 * G, Rs, d), W3[1:].reshape(2 * G, d, d), tuple(src) * 2, Cs, Rs, d) if ctx.needs_input_grad[1] else None
        return (dxt, dxs, dW3[0], dbf[0], dW3[1], dbf[1], dW3[2], dbf[2], din_w, din_b, do_w, do_b, dln_w, dln_b, None, None, None)


def _channel_sums(dz, W, idx, Cn, R, d):
    """dx[c] = sum over the blocks g with idx[g] == c of dz[g] W[g]  (dz [N, R, d], W [N, d, d], dx [Cn, R, d])."""
    N = len(idx)
    dev = dz.device
    dx = torch.empty((Cn, R, d), dtype=torch.float32, device=dev)
    if N == Cn and tuple(idx) == tuple(range(N)):
        _gemm(A=dz, B=W, C=dx, M=R, N=d, K=d, a_i=d, a_k=1, b_j=1, b_k=d, ldc=d, batch=N, a_batch=R * d, b_batch=d * d, c_batch=R * d,
              precision=GRAD_PRECISION)
        return dx
    blocks = [[g for g in range(N) if idx[g] == c] for c in range(Cn)]
    per = len(blocks[0])
    if per > 0 and all(len(b) == per for b in blocks) and d % 32 == 0 and d % 4 == 0:
        # one GEMM per channel over the concatenated reduction (per * d), segment s = block blocks[c][s]
        a_seg = _table(dev, ("aseg", tuple(idx), Cn, R * d), lambda: [g * R * d for b in blocks for g in b])
        b_seg = _table(dev, ("bseg", tuple(idx), Cn, d * d), lambda: [g * d * d for b in blocks for g in b])
        _gemm(A=dz, B=W, C=dx, M=R, N=d, K=per * d, a_i=d, a_k=1, b_j=1, b_k=d, ldc=d, batch=Cn, c_batch=R * d, k_seg=d, a_seg_off=a_seg,
              b_seg_off=b_seg, precision=GRAD_PRECISION)
        return dx
    part = torch.empty((N, R, d), dtype=torch.float32, device=dev)
    _gemm(A=dz, B=W, C=part, M=R, N=d, K=d, a_i=d, a_k=1, b_j=1, b_k=d, ldc=d, batch=N, a_batch=R * d, b_batch=d * d, c_batch=R * d,
          precision=GRAD_PRECISION)
    srct = _table32(dev, ("idx32", tuple(idx)), lambda: list(idx))
    _lib.check(_lib.lib().as_group_reduce(_lib.ptr(part), _lib.ptr(srct), N, Cn, R * d, _lib.ptr(dx), _lib.stream_ptr()), "as_group_reduce")
    return dx


class Heads(torch.autograd.Function):
    """The A stacked ArticulatorPredictor heads + sigmoid (encoder_decoder/models.py:7-33, 141-145) on feat [R, d]."""

    @staticmethod
    def forward(ctx, feat, head_flat, dims, lay):
        L = _lib.lib()
        feat = _c(feat)
        R = feat.shape[0]
        out = torch.empty((R, dims.n_art, 2, dims.n_samp), dtype=torch.float32, device=feat.device)
        ws = torch.empty(L.as_head_workspace_floats(C.byref(dims), R), dtype=torch.float32, device=feat.device)
        _lib.check(L.as_head_fwd(C.byref(dims), C.byref(lay), _lib.ptr(head_flat), _lib.ptr(feat), R, _lib.ptr(out), _lib.ptr(ws), 1,
                                 _lib.stream_ptr()), "as_head_fwd")
        ctx.save_for_backward(head_flat, out, ws)
        ctx.meta = (dims, lay, feat.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        head_flat, out, ws = ctx.saved_tensors
        dims, lay, shape = ctx.meta
        L = _lib.lib()
        dout = _c(dout)
        dx = torch.empty(shape, dtype=torch.float32, device=out.device)
        grads = torch.zeros_like(head_flat)
        _lib.check(L.as_head_bwd(C.byref(dims), C.byref(lay), _lib.ptr(head_flat), _lib.ptr(out), _lib.ptr(dout), shape[0], _lib.ptr(dx),
                                 _lib.ptr(grads), _lib.ptr(ws), _lib.stream_ptr()), "as_head_bwd")
        return dx, grads, None, None
