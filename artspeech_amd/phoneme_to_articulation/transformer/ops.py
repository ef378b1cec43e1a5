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
PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16x6": 2}
GEMM_PRECISION = PRECISIONS[os.environ.get("ARTSPEECH_GEMM_PRECISION", "f32")]


def set_gemm_precision(name):
    global GEMM_PRECISION
    GEMM_PRECISION = PRECISIONS[name]


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


def _slab(dev):
    if dev not in _SLAB:
        _SLAB[dev] = torch.empty(8 << 20, dtype=torch.float32, device=dev)  # split-K partial tiles (32 MB)
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
                      colsum=db, colsum_batch=N)
            if identity:
                slab = _slab(x.device)
                kw.update(b_batch=R * K, splitk_ws=slab, splitk_ws_floats=slab.numel())
            else:
                kw["b_off"] = src_off
            _gemm(**kw)
        if ctx.needs_input_grad[0]:
            part = torch.empty((G, R, K), dtype=torch.float32, device=x.device)
            _gemm(A=dz, B=W, C=part, M=R, N=K, K=N, a_i=N, a_k=1, b_j=1, b_k=K, ldc=K, batch=G, a_batch=R * N, b_batch=N * K,
                  c_batch=R * K)
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
    _MASK_T[k] = (weakref.ref(attn_mask, lambda _r, k=k: _MASK_T.pop(k, None)), tag, mt)
    return mt


class Attention(torch.autograd.Function):
    """Multi-head attention core on projected tensors (nn.MultiheadAttention semantics, float additive masks):
    Q [G, B*T, d], K/V [G, B*Tk, d] -> ctx [G, B*T, d];  P = softmax(Q_h K_h^T / sqrt(dh) + attn_mask[b] + kpm[b])."""

    @staticmethod
    def forward(ctx, Q, K, V, attn_mask, kpm, B, heads):
        Q, K, V = _c(Q), _c(K), _c(V)
        G, R, d = Q.shape
        Rk = K.shape[1]
        T, Tk, dh = R // B, Rk // B, d // heads
        Z = G * B * heads
        dev = Q.device
        L = _lib.lib()
        scale = 1.0 / math.sqrt(dh)
        if FUSED_ATTENTION and L.as_attention_supported(T, Tk, d, heads):
            # scores stay in registers (as_attention_fwd); training additionally keeps the probabilities, key-major
            training = any(ctx.needs_input_grad[:3])
            out = torch.empty_like(Q)
            Pt = torch.empty((Z, Tk, T), dtype=torch.float32, device=dev) if training else None
            mt = _key_major_mask(attn_mask, Tk, T) if attn_mask is not None else None
            km = _c(kpm) if kpm is not None else None
            _lib.check(L.as_attention_fwd(_lib.ptr(Q), _lib.ptr(K), _lib.ptr(V), _lib.ptr(mt), _lib.ptr(km), _lib.ptr(out), None,
                                          _lib.ptr(Pt), G, B, heads, T, Tk, d, scale, _lib.stream_ptr()), "as_attention_fwd")
            if training:
                ctx.save_for_backward(Q, K, V, Pt, out)
                ctx.meta = (B, heads, scale, None, None, None)
            return out
        zq = _table(dev, ("zq", G, B, T, d, heads), lambda: [g * R * d + b * T * d + h * dh for g in range(G) for b in range(B) for h in range(heads)])
        zk = _table(dev, ("zq", G, B, Tk, d, heads), lambda: [g * Rk * d + b * Tk * d + h * dh for g in range(G) for b in range(B) for h in range(heads)])
        zs = _table(dev, ("zs", Z, T, Tk), lambda: [z * T * Tk for z in range(Z)])
        P = torch.empty((Z, T, Tk), dtype=torch.float32, device=dev)
        _gemm(A=Q, B=K, C=P, M=T, N=Tk, K=dh, a_i=d, a_k=1, b_j=d, b_k=1, ldc=Tk, batch=Z, a_off=zq, b_off=zk, c_off=zs)
        scale = 1.0 / math.sqrt(dh)
        am = _c(attn_mask) if attn_mask is not None else None
        km = _c(kpm) if kpm is not None else None
        _lib.check(_lib.lib().as_attn_softmax(_lib.ptr(P), Z, T, Tk, heads, B, scale, _lib.ptr(am), _lib.ptr(km), _lib.stream_ptr()),
                   "as_attn_softmax")
        out = torch.empty_like(Q)
        _gemm(A=P, B=V, C=out, M=T, N=dh, K=Tk, a_i=Tk, a_k=1, b_j=1, b_k=d, ldc=d, batch=Z, a_off=zs, b_off=zk, c_off=zq)
        ctx.save_for_backward(Q, K, V, P)
        ctx.meta = (B, heads, scale, zq, zk, zs)
        return out

    @staticmethod
    def backward(ctx, dctx):
        saved = ctx.saved_tensors   # read ONCE (torch.utils.checkpoint's unpack hooks allow a single access)
        if len(saved) == 5:
            return Attention._backward_key_major(ctx, dctx, saved)
        Q, K, V, P = saved
        B, heads, scale, zq, zk, zs = ctx.meta
        G, R, d = Q.shape
        Rk = K.shape[1]
        T, Tk, dh = R // B, Rk // B, d // heads
        Z = G * B * heads
        dctx = _c(dctx)
        dP = torch.empty_like(P)
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
        # dP = dctx V^T ; dV = P^T dctx
        _gemm(A=dctx, B=V, C=dP, M=T, N=Tk, K=dh, a_i=d, a_k=1, b_j=d, b_k=1, ldc=Tk, batch=Z, a_off=zq, b_off=zk, c_off=zs)
        _gemm(A=P, B=dctx, C=dV, M=Tk, N=dh, K=T, a_i=1, a_k=Tk, b_j=1, b_k=d, ldc=d, batch=Z, a_off=zs, b_off=zq, c_off=zk)
        _lib.check(_lib.lib().as_attn_softmax_bwd(_lib.ptr(P), _lib.ptr(dP), Z, T, Tk, scale, _lib.stream_ptr()), "as_attn_softmax_bwd")
        # dQ = dS K ; dK = dS^T Q
        _gemm(A=dP, B=K, C=dQ, M=T, N=dh, K=Tk, a_i=Tk, a_k=1, b_j=1, b_k=d, ldc=d, batch=Z, a_off=zs, b_off=zk, c_off=zq)
        _gemm(A=dP, B=Q, C=dK, M=Tk, N=dh, K=T, a_i=1, a_k=Tk, b_j=1, b_k=d, ldc=d, batch=Z, a_off=zs, b_off=zq, c_off=zk)
        return dQ, dK, dV, None, None, None, None


def _attention_backward_key_major(ctx, dctx, saved):
    """Backward of the fused forward: the same five products as the unfused path, on the KEY-major probabilities
    P^T [Z][Tk][T] that as_attention_fwd left (only the operand strides differ), D = rowsum(dctx * ctx) instead of a
    second pass over the scores."""
    Q, K, V, Pt, out = saved
    B, heads, scale = ctx.meta[:3]
    G, R, d = Q.shape
    Rk = K.shape[1]
    T, Tk, dh = R // B, Rk // B, d // heads
    Z = G * B * heads
    dev = Q.device
    zq = _table(dev, ("zq", G, B, T, d, heads), lambda: [g * R * d + b * T * d + h * dh for g in range(G) for b in range(B) for h in range(heads)])
    zk = _table(dev, ("zq", G, B, Tk, d, heads), lambda: [g * Rk * d + b * Tk * d + h * dh for g in range(G) for b in range(B) for h in range(heads)])
    zs = _table(dev, ("zs", Z, T, Tk), lambda: [z * T * Tk for z in range(Z)])
    dctx = _c(dctx)
    dPt = torch.empty_like(Pt)
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
    # dV = P^T dctx
    _gemm(A=Pt, B=dctx, C=dV, M=Tk, N=dh, K=T, a_i=T, a_k=1, b_j=1, b_k=d, ldc=d, batch=Z, a_off=zs, b_off=zq, c_off=zk)
    if FUSED_DS:
        # dS^T = P^T o (V dctx^T - D) * scale in one kernel: dP is never formed
        _lib.check(_lib.lib().as_attention_bwd_ds(_lib.ptr(V), _lib.ptr(dctx), _lib.ptr(out), _lib.ptr(Pt), _lib.ptr(dPt), G, B, heads, T,
                                                  Tk, d, scale, _lib.stream_ptr()), "as_attention_bwd_ds")
    else:
        _gemm(A=V, B=dctx, C=dPt, M=Tk, N=T, K=dh, a_i=d, a_k=1, b_j=d, b_k=1, ldc=T, batch=Z, a_off=zk, b_off=zq, c_off=zs)
        dsum = torch.empty((Z, T), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().as_attn_softmax_bwd_t(_lib.ptr(Pt), _lib.ptr(dPt), _lib.ptr(out), _lib.ptr(dctx), _lib.ptr(dsum), G, B,
                                                    heads, T, Tk, d, scale, _lib.stream_ptr()), "as_attn_softmax_bwd_t")
    # dQ = dS K (dS read through its transpose) ; dK = dS^T Q
    _gemm(A=dPt, B=K, C=dQ, M=T, N=dh, K=Tk, a_i=1, a_k=T, b_j=1, b_k=d, ldc=d, batch=Z, a_off=zs, b_off=zk, c_off=zq)
    _gemm(A=dPt, B=Q, C=dK, M=Tk, N=dh, K=T, a_i=T, a_k=1, b_j=1, b_k=d, ldc=d, batch=Z, a_off=zs, b_off=zq, c_off=zk)
    return dQ, dK, dV, None, None, None, None


Attention._backward_key_major = staticmethod(_attention_backward_key_major)


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
