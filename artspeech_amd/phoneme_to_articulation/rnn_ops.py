"""Bidirectional recurrent layer (GRU or LSTM, packed-sequence semantics) as a ``torch.autograd.Function`` whose forward
and backward are C-ABI launches: input-projection GEMM -> persistent recurrence kernel; backward: recurrence kernel ->
input-gradient GEMM + time-batched weight-gradient GEMMs (dW_hh through the time-shifted operand of ``as_gemm_f32``).
Used by the models that choose the cell with the reference's ``RNNType`` switch (phoneme_to_articulation/__init__.py:47-49).
"""
import ctypes as C

import torch

from .. import _lib
from .transformer.ops import _c, _gemm, _slab

GATES = {"gru": 3, "lstm": 4}


def check_lengths(lengths, batch_size, padded_len):
    """The checks ``pack_padded_sequence(enforce_sorted=True)`` makes (same messages); returns (cpu int32 tensor, max)."""
    lengths_cpu = torch.as_tensor(lengths, dtype=torch.int32, device="cpu")
    if lengths_cpu.numel() != batch_size:
        raise RuntimeError(f"Expected `len(lengths)` to be equal to batch_size, but got {lengths_cpu.numel()} (batch_size={batch_size})")
    if lengths_cpu.numel() > 1 and bool((lengths_cpu[1:] > lengths_cpu[:-1]).any()):
        raise RuntimeError("`lengths` array must be sorted in decreasing order when `enforce_sorted` is True.")
    if int(lengths_cpu.min()) <= 0:
        raise RuntimeError("Length of all samples has to be greater than 0, but found an element in 'lengths' that is <= 0")
    T = int(lengths_cpu.max())
    if T > padded_len:
        raise RuntimeError(f"lengths.max()={T} exceeds the padded sequence length {padded_len}")
    return lengths_cpu, T


class BiRNNLayer(torch.autograd.Function):
    """x [B, T, I], w_ih [2, G*H, I], w_hh [2, G*H, H], b_ih / b_hh [2, G*H], lengths (int32, device) -> y [B, T, 2H]
    (forward direction in [:H], reverse in [H:], zeros at padded frames).  kind: "gru" (G = 3, rows r, z, n) or "lstm"
    (G = 4, rows i, f, g, o)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, lengths_dev, kind):
        x, w_ih, w_hh, b_ih, b_hh = _c(x), _c(w_ih), _c(w_hh), _c(b_ih), _c(b_hh)
        B, T, I = x.shape
        G = GATES[kind]
        H = w_hh.shape[2]
        L, st, dev = _lib.lib(), _lib.stream_ptr(), x.device
        gi = torch.empty((B * T, 2 * G * H), dtype=torch.float32, device=dev)
        _gemm(A=x, B=w_ih, C=gi, bias=b_ih, M=B * T, N=2 * G * H, K=I, a_i=I, a_k=1, b_j=I, b_k=1, ldc=2 * G * H)
        y = torch.empty((B, T, 2 * H), dtype=torch.float32, device=dev)
        train = any(ctx.needs_input_grad[:5])
        gates = torch.empty((B, T, 2, G + 1, H), dtype=torch.float32, device=dev) if train else None
        if kind == "gru":
            _lib.check(L.as_gru_bidir_fwd(_lib.ptr(gi), None, 0, _lib.ptr(w_hh), _lib.ptr(b_hh), _lib.ptr(lengths_dev), B, T, H, _lib.ptr(y),
                                          _lib.ptr(gates), st), "as_gru_bidir_fwd")
        else:
            _lib.check(L.as_lstm_bidir_fwd(_lib.ptr(gi), None, 0, _lib.ptr(w_hh), _lib.ptr(b_hh), _lib.ptr(lengths_dev), B, T, H, _lib.ptr(y),
                                           _lib.ptr(gates), st), "as_lstm_bidir_fwd")
        if train:
            ctx.save_for_backward(x, w_ih, w_hh, y, gates, lengths_dev)
            ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_ih, w_hh, y, gates, lengths_dev = ctx.saved_tensors
        kind = ctx.kind
        B, T, I = x.shape
        G = GATES[kind]
        H = w_hh.shape[2]
        R, W = B * T, G * H
        L, st, dev = _lib.lib(), _lib.stream_ptr(), x.device
        dy = _c(dy)
        dgi = torch.empty((R, 2 * W), dtype=torch.float32, device=dev)
        if kind == "gru":
            dgh = torch.empty_like(dgi)
            _lib.check(L.as_gru_bidir_bwd(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(gates), _lib.ptr(w_hh), _lib.ptr(lengths_dev), B, T, H,
                                          _lib.ptr(dgi), _lib.ptr(dgh), st), "as_gru_bidir_bwd")
        else:
            _lib.check(L.as_lstm_bidir_bwd(_lib.ptr(dy), _lib.ptr(gates), _lib.ptr(w_hh), _lib.ptr(lengths_dev), B, T, H, _lib.ptr(dgi), st),
                       "as_lstm_bidir_bwd")
            dgh = dgi
        slab = _slab(dev)
        dx = dw_ih = db_ih = dw_hh = db_hh = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _gemm(A=dgi, B=w_ih, C=dx, M=R, N=I, K=2 * W, a_i=2 * W, a_k=1, b_j=1, b_k=I, ldc=I)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[3]:
            dw_ih, db_ih = torch.empty_like(w_ih), torch.empty((2, W), dtype=torch.float32, device=dev)
            _gemm(A=dgi, B=x, C=dw_ih, M=2 * W, N=I, K=R, a_i=1, a_k=2 * W, b_j=1, b_k=I, ldc=I, colsum=db_ih, colsum_batch=0,
                  splitk_ws=slab, splitk_ws_floats=slab.numel())
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[4]:
            dw_hh, db_hh = torch.empty_like(w_hh), torch.empty((2, W), dtype=torch.float32, device=dev)
            for d in range(2):  # dW_hh = dgh^T . h_prev: the layer output shifted by one frame against the walk direction
                _gemm(A=dgh.data_ptr() + 4 * d * W, B=y.data_ptr() + 4 * d * H, C=dw_hh.data_ptr() + 4 * d * W * H, M=W, N=H, K=R, a_i=1,
                      a_k=2 * W, b_j=1, b_k=2 * H, ldc=H, colsum=db_hh.data_ptr() + 4 * d * W, colsum_batch=0, b_kshift=1 if d else -1,
                      b_kT=T, splitk_ws=slab, splitk_ws_floats=slab.numel())
        return dx, dw_ih, dw_hh, db_ih, db_hh, None, None


class Dropout(torch.autograd.Function):
    """Inverted dropout with the library's counter-based mask (the backward regenerates it from the seed)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = _c(x)
        y = torch.empty_like(x)
        _lib.check(_lib.lib().as_dropout_fwd(_lib.ptr(x), _lib.ptr(y), x.numel(), p, seed, _lib.stream_ptr()), "as_dropout_fwd")
        ctx.meta = (p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.meta
        dy = _c(dy)
        dx = torch.empty_like(dy)
        _lib.check(_lib.lib().as_dropout_fwd(_lib.ptr(dy), _lib.ptr(dx), dy.numel(), p, seed, _lib.stream_ptr()), "as_dropout_fwd")
        return dx, None, None


def birnn_stack(rnn, x, lengths_dev, kind, dropout, training):
    """Run the layers of an ``nn.GRU`` / ``nn.LSTM`` parameter container (bidirectional, batch_first) on the C ABI."""
    for layer in range(rnn.num_layers):
        p = [getattr(rnn, f"{name}_l{layer}{sfx}") for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh") for sfx in ("", "_reverse")]
        w_ih, w_hh, b_ih, b_hh = (torch.stack(p[2 * i:2 * i + 2]) for i in range(4))
        x = BiRNNLayer.apply(x, w_ih, w_hh, b_ih, b_hh, lengths_dev, kind)
        if training and dropout > 0.0 and layer + 1 < rnn.num_layers:
            x = Dropout.apply(x, float(dropout), int(torch.randint(0, 2 ** 62, (1,)).item()))
    return x
