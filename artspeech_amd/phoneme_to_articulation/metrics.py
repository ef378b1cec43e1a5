"""Losses of the path on MI355X (reference: phoneme_to_articulation/metrics.py).

``EuclideanDistance`` (:5-24) and ``MeanP2CPDistance`` (:27-46) keep the reference's constructor and
call signatures, including the ``reduction = getattr(torch, name, identity)`` convention ("none" ->
identity, "mean" -> torch.mean, ...).  ``masked_euclidean_loss`` is the fused form of the training
loop's criterion + padding mask + mean (train_phoneme_to_articulation.py:86-90).
"""
import torch
import torch.nn as nn

from .. import _lib


class _EuclidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, targets):
        L = _lib.lib()
        A, two, N = outputs.shape[-3:]
        frames = outputs.numel() // (A * 2 * N)
        o, t = outputs.contiguous(), targets.contiguous()
        dist = torch.empty((*outputs.shape[:-2], N), dtype=torch.float32, device=outputs.device)
        _lib.check(L.as_euclid_fwd(_lib.ptr(o), _lib.ptr(t), frames, A, N, _lib.ptr(dist), _lib.stream_ptr()), "as_euclid_fwd")
        ctx.save_for_backward(o, t)
        ctx.dims = (frames, A, N)
        return dist

    @staticmethod
    def backward(ctx, ddist):
        o, t = ctx.saved_tensors
        frames, A, N = ctx.dims
        L = _lib.lib()
        dout = torch.empty_like(o)
        ddist = ddist.contiguous()  # named: must stay alive until the launch is enqueued
        _lib.check(L.as_euclid_bwd(_lib.ptr(o), _lib.ptr(t), _lib.ptr(ddist), frames, A, N, _lib.ptr(dout),
                                   _lib.stream_ptr()), "as_euclid_bwd")
        return dout, (-dout if ctx.needs_input_grad[1] else None)


def _check_pair(outputs, targets):
    _lib.require_gpu(outputs, "outputs")
    _lib.require_gpu(targets, "targets")
    if outputs.shape != targets.shape or outputs.dim() < 3 or outputs.shape[-2] != 2:
        raise RuntimeError(f"expected two tensors of shape (..., N_art, 2, N_samples), got {tuple(outputs.shape)} "
                           f"and {tuple(targets.shape)}")
    if outputs.dtype != torch.float32 or targets.dtype != torch.float32:
        raise RuntimeError("artspeech_amd metrics compute in float32")


class EuclideanDistance(nn.Module):
    def __init__(self, reduction="mean"):
        super().__init__()
        self.reduction_name = reduction
        self.reduction = getattr(torch, reduction, lambda x: x)

    def forward(self, outputs, targets):
        """
        Args:
        outputs (torch.tensor): Torch tensor with shape (bs, seq_len, N_art, 2, N_samples).
        targets (torch.tensor): Torch tensor with shape (bs, seq_len, N_art, 2, N_samples).
        """
        _check_pair(outputs, targets)
        return self.reduction(_EuclidFn.apply(outputs, targets))


class _MaskedEuclidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, targets, lengths_dev, scale):
        L = _lib.lib()
        B, T, A, _, N = outputs.shape
        o, t = outputs.contiguous(), targets.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=o.device)
        need_grad = bool(ctx.needs_input_grad[0])
        dout = torch.empty_like(o) if need_grad else None
        partial = torch.empty(L.as_euclid_masked_partials(), dtype=torch.float32, device=o.device)
        _lib.check(L.as_euclid_masked_fwd_bwd(_lib.ptr(o), _lib.ptr(t), t.shape[1], _lib.ptr(lengths_dev), B, T, A, N,
                                              float(scale), _lib.ptr(loss), _lib.ptr(dout), _lib.ptr(partial),
                                              _lib.stream_ptr()), "as_euclid_masked_fwd_bwd")
        if need_grad:
            ctx.save_for_backward(dout)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dout,) = ctx.saved_tensors
        return dout * dloss, None, None, None


def masked_euclidean_loss(outputs, targets, lengths, n_valid_global=None):
    """mean over {valid frames} x articulators x points of the Euclidean distance -- the criterion,
    padding mask and mean of train_phoneme_to_articulation.py:86-90 in one kernel (its gradient comes
    out of the same pass).  targets may be padded longer than outputs (T_out = max(lengths)).
    n_valid_global: total number of valid frames of the GLOBAL batch when this rank holds a shard
    (data parallel): shard losses then SUM to the reference's global mean."""
    _check_pair(outputs[:, :1], targets[:, :1])
    lengths_cpu = torch.as_tensor(lengths, dtype=torch.int32, device="cpu")
    n_valid = int(lengths_cpu.sum()) if n_valid_global is None else int(n_valid_global)
    B, T, A, _, N = outputs.shape
    scale = 1.0 / (n_valid * A * N)
    return _MaskedEuclidFn.apply(outputs, targets, lengths_cpu.to(outputs.device, non_blocking=True), scale)


def _planar_strides(t):
    """(tile, point, xy) element strides of a (*, n, 2) tensor if its batch dims collapse to one
    uniform tile stride, else None."""
    n = t.shape[-2]
    if t.dim() == 2:
        return 0, t.stride(0), t.stride(1)
    lead_shape, lead_strides = t.shape[:-2], t.stride()[:-2]
    tile = lead_strides[-1]
    expect = tile
    for size, stride in zip(reversed(lead_shape), reversed(lead_strides)):
        if size != 1 and stride != expect:
            return None
        expect = stride * size if size != 1 else expect
    return tile, t.stride(-2), t.stride(-1)


def mean_p2cp(u_, v_):
    """MeanP2CPDistance, reduction "none": u_ (*, N, 2), v_ (*, M, 2) -> (*).  Transposed views of
    (*, 2, N) storage (how the reference calls it: metrics.py:47-50, encoder_decoder/metrics.py:19-22)
    are consumed in place through strides, no copy."""
    _lib.require_gpu(u_, "u_")
    _lib.require_gpu(v_, "v_")
    if u_.shape[-1] != 2 or v_.shape[-1] != 2 or u_.shape[:-2] != v_.shape[:-2]:
        raise RuntimeError(f"expected (*, N, 2) and (*, M, 2), got {tuple(u_.shape)} and {tuple(v_.shape)}")
    L = _lib.lib()
    su, sv = _planar_strides(u_), _planar_strides(v_)
    if su is None:
        u_ = u_.contiguous()
        su = _planar_strides(u_)
    if sv is None:
        v_ = v_.contiguous()
        sv = _planar_strides(v_)
    lead = u_.shape[:-2]
    tiles = 1
    for s in lead:
        tiles *= s
    out = torch.empty(lead, dtype=torch.float32, device=u_.device)
    with torch.no_grad():
        _lib.check(L.as_p2cp_fwd(_lib.ptr(u_), su[0], su[1], su[2], u_.shape[-2], _lib.ptr(v_), sv[0], sv[1], sv[2],
                                 v_.shape[-2], tiles, _lib.ptr(out), _lib.stream_ptr()), "as_p2cp_fwd")
    return out


class MeanP2CPDistance(nn.Module):
    def __init__(self, reduction="mean"):
        super().__init__()
        self.reduction = getattr(torch, reduction, lambda x: x)

    def forward(self, u_, v_):
        """
        Args:
        u_ (torch.tensor): Tensor of shape (*, N, 2)
        v_ (torch.tensor): Tensor of shape (*, M, 2)
        """
        return self.reduction(mean_p2cp(u_.detach(), v_.detach()))
