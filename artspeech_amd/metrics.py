"""Test-time metrics of the path on MI355X (reference: metrics.py at the repository root)."""
import torch

from . import _lib
from .phoneme_to_articulation.metrics import EuclideanDistance, MeanP2CPDistance


def pearsons_correlation(outputs, targets):
    """Pearson correlation over time per (batch, articulator, point) for x and y (reference :9-35), one HIP launch
    (``as_pearson_fwd``: a thread per column, two passes over the frames).

    NOTE (reproduced as is): the reference centres the x TARGETS with the mean of the x OUTPUTS
    (metrics.py:22); y is centred with its own mean (metrics.py:30)."""
    _lib.require_gpu(outputs, "outputs")
    _lib.require_gpu(targets, "targets")
    if outputs.shape != targets.shape or outputs.dim() != 5 or outputs.shape[3] != 2:
        raise ValueError(f"pearsons_correlation expects two (bs, seq_len, N_art, 2, N_samples) tensors, got "
                         f"{tuple(outputs.shape)} and {tuple(targets.shape)}")
    bs, seq_len, n_art, _, n_samples = targets.shape

    def frames(t):  # the (N_art, 2, N_samples) block of a frame must be contiguous; batch / time strides are free
        t = t.detach().float()
        ok = t.stride(4) == 1 and t.stride(3) == n_samples and t.stride(2) == 2 * n_samples
        return t if ok else t.contiguous()

    o, g = frames(outputs), frames(targets)
    x_corr = torch.empty((bs, n_art, n_samples), dtype=torch.float32, device=o.device)
    y_corr = torch.empty_like(x_corr)
    _lib.check(_lib.lib().as_pearson_fwd(_lib.ptr(o), o.stride(0), o.stride(1), _lib.ptr(g), g.stride(0), g.stride(1), bs, seq_len,
                                         n_art, n_samples, 1e-5, _lib.ptr(x_corr), _lib.ptr(y_corr), _lib.stream_ptr()),
               "as_pearson_fwd")
    return x_corr, y_corr


def p2cp_distance(outputs, targets):
    """(bs, seq_len, N_art, 2, N_samples) x2 -> P2CP (bs, seq_len, N_art)  (reference :38-52)."""
    return MeanP2CPDistance(reduction="none")(outputs.transpose(-1, -2), targets.transpose(-1, -2))


def euclidean_distance(outputs, targets):
    """(bs, seq_len, N_art, 2, N_samples) x2 -> mean Euclidean distance (bs, seq_len, N_art)  (reference :54-68)."""
    return EuclideanDistance(reduction="none")(outputs, targets).mean(dim=-1)
