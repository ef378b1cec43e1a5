"""Test-time metrics of the path on MI355X (reference: metrics.py at the repository root)."""
import torch

from .phoneme_to_articulation.metrics import EuclideanDistance, MeanP2CPDistance


def pearsons_correlation(outputs, targets):
    """Pearson correlation over time per (batch, articulator, point) for x and y (reference :9-35).

    NOTE (reproduced as is): the reference centres the x TARGETS with the mean of the x OUTPUTS
    (metrics.py:22); y is centred with its own mean (metrics.py:30)."""
    eps = 1e-5
    x_outputs, y_outputs = outputs[:, :, :, 0, :], outputs[:, :, :, 1, :]
    x_targets, y_targets = targets[:, :, :, 0, :], targets[:, :, :, 1, :]

    vx_outputs = x_outputs - x_outputs.mean(dim=1, keepdim=True)
    vx_targets = x_targets - x_outputs.mean(dim=1, keepdim=True)
    x_corr = torch.sum(vx_outputs * vx_targets, dim=1) / (
        torch.sqrt(torch.sum(vx_outputs ** 2, dim=1)) * torch.sqrt(torch.sum(vx_targets ** 2, dim=1)) + eps)

    vy_outputs = y_outputs - y_outputs.mean(dim=1, keepdim=True)
    vy_targets = y_targets - y_targets.mean(dim=1, keepdim=True)
    y_corr = torch.sum(vy_outputs * vy_targets, dim=1) / (
        torch.sqrt(torch.sum(vy_outputs ** 2, dim=1)) * torch.sqrt(torch.sum(vy_targets ** 2, dim=1)) + eps)
    return x_corr, y_corr


def p2cp_distance(outputs, targets):
    """(bs, seq_len, N_art, 2, N_samples) x2 -> P2CP (bs, seq_len, N_art)  (reference :38-52)."""
    return MeanP2CPDistance(reduction="none")(outputs.transpose(-1, -2), targets.transpose(-1, -2))


def euclidean_distance(outputs, targets):
    """(bs, seq_len, N_art, 2, N_samples) x2 -> mean Euclidean distance (bs, seq_len, N_art)  (reference :54-68)."""
    return EuclideanDistance(reduction="none")(outputs, targets).mean(dim=-1)
