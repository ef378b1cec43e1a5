"""Validation metric of the model-free path (reference: phoneme_to_articulation/encoder_decoder/metrics.py)."""
import torch
import torch.nn as nn

from ... import _lib
from ..metrics import mean_p2cp


class P2CPDistance(nn.Module):
    """P2CP in millimetres: per-utterance mean over valid frames and articulators, then the mean over
    the batch (reference :7-26).  Returns a CPU scalar tensor like the reference does."""

    def __init__(self, dataset_config):
        super().__init__()
        self.dataset_config = dataset_config
        self.to_mm = self.dataset_config.RES * self.dataset_config.PIXEL_SPACING

    def forward(self, outputs, targets, lengths):
        L = _lib.lib()
        B, T, A = outputs.shape[:3]
        targets = targets[:, :T]
        p2cp = mean_p2cp(outputs.detach().transpose(-1, -2), targets.detach().transpose(-1, -2)).contiguous()  # (B, T, A)
        lengths_dev = torch.as_tensor(lengths, dtype=torch.int32, device="cpu").to(outputs.device)
        result = torch.empty(1, dtype=torch.float32, device=outputs.device)
        _lib.check(L.as_p2cp_utterance_mean(_lib.ptr(p2cp), _lib.ptr(lengths_dev), B, T, A, float(self.to_mm),
                                            _lib.ptr(result), _lib.stream_ptr()), "as_p2cp_utterance_mean")
        return result[0].cpu()
