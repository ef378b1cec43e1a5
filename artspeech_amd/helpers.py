"""Host helpers of the path (reference: helpers.py)."""
import random

import numpy as np
import torch


def set_seeds(worker_id):
    """DataLoader worker_init_fn (reference helpers.py:8-11)."""
    seed = torch.initial_seed() % 2 ** 31
    np.random.seed(seed + 1)
    random.seed(seed + 2)


def make_padding_mask(lengths):
    """Bool mask (B, max(lengths)), True on valid frames (reference helpers.py:79-91).

    lengths: tensor of shape (B,).  The mask is built on the device of ``lengths`` (CPU in the
    reference's training loop)."""
    max_length = int(lengths.max())
    steps = torch.arange(1, max_length + 1, device=lengths.device)
    return steps.unsqueeze(0) <= lengths.unsqueeze(1)
