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


def sequences_from_dict(datadir, sequences_dict):
    """{subject: [sequence, ...]} -> [(subject, sequence)]; an empty list selects every sequence
    directory of the subject (reference helpers.py:63-76)."""
    import os
    sequences = []
    for subj, seqs in sequences_dict.items():
        use_seqs = seqs
        if len(seqs) == 0:
            use_seqs = [s for s in sorted(os.listdir(os.path.join(datadir, subj)))
                        if os.path.isdir(os.path.join(datadir, subj, s))]
        sequences.extend([(subj, seq) for seq in use_seqs])
    return sequences


def make_indices_dict(num_components):
    """{articulator: n_components} -> {articulator: [latent indices]}, consecutive ranges in insertion order
    (reference helpers.py:94-114): {'a': 3, 'b': 2} -> {'a': [0, 1, 2], 'b': [3, 4]}."""
    indices, start = {}, 0
    for name, count in num_components.items():
        indices[name] = list(range(start, start + count))
        start += count
    return indices


def npy_to_xarticul(array, filepath=None):
    """(N, 2) coordinates -> the point list of an Xarticul contour file: one "x y" line per point and the terminating
    "-1 -1" (reference helpers.py:27-45); written to ``filepath`` when given.  Returns the list of lines."""
    lines = [f"{x} {y}" for x, y in array]
    lines.append("-1 -1")  # end-of-contour tag of Xarticul
    if filepath is not None:
        with open(filepath, "w") as f:
            f.write("\n".join(lines))
    return lines


def xarticul_to_npy(filepath):
    """Xarticul contour file -> (N, 2) float64 array; the last line ("-1 -1") is the end tag (reference helpers.py:48-60)."""
    with open(filepath, "r") as f:
        rows = [line.strip().split() for line in f.readlines()][:-1]
    return np.array([[float(v) for v in row] for row in rows])
