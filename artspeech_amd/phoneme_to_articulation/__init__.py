"""phoneme_to_articulation package of the MI355X engine (reference: phoneme_to_articulation/__init__.py)."""
import csv
import os
from enum import Enum

import numpy as np
import torch.nn as nn


class RNNType(Enum):
    """Recurrent cell switch (reference phoneme_to_articulation/__init__.py:47-49).  The values are the torch classes only
    because the reference's are (they serve as PARAMETER CONTAINERS here: same keys, shapes and initialisation); the
    recurrences run on the HIP kernels of csrc/gru.hip and csrc/lstm.hip."""
    LSTM = nn.LSTM
    GRU = nn.GRU


def save_outputs(sentences_ids, frame_ids, outputs, targets, lengths, phonemes, articulators, save_to, regularize_out=False):
    """Per-sentence contour dumps of the test loops (reference phoneme_to_articulation/__init__.py:121-198): for every valid
    frame and articulator (sorted names) ``<save_to>/<sentence>/contours/<frame>_<articulator>.npy`` (prediction, (2, N)
    float32) and ``..._true.npy`` (target), plus ``phonemes.csv`` (sentence, frame, phoneme).  ``outputs`` / ``targets``
    (bs, seq_len, n_articulators, 2, n_samples) are copied to the host once per batch.  ``regularize_out`` asks for the
    B-spline regularisation of the un-vendored ``vt_tools`` package (:178-180), which this build does not restate."""
    if regularize_out:
        raise NotImplementedError("regularize_out=True needs vt_tools.bs_regularization.regularize_Bsplines (external package)")
    outputs = outputs.detach().cpu().numpy() if hasattr(outputs, "detach") else np.asarray(outputs)
    targets = targets.detach().cpu().numpy() if hasattr(targets, "detach") else np.asarray(targets)
    names = sorted(articulators)
    for b, (sentence_id, length) in enumerate(zip(sentences_ids, lengths)):
        contour_dir = os.path.join(save_to, sentence_id, "contours")
        os.makedirs(contour_dir, exist_ok=True)
        rows = []
        for t, (phoneme, frame) in enumerate(zip(phonemes[b], frame_ids[b])):
            if t >= int(length):
                break
            rows.append((sentence_id, frame, phoneme))
            for i_art, art in enumerate(names):
                np.save(os.path.join(contour_dir, f"{frame}_{art}.npy"), outputs[b, t, i_art])
                np.save(os.path.join(contour_dir, f"{frame}_{art}_true.npy"), targets[b, t, i_art])
        with open(os.path.join(save_to, sentence_id, "phonemes.csv"), "w", newline="") as f:
            writer = csv.writer(f, lineterminator="\n")  # pandas.DataFrame.to_csv(index=False) layout
            writer.writerow(("sentence", "frame", "phoneme"))
            writer.writerows(rows)


REQUIRED_ARTICULATORS_FOR_TVS = ["lower-lip", "pharynx", "soft-palate-midline", "tongue", "upper-lip", "upper-incisor"]  # :25-32


def tract_variables(sentences_ids, frame_ids, outputs, targets, lengths, phonemes, articulators, save_to):
    """``<save_to>/<sentence>/tract_variables.csv`` with the reference's columns (phoneme_to_articulation/__init__.py:
    201-297); the tract variables of all frames of the batch come from one launch of the HIP kernel."""
    from .encoder_decoder.evaluation import _write_tract_variables
    return _write_tract_variables(save_to, sentences_ids, frame_ids, outputs, targets, lengths, phonemes, articulators)
