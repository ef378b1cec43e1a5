"""Test loop of the model-free path on MI355X (reference: phoneme_to_articulation/encoder_decoder/evaluation.py:17-161).

Returns the same nested ``info`` dict (``loss`` + per articulator ``x_corr, y_corr, p2cp, p2cp_mm, med,
med_mm``).  All per-utterance metrics are computed on the GPU for the whole batch at once (the reference
moves everything to the CPU and loops over utterances and frames); only the final means cross to the host.
When the six tract-variable articulators are available (upper incisor injected from the reference
contour, :93-109) ``tract_variables.csv`` is written per sentence with the reference's column names
(phoneme_to_articulation/__init__.py:201-297), and the per-frame contour ``.npy`` dumps + ``phonemes.csv`` of ``save_outputs``
(:121-198; the B-spline regularised variant needs the external vt_tools package and raises).
"""
import csv
import os

import numpy as np
import torch

from ... import metrics as root_metrics
from ...tract_variables import REQUIRED_ARTICULATORS, TV_NAMES, UPPER_INCISOR, tract_variables_batched
from .. import save_outputs
from ..metrics import masked_euclidean_loss


def _write_tract_variables(save_to, sentences_ids, frame_ids, outputs, targets, lengths, phonemes, articulators):
    B, T = outputs.shape[:2]
    res = {}
    for key, tensor in (("pred", outputs), ("target", targets)):
        v, p1, p2, _ = tract_variables_batched(tensor.reshape(B * T, *tensor.shape[2:]), articulators)
        res[key] = (v.view(B, T, 4).cpu(), p1.view(B, T, 4, 2).cpu(), p2.view(B, T, 4, 2).cpu())
    for b, (sid, length) in enumerate(zip(sentences_ids, lengths)):
        sentence_dir = os.path.join(save_to, sid)
        os.makedirs(sentence_dir, exist_ok=True)
        rows = []
        for t in range(int(length)):
            item = {"sentence": sid, "frame": frame_ids[b][t], "phoneme": phonemes[b][t]}
            for key in ("target", "pred"):
                v, p1, p2 = res[key]
                for j, tv in enumerate(TV_NAMES):
                    item[f"{tv}_{key}"] = float(v[b, t, j])
                    item[f"{tv}_{key}_poc_1_x"], item[f"{tv}_{key}_poc_1_y"] = float(p1[b, t, j, 0]), float(p1[b, t, j, 1])
                    item[f"{tv}_{key}_poc_2_x"], item[f"{tv}_{key}_poc_2_y"] = float(p2[b, t, j, 0]), float(p2[b, t, j, 1])
            rows.append(item)
        with open(os.path.join(sentence_dir, "tract_variables.csv"), "w", newline="") as f:
            writer = csv.DictWriter(f, fieldnames=list(rows[0].keys()), lineterminator="\n")  # DataFrame.to_csv layout
            writer.writeheader()
            writer.writerows(rows)


class _Accumulator:
    """Per-articulator metric lists + file outputs shared by run_test and run_transformer_test."""

    def __init__(self, articulators, epoch_outputs_dir, device, regularize_out=False):
        self.articulators, self.dir, self.device = list(articulators), epoch_outputs_dir, device
        self.regularize_out = regularize_out
        n = len(self.articulators)
        self.losses = []
        self.euclid, self.p2cp = [[] for _ in range(n)], [[] for _ in range(n)]
        self.x_corrs, self.y_corrs = [[] for _ in range(n)], [[] for _ in range(n)]

    def add(self, loss, outputs, targets, lengths, sentences_ids, sentence_frames, phonemes, reference_arrays):
        device, arts = self.device, self.articulators
        with torch.no_grad():
            p2cp_bta = root_metrics.p2cp_distance(outputs, targets)            # (B, T, A)
            med_bta = root_metrics.euclidean_distance(outputs, targets)        # (B, T, A)
        self.losses.append(float(loss))
        for b, length in enumerate(lengths):
            length = int(length)
            pv, mv = p2cp_bta[b, :length].cpu().numpy(), med_bta[b, :length].cpu().numpy()
            # NOTE as the reference (:68-74): .mean(dim=1) collapses the frames of the single utterance
            xc, yc = root_metrics.pearsons_correlation(outputs[b:b + 1, :length], targets[b:b + 1, :length])
            xc, yc = xc.mean(dim=-1)[0].cpu().numpy(), yc.mean(dim=-1)[0].cpu().numpy()
            for i in range(len(arts)):
                self.x_corrs[i].append(float(xc[i]))
                self.y_corrs[i].append(float(yc[i]))
                self.p2cp[i].append(float(pv[:, i].mean()))
                self.euclid[i].append(float(mv[:, i].mean()))
        # upper incisor = reference of the coordinate system: injected for the tract variables (:93-109)
        if UPPER_INCISOR not in arts:
            tv_articulators = sorted(arts + [UPPER_INCISOR])
            ref_idx = tv_articulators.index(UPPER_INCISOR)
            ref = reference_arrays[:, :outputs.shape[1]].to(device)
            outputs = torch.cat([outputs[:, :, :ref_idx], ref, outputs[:, :, ref_idx:]], dim=2)
            targets = torch.cat([targets[:, :, :ref_idx], ref, targets[:, :, ref_idx:]], dim=2)
        else:
            tv_articulators = arts
        if all(a in tv_articulators for a in REQUIRED_ARTICULATORS) and outputs.shape[-1] >= 50:
            _write_tract_variables(self.dir, sentences_ids, sentence_frames, outputs, targets, lengths, phonemes, tv_articulators)
        save_outputs(sentences_ids, sentence_frames, outputs, targets, lengths, phonemes, tv_articulators, self.dir, self.regularize_out)

    def info(self, dataset_config):
        to_mm = dataset_config.RES * dataset_config.PIXEL_SPACING
        info = {"loss": float(np.mean(self.losses))}
        info.update({
            art: {
                "x_corr": float(np.mean(self.x_corrs[i])), "y_corr": float(np.mean(self.y_corrs[i])),
                "p2cp": float(np.mean(self.p2cp[i])), "p2cp_mm": float(np.mean(self.p2cp[i]) * to_mm),
                "med": float(np.mean(self.euclid[i])), "med_mm": float(np.mean(self.euclid[i]) * to_mm),
            }
            for i, art in enumerate(self.articulators)
        })
        return info


def run_test(epoch, model, dataloader, criterion, outputs_dir, articulators, device=None, regularize_out=False):
    if device is None:
        device = torch.device("cuda")
    epoch_outputs_dir = os.path.join(outputs_dir, str(epoch))
    os.makedirs(epoch_outputs_dir, exist_ok=True)
    model.eval()
    acc = _Accumulator(articulators, epoch_outputs_dir, device, regularize_out)
    for sentences_ids, sentences, targets, lengths, phonemes, reference_arrays, sentence_frames, _ in dataloader:
        sentences, targets = sentences.to(device), targets.to(device)
        with torch.no_grad():
            outputs = model(sentences, lengths)
            targets = targets[:, :outputs.shape[1]]
            loss = masked_euclidean_loss(outputs, targets, lengths)  # criterion + padding mask + mean (:56-63)
        acc.add(loss.item(), outputs, targets, lengths, sentences_ids, sentence_frames, phonemes, reference_arrays)
    return acc.info(dataloader.dataset.dataset_config)
