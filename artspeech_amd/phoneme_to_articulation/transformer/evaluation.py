"""Test loop of the transformer variant (reference: phoneme_to_articulation/transformer/evaluation.py:19-191): free-running
``model.generate`` under no_grad, utterances whose prediction contains NaN are reported and left out (:69-86), then the
same per-utterance metrics, upper-incisor injection and tract-variable CSVs as the model-free test loop."""
import os

import torch

from ..encoder_decoder.evaluation import _Accumulator
from ..metrics import masked_euclidean_loss


def run_transformer_test(epoch, model, dataloader, criterion, outputs_dir, articulators, device=None, regularize_out=False):
    if device is None:
        device = torch.device("cuda")
    epoch_outputs_dir = os.path.join(outputs_dir, str(epoch))
    os.makedirs(epoch_outputs_dir, exist_ok=True)
    model.eval()
    acc = _Accumulator(articulators, epoch_outputs_dir, device, regularize_out)
    for (sentences_ids, sentences, targets, lengths, phonemes, reference_arrays, sentence_frames, _, src_key_padding_mask, _, _,
         _) in dataloader:
        sentences, targets = sentences.to(device), targets.to(device)
        with torch.no_grad():
            outputs = model.generate(sentences, src_key_padding_mask=src_key_padding_mask.to(device))
        nan = torch.isnan(outputs).flatten(1).any(dim=1).cpu()
        keep = [i for i in range(outputs.shape[0]) if not bool(nan[i])]
        if not keep:
            continue
        if len(keep) < outputs.shape[0]:
            bad = "\n".join(sentences_ids[i] for i in range(outputs.shape[0]) if bool(nan[i]))
            print(f"Invalid outputs produced for sentences:\n{bad}\n")
        idx = torch.tensor(keep, device=device)
        outputs, targets = outputs[idx], targets[idx]
        reference_arrays = reference_arrays[keep]
        # the loss sees the kept utterances with their own lengths (reference :81-86: the padding mask is filtered) ...
        loss = masked_euclidean_loss(outputs.contiguous(), targets.contiguous(), lengths[keep])
        # ... but the metrics and the files do not: the reference zips the KEPT outputs with the UNFILTERED lengths, sentence
        # ids, frames and phonemes (:96, :146-168), so kept utterance j is measured over lengths[j] frames and reported under
        # sentences_ids[j] of the whole batch.  Kept as it is (pinned by tests/golden/transformer_loops.npz): lengths are
        # sorted in decreasing order, so lengths[j] never exceeds the padded length of a kept prediction.
        n = len(keep)
        acc.add(loss.item(), outputs, targets, lengths[:n], list(sentences_ids[:n]), list(sentence_frames[:n]), list(phonemes[:n]),
                reference_arrays)
    return acc.info(dataloader.dataset.dataset_config)
