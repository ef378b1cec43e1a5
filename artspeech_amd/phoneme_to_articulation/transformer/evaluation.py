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
        lengths_k = lengths[keep]
        # the kept utterances stay sorted by length; the masked loss only needs lengths <= T
        loss = masked_euclidean_loss(outputs.contiguous(), targets.contiguous(), lengths_k)
        acc.add(loss.item(), outputs, targets, lengths_k, [sentences_ids[i] for i in keep], [sentence_frames[i] for i in keep],
                [phonemes[i] for i in keep], reference_arrays)
    return acc.info(dataloader.dataset.dataset_config)
