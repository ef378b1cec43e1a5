####################################################################################################
# Evaluation CLI of the model-free phoneme-to-articulation path on MI355X
# (reference: test_phoneme_to_articulation.py:23-123 -- checkpoint -> test split -> test_results.{json,csv}).
#
#   python test_phoneme_to_articulation.py --config configs/test_synthetic.yaml
#
# The YAML keys are the keyword arguments of main(), as in the reference.  Extras: `datadir: synthetic` evaluates on
# SyntheticArtSpeechDataset (`synthetic:` options, `seed`); `regularize_out` (default false) asks for the B-spline
# regularised contour dumps, which need the external vt_tools package (the reference always passes True, :92).
####################################################################################################
import argparse
import csv
import json
import os

import torch
import yaml
from torch.utils.data import DataLoader

from artspeech_amd.helpers import set_seeds
from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_collate_fn
from artspeech_amd.phoneme_to_articulation.encoder_decoder.evaluation import run_test
from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
from artspeech_amd.phoneme_to_articulation.metrics import EuclideanDistance
from train_phoneme_to_articulation import _make_dataset, build_vocabulary


def write_results(test_results, articulators, save_to):
    """test_results.json + the one-row test_results.csv of the reference (:95-112)."""
    with open(os.path.join(save_to, "test_results.json"), "w") as f:
        json.dump(test_results, f)
    item = {"exp": None, "loss": test_results["loss"]}
    for articulator in articulators:
        for key in ("p2cp", "p2cp_mm", "med", "med_mm"):
            item[f"{key}_{articulator}"] = test_results[articulator][key]
    with open(os.path.join(save_to, "test_results.csv"), "w", newline="") as f:
        writer = csv.DictWriter(f, fieldnames=list(item.keys()), lineterminator="\n")  # DataFrame([item]).to_csv(index=False)
        writer.writeheader()
        writer.writerow({k: ("" if v is None else v) for k, v in item.items()})


def main(datadir, database_name, batch_size, test_seq_dict, state_dict_fpath, vocab_filepath, articulators, save_to,
         model_kwargs=None, clip_tails=True, num_workers=0, synthetic=None, seed=0, regularize_out=False):
    device = torch.device("cuda", torch.cuda.current_device())
    vocabulary = build_vocabulary(vocab_filepath)
    test_dataset = _make_dataset(datadir, database_name, test_seq_dict, vocabulary, articulators, clip_tails, synthetic, seed + 2)
    test_dataloader = DataLoader(test_dataset, batch_size=batch_size, shuffle=False, num_workers=num_workers,
                                 worker_init_fn=set_seeds, collate_fn=pad_sequence_collate_fn)
    best_model = ArtSpeech(len(vocabulary), len(articulators), **(model_kwargs or {}))
    if state_dict_fpath is not None:
        best_model.load_state_dict(torch.load(state_dict_fpath, map_location="cpu"))
    best_model.to(device)
    print(f"\nArtSpeech -- {best_model.total_parameters} parameters\n")

    test_outputs_dir = os.path.join(save_to, "test_outputs")
    os.makedirs(test_outputs_dir, exist_ok=True)
    test_results = run_test(epoch=0, model=best_model, dataloader=test_dataloader, criterion=EuclideanDistance("none"),
                            outputs_dir=test_outputs_dir, articulators=articulators, device=device, regularize_out=regularize_out)
    write_results(test_results, test_dataset.articulators, save_to)
    return test_results


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", dest="cfg_filepath")
    args = parser.parse_args()
    with open(args.cfg_filepath) as f:
        cfg = yaml.safe_load(f.read())
    main(**cfg)
