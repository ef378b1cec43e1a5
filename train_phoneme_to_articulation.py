####################################################################################################
#
# Train the model-free phoneme-to-articulation network on MI355X
#
# Entry point kept from the reference (train_phoneme_to_articulation.py): same CLI
#   python train_phoneme_to_articulation.py --config cfg.yaml [--mlflow URI --experiment NAME
#          --run_id ID --run_name NAME --checkpoint PATH]
# same YAML keys (= the keyword arguments of main()), same run_epoch() contract, same checkpoint dict.
# Extras: `datadir: synthetic` trains on SyntheticArtSpeechDataset; launched under torchrun
# (one process per GPU) every global batch is sharded by utterance and the flat gradient buffer is
# all-reduced over RCCL before the optimizer step.
#
####################################################################################################
import argparse
import json
import logging
import os
import random
import shutil
import tempfile
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist
import yaml
from torch.optim import Adam
from torch.optim.lr_scheduler import ReduceLROnPlateau
from torch.utils.data import DataLoader

from artspeech_amd import distributed as dp
from artspeech_amd.helpers import make_padding_mask, set_seeds
from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import (
    ArtSpeechDataset,
    HBMResidentDataset,
    SyntheticArtSpeechDataset,
    pad_sequence_collate_fn,
)
from artspeech_amd.phoneme_to_articulation.encoder_decoder.evaluation import run_test
from artspeech_amd.phoneme_to_articulation.encoder_decoder.metrics import P2CPDistance
from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
from artspeech_amd.phoneme_to_articulation.metrics import EuclideanDistance, masked_euclidean_loss
from artspeech_amd.settings import BLANK, DATASET_CONFIG, TRAIN, UNKNOWN, VALID

try:  # mlflow is optional here (absent from the MI355X image): same flags, no-op logging
    import mlflow
except ImportError:  # pragma: no cover
    mlflow = None


def _mlflow(fn, *args, **kwargs):
    if mlflow is not None:
        return getattr(mlflow, fn)(*args, **kwargs)


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def run_epoch(phase, epoch, model, dataloader, optimizer, criterion, fn_metrics=None, scheduler=None, device=None):
    """One pass over `dataloader` (reference :45-121).  Returns {"loss": mean, metric_name: mean, ...}."""
    if device is None:
        device = torch.device("cuda")
    fn_metrics = fn_metrics or {}
    training = phase == TRAIN
    model.train() if training else model.eval()
    rank, world = _world()

    losses = []
    metrics_values = {name: [] for name in fn_metrics}
    # reduction "none" (the training configuration): criterion + mask + mean collapse into one kernel
    fused = isinstance(criterion, EuclideanDistance) and getattr(torch, criterion.reduction_name, None) is None
    deferred = hasattr(model, "check_tokens")   # token-id check next to the loop's own loss.item() instead of a sync per forward
    keep_defer = getattr(model, "defer_token_check", False)
    if deferred:
        model.defer_token_check = True          # for this loop only: restored below, whatever happens
    try:
        return _run_epoch_batches(phase, model, dataloader, optimizer, criterion, fn_metrics, scheduler, device, training, rank, world,
                                  losses, metrics_values, fused, deferred)
    finally:
        if deferred:
            model.defer_token_check = keep_defer
            model.check_tokens()                # nothing stays pending (and an id outside the vocabulary still raises)


def _run_epoch_batches(phase, model, dataloader, optimizer, criterion, fn_metrics, scheduler, device, training, rank, world, losses,
                       metrics_values, fused, deferred):
    for _, sentence, targets, lengths, _, _, _, _ in dataloader:
        n_valid_global = int(lengths.sum())
        if world > 1:  # every rank sees the same global batch (same sampler seed) and keeps its shard
            sentence, targets, lengths, n_valid_global = dp.shard_batch(sentence, targets, lengths, rank, world)
        sentence, targets = sentence.to(device), targets.to(device)
        optimizer.zero_grad()
        with torch.set_grad_enabled(training):
            outputs = model(sentence, lengths)
            if fused:  # criterion + padding mask + mean in one kernel; shard losses sum to the global mean
                loss = masked_euclidean_loss(outputs, targets, lengths, n_valid_global=n_valid_global)
            else:      # the reference's expression (:86-90)
                loss = criterion(outputs, targets[:, :outputs.shape[1]])
                padding_mask = make_padding_mask(lengths)
                bs, max_len, num_articulators, features = loss.shape
                loss = loss.view(bs * max_len, num_articulators, features)
                loss = loss[padding_mask.view(bs * max_len).to(device)].mean()
            if training:
                loss.backward()
                if world > 1:
                    dp.all_reduce_flat(model.flat.grad)
                optimizer.step()
                if scheduler is not None:
                    scheduler.step()
            step_loss = loss.detach().clone()
            if world > 1:
                dp.all_reduce_flat(step_loss)
            for name, fn_metric in fn_metrics.items():
                metrics_values[name].append(fn_metric(outputs, targets, lengths).item())
            losses.append(step_loss.item())
            if deferred:
                model.check_tokens()   # IndexError like nn.Embedding (reference models.py:135) for ids outside the vocabulary
    info = {"loss": float(np.mean(losses))}
    info.update({name: float(np.mean(vals)) for name, vals in metrics_values.items()})
    return info


def _make_dataset(datadir, database_name, seq_dict, vocabulary, articulators, clip_tails, synthetic, seed):
    if datadir == "synthetic":
        cfg = dict(synthetic or {})
        n = cfg.pop("num_sentences", 64) if not isinstance(seq_dict, dict) else seq_dict.get("num_sentences", cfg.pop("num_sentences", 64))
        return SyntheticArtSpeechDataset(n, vocabulary, articulators, seed=seed, database_name=database_name, **cfg)
    from artspeech_amd.helpers import sequences_from_dict
    return ArtSpeechDataset(datadir, database_name, sequences_from_dict(datadir, seq_dict), vocabulary, articulators,
                            clip_tails=clip_tails)


def build_vocabulary(vocab_filepath):
    """{token: index}: the two default tokens first, then the JSON list (reference :151-156); without a file, 43 synthetic
    phoneme names (V = 45)."""
    vocabulary = {token: i for i, token in enumerate([BLANK, UNKNOWN])}
    if vocab_filepath is not None:
        with open(vocab_filepath) as f:
            tokens = json.load(f)
    else:
        tokens = [f"ph{i:02d}" for i in range(43)]
    for i, token in enumerate(tokens, start=len(vocabulary)):
        vocabulary[token] = i
    return vocabulary


def main(datadir, database_name, num_epochs, batch_size, patience, learning_rate, weight_decay, train_seq_dict,
         valid_seq_dict, test_seq_dict, vocab_filepath, articulators, model_kwargs=None, num_workers=0, clip_tails=True,
         state_dict_filepath=None, checkpoint_filepath=None, seed=0, synthetic=None, results_dir=None, hbm_resident=False):
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        backend = os.environ.get("ARTSPEECH_DIST_BACKEND", "nccl")  # "gloo": rehearsal with several ranks on one GPU
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
        dist.init_process_group(backend)
    rank, world = _world()
    device = torch.device("cuda", torch.cuda.current_device())
    logging.info(f"Running on '{device}' (rank {rank}/{world})")
    results_dir = results_dir or RESULTS_DIR
    os.makedirs(results_dir, exist_ok=True)
    best_model_path = os.path.join(results_dir, "best_model.pt")
    last_model_path = os.path.join(results_dir, "last_model.pt")
    save_checkpoint_path = os.path.join(results_dir, "checkpoint.pt")

    vocabulary = build_vocabulary(vocab_filepath)

    model = ArtSpeech(len(vocabulary), len(articulators), **(model_kwargs or {}))
    if state_dict_filepath is not None:
        model.load_state_dict(torch.load(state_dict_filepath, map_location="cpu"))
    model.to(device)
    dp.broadcast_parameters(model)
    if rank == 0:
        print(f"\nArtSpeech -- {model.total_parameters} parameters\n")
    _mlflow("log_param", "num_network_params", model.total_parameters)

    loss_fn = EuclideanDistance(reduction="none")
    optimizer = Adam(model.parameters(), lr=learning_rate, weight_decay=weight_decay)
    scheduler = ReduceLROnPlateau(optimizer, factor=0.1, patience=10)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)

    def loader(seq_dict, shuffle, ds_seed):
        ds = _make_dataset(datadir, database_name, seq_dict, vocabulary, articulators, clip_tails, synthetic, ds_seed)
        if hbm_resident:   # YAML key `hbm_resident: true`: the data set lives in HBM, batches are collated on the device
            ds = HBMResidentDataset(ds, device)
            return DataLoader(ds, batch_size=batch_size, shuffle=shuffle, num_workers=0, collate_fn=ds.collate, generator=gen)
        return DataLoader(ds, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, worker_init_fn=set_seeds,
                          collate_fn=pad_sequence_collate_fn, generator=gen)

    train_dataloader = loader(train_seq_dict, True, seed)
    valid_dataloader = loader(valid_seq_dict, False, seed + 1)
    fn_metrics = {"p2cp_mean": P2CPDistance(dataset_config=DATASET_CONFIG[database_name])}

    epochs = range(1, num_epochs + 1)
    best_metric, epochs_since_best = np.inf, 0
    if checkpoint_filepath is not None:
        checkpoint = torch.load(checkpoint_filepath, map_location="cpu")
        model.load_state_dict(checkpoint["model"])
        optimizer.load_state_dict(checkpoint["optimizer"])
        scheduler.load_state_dict(checkpoint["scheduler"])
        epochs = range(checkpoint["epoch"] + 1, num_epochs + 1)
        best_metric, epochs_since_best = checkpoint["best_metric"], checkpoint["epochs_since_best"]
        best_model_path, last_model_path = checkpoint["best_model_path"], checkpoint["last_model_path"]

    for epoch in epochs:
        info_train = run_epoch(TRAIN, epoch, model, train_dataloader, optimizer, loss_fn, device=device)
        _mlflow("log_metrics", {f"train_{k}": v for k, v in info_train.items()}, step=epoch)
        info_valid = run_epoch(VALID, epoch, model, valid_dataloader, optimizer, loss_fn, fn_metrics=fn_metrics, device=device)
        _mlflow("log_metrics", {f"valid_{k}": v for k, v in info_valid.items()}, step=epoch)
        if rank == 0:
            print(f"epoch {epoch}: train loss {info_train['loss']:.5f}  valid loss {info_valid['loss']:.5f}  "
                  f"p2cp_mean {info_valid['p2cp_mean']:.3f} mm", flush=True)
        scheduler.step(info_valid["loss"])          # LR schedule follows the validation LOSS (:290)
        if info_valid["p2cp_mean"] < best_metric:   # model selection follows p2cp_mean (:292-298)
            best_metric, epochs_since_best = info_valid["p2cp_mean"], 0
            if rank == 0:
                torch.save(model.state_dict(), best_model_path)
        else:
            epochs_since_best += 1
        if rank == 0:
            torch.save(model.state_dict(), last_model_path)
            torch.save({
                "epoch": epoch, "model": model.state_dict(), "optimizer": optimizer.state_dict(),
                "scheduler": scheduler.state_dict(), "best_metric": best_metric, "epochs_since_best": epochs_since_best,
                "best_model_path": best_model_path, "last_model_path": last_model_path,
            }, save_checkpoint_path)
        if epochs_since_best > patience:
            break

    results = None
    if rank == 0:
        test_dataloader = loader(test_seq_dict, False, seed + 2)
        if os.path.exists(best_model_path):
            model.load_state_dict(torch.load(best_model_path, map_location="cpu"))
        results = run_test(epoch=0, model=model, dataloader=test_dataloader, criterion=loss_fn,
                           outputs_dir=os.path.join(results_dir, "test_outputs"), articulators=sorted(articulators),
                           device=device)
        with open(os.path.join(results_dir, "test_results.json"), "w") as f:
            json.dump(results, f, indent=1)
    if world > 1:
        dist.barrier()
    return results


TMP_DIR = tempfile.mkdtemp(prefix="artspeech_")
RESULTS_DIR = os.path.join(TMP_DIR, "results")

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", dest="config_filepath")
    parser.add_argument("--mlflow", dest="mlflow_tracking_uri", default=None)
    parser.add_argument("--experiment", dest="experiment_name", default="phoneme_to_articulation")
    parser.add_argument("--run_id", dest="run_id", default=None)
    parser.add_argument("--run_name", dest="run_name", default=None)
    parser.add_argument("--checkpoint", dest="checkpoint_filepath", default=None)
    args = parser.parse_args()
    seed = 0
    random.seed(seed)
    torch.manual_seed(seed)
    np.random.seed(seed)
    with open(args.config_filepath) as f:
        cfg = yaml.safe_load(f)
    if mlflow is not None and args.mlflow_tracking_uri is not None:
        mlflow.set_tracking_uri(args.mlflow_tracking_uri)
        mlflow.set_experiment(args.experiment_name)
    try:
        main(**cfg, checkpoint_filepath=args.checkpoint_filepath, seed=seed)
    finally:
        shutil.rmtree(TMP_DIR, ignore_errors=True)
