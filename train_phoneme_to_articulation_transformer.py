####################################################################################################
#
# Train the transformer phoneme-to-articulation network on MI355X
#
# Entry point kept from the reference (train_phoneme_to_articulation_transformer.py): same CLI, same YAML keys
# (= keyword arguments of main()), same run_epoch() contract: 12-field batches from
# pad_sequence_transformer_collate_fn; the decoder input is zeros followed by targets[:, 1:] (reference :99-102 --
# reproduced as is: frame t >= 1 sees its own ground truth); masked mean Euclidean loss (:114-118).
# `datadir: synthetic` trains on SyntheticArtSpeechDataset; under torchrun every global batch is sharded by
# utterance and gradients are all-reduced over RCCL before the optimizer step.
#
####################################################################################################
import argparse
import os
import random
import shutil
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import yaml
from torch.optim import Adam
from torch.optim.lr_scheduler import ReduceLROnPlateau
from torch.utils.data import DataLoader

from artspeech_amd import distributed as dp
from artspeech_amd.helpers import set_seeds
from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import (
    SyntheticArtSpeechDataset,
    pad_sequence_transformer_collate_fn,
)
from artspeech_amd.phoneme_to_articulation.encoder_decoder.metrics import P2CPDistance
from artspeech_amd.phoneme_to_articulation.metrics import EuclideanDistance, masked_euclidean_loss
from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
from artspeech_amd.settings import DATASET_CONFIG, TRAIN, VALID
from train_phoneme_to_articulation import build_vocabulary


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


class _GradientExchange:
    """Data-parallel gradient all-reduce (SUM; shard losses are scaled by the global frame count) overlapped with the
    backward: a post-accumulate hook on every parameter starts that tensor's all-reduce the moment autograd has finished its
    gradient -- the parameters are stored stacked per decoder layer, so the 1.68 GB of gradients go out as ~30 large
    collectives, last layers first, while the earlier layers are still being differentiated.  `wait()` before the optimizer."""

    def __init__(self, model):
        self.pending = []
        self.asynchronous = dist.get_backend() == "nccl"  # gloo (CPU rehearsals): blocking calls
        for p in model.parameters():
            if p.requires_grad:
                p.register_post_accumulate_grad_hook(self._hook)

    def _hook(self, p):
        work = dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=self.asynchronous)
        if self.asynchronous:
            self.pending.append(work)

    def wait(self):
        for work in self.pending:
            work.wait()
        self.pending.clear()


_EXCHANGE = {}


def _gradient_exchange(model):
    if id(model) not in _EXCHANGE:
        _EXCHANGE[id(model)] = _GradientExchange(model)
    return _EXCHANGE[id(model)]


def run_epoch(phase, epoch, model, dataloader, optimizer, criterion, fn_metrics=None, scheduler=None, device=None):
    """One pass over `dataloader` (reference :49-149).  Returns {"loss": mean, metric_name: mean, ...}."""
    device = device or torch.device("cuda")
    fn_metrics = fn_metrics or {}
    training = phase == TRAIN
    model.train() if training else model.eval()
    rank, world = _world()
    losses, metrics_values = [], {name: [] for name in fn_metrics}
    for (_, sentence, targets, lengths, _, _, _, _, src_kpm, tgt_kpm, src_attn_mask, tgt_attn_mask) in dataloader:
        n_valid_global = int(lengths.sum())
        if world > 1:  # round-robin shard of the (length sorted) batch; masks follow the shard's own max length
            idx = dp.shard_indices(sentence.shape[0], rank, world)
            lengths = lengths[idx]
            t_max = int(lengths.max())
            sentence, targets = sentence[idx][:, :t_max], targets[idx][:, :t_max]
            src_kpm, tgt_kpm = src_kpm[idx][:, :t_max], tgt_kpm[idx][:, :t_max]
            src_attn_mask, tgt_attn_mask = src_attn_mask[idx][:, :t_max, :t_max], tgt_attn_mask[idx][:, :t_max, :t_max]
        sentence, targets = sentence.to(device), targets.to(device)
        src_kpm, tgt_kpm = src_kpm.to(device), tgt_kpm.to(device)
        src_attn_mask, tgt_attn_mask = src_attn_mask.to(device), tgt_attn_mask.to(device)
        bs, seq_len, channels, _, features = targets.shape
        optimizer.zero_grad()
        with torch.set_grad_enabled(training):
            targets_right_shifted = torch.cat([torch.zeros(bs, 1, channels, 2 * features, device=device),
                                               targets[:, 1:].reshape(bs, seq_len - 1, channels, 2 * features)], dim=1)
            outputs = model(sentence, targets_right_shifted, src_key_padding_mask=src_kpm, tgt_key_padding_mask=tgt_kpm,
                            src_attn_mask=src_attn_mask, tgt_attn_mask=tgt_attn_mask)
            loss = masked_euclidean_loss(outputs, targets, lengths, n_valid_global=n_valid_global)
            if training:
                exchange = _gradient_exchange(model) if world > 1 else None  # hooks registered once per model
                loss.backward()
                if exchange is not None:
                    exchange.wait()
                optimizer.step()
                if scheduler is not None:
                    scheduler.step()
            step_loss = loss.detach().clone()
            if world > 1:
                dp.all_reduce_flat(step_loss)
            for name, fn_metric in fn_metrics.items():
                metrics_values[name].append(fn_metric(outputs.detach(), targets, lengths).item())
            losses.append(step_loss.item())
    info = {"loss": float(np.mean(losses))}
    info.update({name: float(np.mean(v)) for name, v in metrics_values.items()})
    return info


def main(datadir, database_name, num_epochs, batch_size, patience, learning_rate, weight_decay, train_seq_dict, valid_seq_dict,
         test_seq_dict, vocab_filepath, articulators, model_kwargs=None, num_workers=0, clip_tails=True, state_dict_filepath=None,
         checkpoint_filepath=None, seed=0, synthetic=None, results_dir=None):
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        backend = os.environ.get("ARTSPEECH_DIST_BACKEND", "nccl")  # "gloo": rehearsal with several ranks on one GPU
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
        dist.init_process_group(backend)
    rank, world = _world()
    device = torch.device("cuda", torch.cuda.current_device())
    results_dir = results_dir or RESULTS_DIR
    os.makedirs(results_dir, exist_ok=True)
    vocabulary = build_vocabulary(vocab_filepath)
    if datadir != "synthetic":
        raise NotImplementedError("real-data loading needs the reference's database_collector / vt_shape_gen stack; "
                                  "use `datadir: synthetic`")
    model = ArtSpeechTransformer(len(vocabulary), len(articulators), **(model_kwargs or {}))
    if state_dict_filepath is not None:
        model.load_state_dict(torch.load(state_dict_filepath, map_location="cpu"))
    model.to(device)
    if world > 1:
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
    if rank == 0:
        print(f"\nArtSpeechTransformer -- {model.total_parameters} parameters\n")
    loss_fn = EuclideanDistance(reduction="none")
    optimizer = Adam(model.parameters(), lr=learning_rate, weight_decay=weight_decay)
    scheduler = ReduceLROnPlateau(optimizer, factor=0.1, patience=10)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)

    def loader(seq_dict, shuffle, ds_seed):
        cfg = dict(synthetic or {})
        n = seq_dict.get("num_sentences", 64) if isinstance(seq_dict, dict) else 64
        ds = SyntheticArtSpeechDataset(n, vocabulary, articulators, seed=ds_seed, database_name=database_name, **cfg)
        return DataLoader(ds, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, worker_init_fn=set_seeds,
                          collate_fn=pad_sequence_transformer_collate_fn, generator=gen)

    train_dataloader, valid_dataloader = loader(train_seq_dict, True, seed), loader(valid_seq_dict, False, seed + 1)
    fn_metrics = {"p2cp_mean": P2CPDistance(dataset_config=DATASET_CONFIG[database_name])}
    best_metric, epochs_since_best = np.inf, 0
    for epoch in range(1, num_epochs + 1):
        info_train = run_epoch(TRAIN, epoch, model, train_dataloader, optimizer, loss_fn, device=device)
        info_valid = run_epoch(VALID, epoch, model, valid_dataloader, optimizer, loss_fn, fn_metrics=fn_metrics, device=device)
        if rank == 0:
            print(f"epoch {epoch}: train loss {info_train['loss']:.5f}  valid loss {info_valid['loss']:.5f}  "
                  f"p2cp_mean {info_valid['p2cp_mean']:.3f} mm", flush=True)
        scheduler.step(info_valid["loss"])
        if info_valid["p2cp_mean"] < best_metric:
            best_metric, epochs_since_best = info_valid["p2cp_mean"], 0
            if rank == 0:
                torch.save(model.state_dict(), os.path.join(results_dir, "best_model.pt"))
        else:
            epochs_since_best += 1
        if rank == 0:
            torch.save(model.state_dict(), os.path.join(results_dir, "last_model.pt"))
        if epochs_since_best > patience:
            break
    if world > 1:
        dist.barrier()
    return {"best_p2cp_mean": best_metric}


TMP_DIR = tempfile.mkdtemp(prefix="artspeech_tr_")
RESULTS_DIR = os.path.join(TMP_DIR, "results")

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", dest="config_filepath")
    parser.add_argument("--mlflow", dest="mlflow_tracking_uri", default=None)
    parser.add_argument("--experiment", dest="experiment_name", default="phoneme_to_articulation_transformer")
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
    try:
        main(**cfg, checkpoint_filepath=args.checkpoint_filepath, seed=seed)
    finally:
        shutil.rmtree(TMP_DIR, ignore_errors=True)
