"""Data-parallel rehearsal of the transformer trainer's gradient exchange (all-reduce hooks fired during the backward):
summed shard gradients must equal the full-batch gradients.  Launch with torch.distributed.run, e.g.
  ARTSPEECH_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/check_dp_transformer.py"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artspeech_amd import distributed as dp  # noqa: E402
from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn  # noqa: E402
from artspeech_amd.phoneme_to_articulation.metrics import masked_euclidean_loss  # noqa: E402
from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer  # noqa: E402
from train_phoneme_to_articulation_transformer import _GradientExchange  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("ARTSPEECH_DIST_BACKEND", "nccl")
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend, rank=rank, world_size=world)
V, A, nf = 13, 3, 20


def grads_of(model, c, idx):
    tokens, targets, lengths = c[1][idx], c[2][idx], c[3][idx]
    t_max = int(lengths.max())
    tokens, targets = tokens[:, :t_max].to(dev), targets[:, :t_max].to(dev)
    kw = dict(src_key_padding_mask=c[8][idx][:, :t_max].to(dev), tgt_key_padding_mask=c[9][idx][:, :t_max].to(dev),
              src_attn_mask=c[10][idx][:, :t_max, :t_max].to(dev), tgt_attn_mask=c[11][idx][:, :t_max, :t_max].to(dev))
    bs = tokens.shape[0]
    shifted = torch.cat([torch.zeros(bs, 1, A, nf, device=dev), targets[:, 1:].reshape(bs, t_max - 1, A, nf)], dim=1)
    for p in model.parameters():
        p.grad = None
    loss = masked_euclidean_loss(model(tokens, shifted, **kw), targets, lengths, n_valid_global=int(c[3].sum()))
    loss.backward()
    return loss.detach()


torch.manual_seed(0)
batch = [(f"s{i}", torch.randint(1, V, (l,)), torch.rand(l, A, 2, nf // 2), ["p"] * l, torch.rand(l, 1, 2, nf // 2),
          torch.tensor([], dtype=torch.int), list(range(l)), torch.zeros(l)) for i, l in enumerate((9, 8, 6, 5, 3, 2))]
c = pad_sequence_transformer_collate_fn(batch)
torch.manual_seed(1)
full = ArtSpeechTransformer(V, A, embed_dim=32, num_heads=4, num_layers=2, num_feat=nf).to(dev).eval()
torch.manual_seed(1)
sharded = ArtSpeechTransformer(V, A, embed_dim=32, num_heads=4, num_layers=2, num_feat=nf).to(dev).eval()
full_loss = grads_of(full, c, list(range(len(batch))))
exchange = _GradientExchange(sharded)
shard_loss = grads_of(sharded, c, dp.shard_indices(len(batch), rank, world))
exchange.wait()
dist.all_reduce(shard_loss)
worst = max(float((a.grad - b.grad).abs().max() / (b.grad.abs().max() + 1e-12)) for a, b in zip(sharded.parameters(), full.parameters()))
ok = worst < 1e-4 and abs(float(shard_loss) - float(full_loss)) < 1e-6
print(f"rank {rank}/{world}: summed shard gradients vs full batch: worst relative error {worst:.2e}; loss {float(shard_loss):.6f} vs "
      f"{float(full_loss):.6f}; ok = {ok}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
