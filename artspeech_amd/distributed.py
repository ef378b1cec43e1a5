"""Data parallelism for the model-free path: one process per GPU, `torch.distributed` ("nccl" = RCCL over
xGMI on ROCm; "gloo" for CPU rehearsals).  The reference has no multi-device code; this is
the batch-sharded scheme of SURVEY 8e:

  * the length-sorted batch is dealt round-robin (rank r takes utterances r, r+G, ...): every rank keeps
    a similar length mix and its shard stays sorted descending, as the packed GRU requires;
  * each rank scales its summed loss by 1 / (N_valid_GLOBAL * A * N), known on the host from `lengths`
    (no collective), so shard losses and shard gradients SUM to the reference's full-batch values;
  * one all-reduce(SUM) of the single flat gradient buffer per step.
"""
import torch
import torch.distributed as dist


def shard_indices(batch_size, rank, world_size):
    """Utterance indices of the (length-sorted) global batch owned by `rank`."""
    return list(range(rank, batch_size, world_size))


def shard_batch(tokens, targets, lengths, rank, world_size):
    """Round-robin shard of a collated batch.  Returns (tokens, targets, lengths, n_valid_global).
    tokens/targets are cut at the shard's own max length (T = max(lengths) is what the model expects)."""
    idx = shard_indices(tokens.shape[0], rank, world_size)
    if not idx:
        raise ValueError(f"global batch of {tokens.shape[0]} utterances cannot feed rank {rank} of {world_size}")
    lengths = torch.as_tensor(lengths)
    n_valid_global = int(lengths.sum())
    sel = torch.as_tensor(idx, device=tokens.device)
    my_len = lengths[idx]
    t_max = int(my_len.max())
    return tokens[sel][:, :t_max], targets[sel.to(targets.device)][:, :t_max], my_len, n_valid_global


def loss_scale(n_valid_global, n_articulators, n_samples):
    return 1.0 / (n_valid_global * n_articulators * n_samples)


def all_reduce_flat(flat, group=None):
    """SUM all-reduce of a flat buffer (gradients, or scalar metrics packed into one tensor)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_parameters(model, src=0, group=None):
    """Make every rank start from rank `src`'s parameters (one broadcast of the flat buffer)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(model.flat.data, src=src, group=group)
