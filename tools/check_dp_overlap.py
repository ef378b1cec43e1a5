"""Data-parallel rehearsal: the two-piece gradient all-reduce that overlaps the GRU backward (TrainStep.ar_overlap) must give
bit-identical parameters to the single all-reduce after the backward.  Launch with torch.distributed.run (any backend that
takes GPU tensors; `gloo` lets two ranks share one GPU):
  ARTSPEECH_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/check_dp_overlap.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd.engine import TrainStep  # noqa: E402
from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("ARTSPEECH_DIST_BACKEND", "nccl")
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend, rank=rank, world_size=world)
B, T, A, N, V = 6, 40, 3, 10, 17
results = []
# third case (round 3): the PIPELINED engine -- the layer-2 weight gradient, its all-reduce and its Adam slice run one step
# late on a side stream -- must leave the same parameters after flush() as the plain one
for overlap, pipeline in ((True, False), (False, False), (False, True)):
    torch.manual_seed(0)
    model = ArtSpeech(V, A, embed_dim=16, hidden_size=32, n_samples=N).to(dev)
    g = torch.Generator().manual_seed(100 + rank)
    lengths = torch.sort(torch.randint(5, T + 1, (B,), generator=g), descending=True).values.int()
    lengths[0] = T
    tokens = torch.randint(1, V, (B, T), generator=g).to(dev)
    targets = torch.rand(B, T, A, 2, N, generator=g).to(dev)
    n_valid = torch.tensor([int(lengths.sum())], device=dev)
    dist.all_reduce(n_valid)
    step = TrainStep(model, B, T, lr=1e-3, pipeline=pipeline)
    assert step.pipeline == pipeline
    if os.environ.get("FORCE_DIST"):  # exercise the collective code path on a single rank (RCCL API usage under streams)
        step.use_dist = True
    step.ar_overlap = overlap and step.use_dist
    if step.ar_overlap and step.comm_stream is None:
        step.comm_stream = torch.cuda.Stream(device=dev)
    for _ in range(4):
        step.step(tokens, lengths.to(dev), targets, 1.0 / (int(n_valid) * A * N))
    step.flush()
    torch.cuda.synchronize()
    results.append((model.flat.data.clone(), step.grads.clone(), float(step.loss)))
same = torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])
same_pipe = torch.equal(results[2][0], results[1][0]) and torch.equal(results[2][1], results[1][1])
print(f"rank {rank}/{world}: pipelined == plain: {same_pipe}", flush=True)
same = same and same_pipe
# every rank must hold the same parameters after the synchronised updates
ref = results[0][0].clone()
dist.broadcast(ref, src=0)
in_sync = torch.equal(ref, results[0][0])
print(f"rank {rank}/{world}: overlap == plain all-reduce: {same}; ranks in sync: {in_sync}; loss shard {results[0][2]:.6f}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if same and in_sync else 1)
