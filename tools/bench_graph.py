"""Does replaying the forward + loss + backward launch sequence as a HIP graph shorten the training step?
(The step is ~60 launches on two streams; the composite C-ABI entry points are capture-safe: no allocation, no sync.)
usage: python tools/bench_graph.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd.engine import TrainStep  # noqa: E402
from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B, T, A, N, V = 32, 200, 11, 50, 45
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtSpeech(V, A).to(dev)
tokens = torch.randint(1, V, (B, T), device=dev)
targets = torch.rand(B, T, A, 2, N, device=dev)
lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
scale = 1.0 / (B * T * A * N)
step = TrainStep(model, B, T, lr=1e-4, weight_decay=1e-6)
for _ in range(5):
    step.step(tokens, lengths, targets, scale)
torch.cuda.synchronize()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


eager = timed(lambda: step.step(tokens, lengths, targets, scale))
loss_eager = float(step.loss)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step.forward_backward(tokens, lengths, targets, scale)


def replay():
    graph.replay()
    step.all_reduce()
    step.adam()


replay()
torch.cuda.synchronize()
g = timed(replay)
print(f"eager launches: {eager:.3f} ms/step; graph replay of forward+loss+backward (+ eager Adam): {g:.3f} ms/step; "
      f"loss {loss_eager:.6f} -> {float(step.loss):.6f}", flush=True)
