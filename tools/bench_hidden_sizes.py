"""Training step (fwd + loss + bwd + Adam) of the BiGRU model at hidden sizes with and without register-resident recurrence
kernels (32 / 64 / 128 vs any other multiple of 4, csrc/gru.hip), B=32 T=200 A=11.  usage: python tools/bench_hidden_sizes.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd.engine import TrainStep  # noqa: E402
from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech  # noqa: E402

dev = torch.device("cuda:0")
B, T, V, A = 32, 200, 45, 11
torch.manual_seed(0)
tokens = torch.randint(1, V, (B, T), device=dev)
targets = torch.rand(B, T, A, 2, 50, device=dev)
lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
scale = 1.0 / (B * T * A * 50)
for H in (64, 96, 128, 192, 256):
    model = ArtSpeech(V, A, hidden_size=H).to(dev)
    step = TrainStep(model, B, T, lr=1e-4, weight_decay=1e-6, pipeline=True)
    for _ in range(10):
        step.step(tokens, lengths, targets, scale)
    step.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        step.step(tokens, lengths, targets, scale)
    step.flush()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"hidden {H:4d}: {ms:7.3f} ms/step  {B * T / ms * 1e3:10.0f} frames/s  loss {float(step.loss_value()):.5f}", flush=True)
