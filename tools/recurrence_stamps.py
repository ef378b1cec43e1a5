"""In-step cost of the recurrences (round-3 evidence, profiles/r03_recurrence_in_step.log): the pipelined training step of
bench.py with the in-kernel stamps of the two BACKWARD recurrences switched on (as_gru_debug_stamps: shader cycles and 100 MHz
wall ticks per workgroup), and the four recurrence launches' HIP-event times from the library's phase table.
    python3 tools/recurrence_stamps.py [steps]
Per launch: microseconds per launch (events), ns per recurrent step (= us / T), and for the backward ones the in-kernel time,
shader cycles per recurrent step and the in-kernel clock, median over the 64 workgroups of the last step."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402
from artspeech_amd.engine import TrainStep  # noqa: E402
from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
V, A, B, T = 45, 11, 32, 200
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtSpeech(V, A).to(dev)
g = torch.Generator().manual_seed(1)
tokens = torch.randint(1, V, (B, T), generator=g).to(dev)
targets = torch.rand(B, T, A, 2, 50, generator=g).to(dev)
lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
scale = 1.0 / (B * T * A * 50)
step = TrainStep(model, B, T, lr=1e-4, weight_decay=1e-6, pipeline=True)
L = _lib.lib()
for _ in range(20):
    step.step(tokens, lengths, targets, scale)
step.flush()
torch.cuda.synchronize()
stamps = torch.zeros(2 * 2 * B * 4, dtype=torch.int64, device=dev)
L.as_gru_debug_stamps(_lib.ptr(stamps))
L.as_profile_reset()
L.as_profile_enable(1)
for _ in range(steps):
    step.step(tokens, lengths, targets, scale)
step.flush()
torch.cuda.synchronize()
L.as_profile_enable(0)
L.as_gru_debug_stamps(None)
buf = C.create_string_buffer(1 << 16)
L.as_profile_report(buf, len(buf))
phases = {}
for line in buf.value.decode().splitlines():
    name, cnt, ms = line.split()
    phases[name] = 1e3 * float(ms) / int(cnt)
st = stamps.cpu().numpy().reshape(2, 2 * B, 4).astype(np.float64)
print(f"B={B} T={T} H=128, pipelined engine step, {steps} steps; HIP-event time per launch and in-kernel stamps (median over 64 workgroups)")
for name, half in (("gru.fwd_l0", None), ("gru.fwd_l1", None), ("gru.bwd_l1", 0), ("gru.bwd_l0", 1)):
    us = phases[name]
    line = f"{name:11s} {us:7.1f} us/launch = {1e3 * us / T:6.0f} ns per recurrent step"
    if half is not None:
        cyc, ticks = st[half][:, 0], st[half][:, 1]
        line += (f" | in kernel {np.median(ticks) / 100:6.1f} us, {np.median(cyc) / T:6.0f} shader cycles per step, "
                 f"clock {np.median(cyc / ticks) * 100:5.0f} MHz")
    print(line, flush=True)
