"""Per-step GPU and host times of the first steps of a cold process (why a 20-step run after 5 warm-up steps reads higher
than a 200-step run): python3 tools/step_warmup_curve.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from artspeech_amd.distributed import loss_scale
from artspeech_amd.engine import TrainStep
from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech

V, E, H, A, N, B, T = 45, 64, 128, 11, 50, 32, 200
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = ArtSpeech(V, A, embed_dim=E, hidden_size=H, n_samples=N).to(dev)
g = torch.Generator().manual_seed(1)
tokens = torch.randint(1, V, (B, T), generator=g).to(dev)
targets = torch.rand(B, T, A, 2, N, generator=g).to(dev)
lengths = torch.full((B,), T, dtype=torch.int32).to(dev)
scale = loss_scale(B * T, A, N)
step = TrainStep(model, B, T, lr=1e-4, weight_decay=1e-6, pipeline=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
torch.cuda.synchronize()
ev[0].record()
for i in range(n):
    t0 = time.perf_counter()
    step.step(tokens, lengths, targets, scale)
    host.append(1e3 * (time.perf_counter() - t0))
    ev[i + 1].record()
step.flush()
torch.cuda.synchronize()
gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for i in range(n):
    print(f"step {i:3d}: gpu {gpu[i]:7.3f} ms   host enqueue {host[i]:7.3f} ms")
