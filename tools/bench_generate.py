"""Free-running ArtSpeechTransformer.generate() (the reference's test-time path: the whole prefix is re-decoded for every
frame, transformer/models.py:391-427) at d=256, L=6, A=11.  usage: python tools/bench_generate.py [B] [T]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd.phoneme_to_articulation.transformer import models as M  # noqa: E402
from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtSpeechTransformer(45, 11, embed_dim=256, num_heads=4, num_layers=6, num_feat=100).to(dev).eval()
tokens = torch.randint(1, 45, (B, T), device=dev)
kpm = torch.zeros(B, T, device=dev)
best = {}
with torch.no_grad():
    model.generate(tokens[:, :8], src_key_padding_mask=kpm[:, :8])  # warm-up
    for rep in range(2):  # alternate the two modes in one process (clocks drift between runs)
        for savings in (False, True):
            M.GENERATE_SAVINGS = savings
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = model.generate(tokens, src_key_padding_mask=kpm)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best[savings] = min(best.get(savings, 1e9), dt)
            assert out.shape == (B, T, 11, 2, 50) and torch.isfinite(out).all()
for savings, name in ((False, "everything re-decoded (as the reference)"), (True, "memory K/V once + last layer on the newest frame")):
    dt = best[savings]
    print(f"generate B={B} T={T}, {name}: {dt:.2f} s -> {B * T / dt:.0f} frames/s", flush=True)
print(f"{T} decoder passes over prefixes 1..{T}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
