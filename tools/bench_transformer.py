"""Forward-only throughput of the transformer variant at BASELINE configs[3] (d=256, L=6, A=11, B=32, T=200).
usage: python tools/bench_transformer.py [B] [T] [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn  # noqa: E402
from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
V, A, d, h, L, nf = 45, 11, 256, 4, 6, 100
dev = torch.device("cuda:0")
t0 = time.time()
model = ArtSpeechTransformer(V, A, embed_dim=d, num_heads=h, num_layers=L, num_feat=nf).to(dev).eval()
print(f"model: {model.total_parameters} parameters, built in {time.time() - t0:.1f} s", flush=True)
batch = [(f"s{i}", torch.randint(1, V, (T,)), torch.rand(T, A, 2, nf // 2), ["p"] * T, torch.rand(T, 1, 2, nf // 2),
          torch.tensor([], dtype=torch.int), list(range(T)), torch.zeros(T)) for i in range(B)]
c = pad_sequence_transformer_collate_fn(batch)
tokens, targets = c[1].to(dev), c[2].to(dev)
shifted = torch.cat([torch.zeros(B, 1, A, nf, device=dev), targets[:, 1:].reshape(B, T - 1, A, nf)], dim=1)
kw = dict(src_key_padding_mask=c[8].to(dev), tgt_key_padding_mask=c[9].to(dev), src_attn_mask=c[10].to(dev), tgt_attn_mask=c[11].to(dev))
with torch.no_grad():
    out = model(tokens, shifted, **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = model(tokens, shifted, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
flops = 2 * 0.503e9 * B * T  # ~0.50 GMAC per frame forward
print(f"forward B={B} T={T}: {dt * 1e3:.1f} ms  -> {B * T / dt:.0f} frames/s, {flops / dt / 1e12:.1f} TFLOP/s (fp32 MFMA peak 157.3); "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)

# ---- training step: forward + masked Euclidean loss + backward (no optimizer), dropout 0 like the parity runs
from artspeech_amd.phoneme_to_articulation.metrics import masked_euclidean_loss  # noqa: E402
model.eval()  # deterministic (the encoder's library-default dropout 0.1 only acts in train mode); grads still flow
lengths = c[3]
torch.cuda.reset_peak_memory_stats()


def step():
    for p_ in model.parameters():
        p_.grad = None
    out = model(tokens, shifted, **kw)
    loss = masked_euclidean_loss(out, targets, lengths)
    loss.backward()
    return loss


loss = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"fwd+bwd B={B} T={T}: {dt * 1e3:.1f} ms -> {B * T / dt:.0f} frames/s, {3 * flops / dt / 1e12:.1f} TFLOP/s; loss {loss.item():.5f}; "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
