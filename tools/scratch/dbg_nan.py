import sys, os
sys.path.insert(0, os.getcwd())
import torch
from artspeech_amd.phoneme_to_articulation.transformer import models as tm, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
V, A, d, h, L, nf = 20, 5, 32, 4, 2, 100
m = tm.ArtSpeechTransformer(V, A, embed_dim=d, num_heads=h, num_layers=L, num_feat=nf).to(dev).eval()
B, T = 3, 13
src = torch.randint(2, V, (B, T), device=dev)
kpm = torch.zeros(B, T, device=dev)
kpm[2, 9:] = float("-inf")
kpm[1, :] = float("-inf")
with torch.no_grad():
    mem = m._encode(src, kpm, True)
    print("memory nan rows:", torch.isnan(mem.view(B, T, d)).flatten(1).any(1).tolist())
    mem2 = m._encode(src, kpm, False)
    print("memory (not zero padded) nan rows:", torch.isnan(mem2.view(B, T, d)).flatten(1).any(1).tolist())
    out = m.generate(src, kpm)
    print("generate nan rows:", torch.isnan(out).flatten(1).any(1).tolist(), "frac nan row1", float(torch.isnan(out[1]).float().mean()))
    for sav in (False,):
        tm.GENERATE_SAVINGS = sav
        out = m.generate(src, kpm)
        print("generate (plain) nan rows:", torch.isnan(out).flatten(1).any(1).tolist())
    # attention op alone
    Q = torch.randn(1, B * T, d, device=dev); K = torch.randn(1, B * T, d, device=dev); Vv = torch.randn(1, B * T, d, device=dev)
    o = ops.Attention.apply(Q, K, Vv, None, kpm, B, h)
    print("Attention op nan rows:", torch.isnan(o.view(B, T, d)).flatten(1).any(1).tolist())
    ops.FUSED_ATTENTION = False
    o = ops.Attention.apply(Q, K, Vv, None, kpm, B, h)
    print("Attention (unfused) nan rows:", torch.isnan(o.view(B, T, d)).flatten(1).any(1).tolist())
