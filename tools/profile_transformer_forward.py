import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn
from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
B, T = 32, 200
V, A, d, h, L, nf = 45, 11, 256, 4, 6, 100
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtSpeechTransformer(V, A, embed_dim=d, num_heads=h, num_layers=L, num_feat=nf).to(dev).eval()
batch = [(f"s{i}", torch.randint(1, V, (T,)), torch.rand(T, A, 2, nf // 2), ["p"] * T, torch.rand(T, 1, 2, nf // 2),
          torch.tensor([], dtype=torch.int), list(range(T)), torch.zeros(T)) for i in range(B)]
c = pad_sequence_transformer_collate_fn(batch)
tokens, targets = c[1].to(dev), c[2].to(dev)
shifted = torch.cat([torch.zeros(B, 1, A, nf, device=dev), targets[:, 1:].reshape(B, T - 1, A, nf)], dim=1)
kw = dict(src_key_padding_mask=c[8].to(dev), tgt_key_padding_mask=c[9].to(dev), src_attn_mask=c[10].to(dev), tgt_attn_mask=c[11].to(dev))
with torch.no_grad():
    for _ in range(3):
        out = model(tokens, shifted, **kw)
torch.cuda.synchronize()
print("ok", float(out.sum()))
