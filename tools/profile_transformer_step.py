"""One warm-up + N training steps (forward + masked loss + backward) of the transformer variant, for rocprofv3 runs.
usage: rocprofv3 --kernel-trace --stats -- python3 tools/profile_transformer_step.py [B] [T] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn  # noqa: E402
from artspeech_amd.phoneme_to_articulation.metrics import masked_euclidean_loss  # noqa: E402
from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
V, A, d, h, L, nf = 45, 11, 256, 4, 6, 100
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtSpeechTransformer(V, A, embed_dim=d, num_heads=h, num_layers=L, num_feat=nf).to(dev).eval()
batch = [(f"s{i}", torch.randint(1, V, (T,)), torch.rand(T, A, 2, nf // 2), ["p"] * T, torch.rand(T, 1, 2, nf // 2),
          torch.tensor([], dtype=torch.int), list(range(T)), torch.zeros(T)) for i in range(B)]
c = pad_sequence_transformer_collate_fn(batch)
tokens, targets, lengths = c[1].to(dev), c[2].to(dev), c[3]
shifted = torch.cat([torch.zeros(B, 1, A, nf, device=dev), targets[:, 1:].reshape(B, T - 1, A, nf)], dim=1)
kw = dict(src_key_padding_mask=c[8].to(dev), tgt_key_padding_mask=c[9].to(dev), src_attn_mask=c[10].to(dev), tgt_attn_mask=c[11].to(dev))
for _ in range(1 + steps):
    for p in model.parameters():
        p.grad = None
    loss = masked_euclidean_loss(model(tokens, shifted, **kw), targets, lengths)
    loss.backward()
torch.cuda.synchronize()
print("loss", float(loss))
