import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from artspeech_amd import _lib
import artspeech_amd.phoneme_to_articulation.encoder_decoder.models as M
from artspeech_amd.phoneme_to_articulation.metrics import masked_euclidean_loss
dev = torch.device("cuda:0")
L = _lib.lib()
FILL = [0.0]
_empty = torch.empty
def poisoned(*a, **k):
    t = _empty(*a, **k)
    if t.dtype == torch.float32 and t.numel() > 1000000:
        t.fill_(FILL[0])
    return t
M.torch.empty = poisoned
def run(mode, V, fill):
    FILL[0] = fill
    L.as_set_matrix_arith(mode)
    torch.manual_seed(4)
    A = 2
    model = M.ArtSpeech(V, A).to(dev)
    B, T = 8, 160
    lengths = np.array([160, 151, 133, 97, 64, 30, 7, 1])
    rng = np.random.RandomState(1)
    x = rng.randint(1, V, (B, T))
    tgt = rng.rand(B, T, A, 2, 50).astype(np.float32)
    for b, l in enumerate(lengths):
        x[b, l:] = 0
        tgt[b, l:] = 0
    out = model(torch.from_numpy(x).to(dev), torch.from_numpy(lengths))
    loss = masked_euclidean_loss(out, torch.from_numpy(tgt).to(dev), lengths)
    loss.backward()
    torch.cuda.synchronize()
    return {k: v.detach().cpu().numpy().copy() for k, v in model.named_grad_views().items()}, out.detach().cpu().numpy()
for V in (45, 100):
    g0, o0 = run(0, V, 0.0)
    for mode, fill in ((0, float('nan')), (1, 0.0), (1, float('nan')), (1, 1e30)):
        g1, o1 = run(mode, V, fill)
        bad = [k for k in g0 if not (np.abs(g0[k] - g1[k]).max() / (np.abs(g0[k]).max() + 1e-30) < 1e-5)]
        print("V", V, "mode", mode, "fill", fill, "out diff", np.abs(o0 - o1).max(), "bad tensors:", len(bad), bad[:4])
