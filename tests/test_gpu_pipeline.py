"""BASELINE configs[4] as a composition, on a small batch, against the CPU oracles CHAINED the same way:
phonemes -> contours (transformer variant, teacher forced) -> tract variables + area function of every frame ->
DeepSpeech2 articulatory scorer -> top-1 phoneme indices (reference: transformer/models.py:348-389,
tract_variables.py:73-125, area_function.py:124-159, phoneme_recognition/deepspeech2.py:90-195 with the input built as
phoneme_recognition/synthetic_shapes.py:133-135 does).  Each stage is checked twice: fed with the GPU's own upstream
output (stage parity, arg-min / arg-max decisions comparable) and end to end from the oracle's contours."""
import numpy as np
import pytest
import torch

from oracle import artspeech_oracle as O
from oracle import deepspeech2_oracle as DO
from oracle import transformer_oracle as TO

pytestmark = pytest.mark.gpu

ARTS = sorted(["arytenoid-cartilage", "epiglottis", "lower-incisor", "lower-lip", "pharynx", "soft-palate-midline",
               "thyroid-cartilage", "tongue", "upper-incisor", "upper-lip", "vocal-folds"])


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    return torch.device("cuda:0")


def test_pipeline_matches_chained_oracles(dev):
    from artspeech_amd.area_function import area_function_batched, evenly_spaced_fx_batched
    from artspeech_amd.phoneme_recognition import DeepSpeech2, top1_phonemes
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn
    from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
    from artspeech_amd.tract_variables import tract_variables_batched
    torch.manual_seed(11)
    V, A, d, h, L, nf = 19, len(ARTS), 32, 2, 1, 100
    N = nf // 2
    p2a = ArtSpeechTransformer(V, A, embed_dim=d, num_heads=h, num_layers=L, num_feat=nf)
    p2a_sd = {k: v.numpy().copy() for k, v in p2a.state_dict().items()}
    scorer_cfg = (2, 2, 1, 32, V, A * N, 40)   # planes, residual blocks, GRU layers, hidden, classes, features, adapter
    scorer = DeepSpeech2(2, 2, 1, 32, num_classes=V, num_features=A * N, adapter_out_features=40)
    with torch.no_grad():
        for mod in scorer.modules():
            if isinstance(mod, torch.nn.LayerNorm):
                mod.weight.uniform_(0.7, 1.3)
                mod.bias.uniform_(-0.2, 0.2)
    sc_sd = {k: v.numpy().copy() for k, v in scorer.state_dict().items()}
    p2a, scorer = p2a.to(dev).eval(), scorer.to(dev).eval()
    lens = [15, 15, 9]
    batch = [(f"s{i}", torch.randint(1, V, (l,)), torch.rand(l, A, 2, N), ["p"] * l, torch.rand(l, 1, 2, N),
              torch.tensor([], dtype=torch.int), list(range(l)), torch.zeros(l)) for i, l in enumerate(lens)]
    c = pad_sequence_transformer_collate_fn(batch)
    tokens, targets = c[1], c[2]
    B, T = tokens.shape
    shifted = torch.cat([torch.zeros(B, 1, A, nf), targets[:, 1:].reshape(B, T - 1, A, nf)], dim=1)

    # ---- stage 1: phonemes -> contours
    with torch.no_grad():
        contours = p2a(tokens.to(dev), shifted.to(dev), src_key_padding_mask=c[8].to(dev), tgt_key_padding_mask=c[9].to(dev),
                       src_attn_mask=c[10].to(dev), tgt_attn_mask=c[11].to(dev))
    o_contours = TO.forward(p2a_sd, (V, A, d, h, L, nf), tokens.numpy(), shifted.numpy(), c[10].numpy(), c[11].numpy(),
                            c[8].numpy(), c[9].numpy(), grad_mode=False)
    g_contours = contours.cpu().numpy()
    err = np.abs(g_contours - o_contours)
    assert (err <= 1e-4 * np.abs(o_contours) + 1e-6).all(), err.max()

    # ---- stage 2: tract variables of every frame (closest-point pairs: same indices as the oracle on the same contours)
    frames = contours.reshape(B * T, A, 2, N)
    tv, poc1, poc2, idx = tract_variables_batched(frames, ARTS)
    tv, idx = tv.cpu().numpy(), idx.cpu().numpy()
    for f in range(B * T):
        ov, op1, op2, oidx = O.tract_variables(g_contours.reshape(B * T, A, 2, N)[f], ARTS, dtype=np.float32)
        assert np.array_equal(idx[f], oidx), f
        assert np.abs(tv[f] - ov).max() < 1e-6
        assert np.array_equal(poc1[f].cpu().numpy(), op1) and np.array_equal(poc2[f].cpu().numpy(), op2)

    # ---- stage 3: area function + resampling; two predicted contours stand in for the tube walls (as tools/bench_pipeline.py)
    tongue, pharynx = ARTS.index("tongue"), ARTS.index("pharynx")
    air = torch.stack([contours[:, :, tongue], contours[:, :, pharynx]], dim=2).reshape(B * T, 2, 2, N).double()
    dists, fx = area_function_batched(air)
    af = evenly_spaced_fx_batched(dists, fx, 200)
    air_np = air.cpu().numpy()
    for f in range(0, B * T, 7):
        od, ofx = O.area_function(air_np[f, 0].T, air_np[f, 1].T)
        assert np.abs(dists[f].cpu().numpy() - od).max() < 1e-13 and np.abs(fx[f].cpu().numpy() - ofx).max() < 1e-13
        oxs, ofs = O.evenly_spaced_fx(od, ofx, 200)
        assert np.abs(af[f, 0].cpu().numpy() - oxs).max() < 1e-5 and np.abs(af[f, 1].cpu().numpy() - ofs).max() < 1e-5

    # ---- stage 4: scorer + top-1 on (B, 2, A*N, T) (synthetic_shapes.py:133-135: permute(2, 1, 3, 0) of a (T, A, 2, N) shape)
    x = contours.permute(0, 3, 2, 4, 1).reshape(B, 2, A * N, T)
    with torch.no_grad():
        logits = scorer(x)
    top = top1_phonemes(logits).cpu().numpy()
    want, _ = DO.forward(sc_sd, x.cpu().numpy(), None)
    assert np.abs(logits.cpu().numpy() - want).max() < 1e-4 * max(1.0, np.abs(want).max())
    srt = np.sort(want, -1)
    decided = (srt[..., -1] - srt[..., -2]) > 1e-4
    assert decided.mean() > 0.9
    assert np.array_equal(top[..., 0][decided], want.argmax(-1)[decided])   # bit-exact phoneme indices

    # ---- end to end: the oracle chain from ITS OWN contours decides the same phonemes wherever its margin is real
    ox = np.transpose(o_contours, (0, 3, 2, 4, 1)).reshape(B, 2, A * N, T).astype(np.float32)
    e2e, _ = DO.forward(sc_sd, ox, None)
    srt = np.sort(e2e, -1)
    decided = (srt[..., -1] - srt[..., -2]) > 1e-3
    assert decided.mean() > 0.8
    assert np.array_equal(top[..., 0][decided], e2e.argmax(-1)[decided])
    assert scorer_cfg[5] == x.shape[2]
