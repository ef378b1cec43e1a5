"""GPU parity of the DeepSpeech2 articulatory scorer (inference) against fixtures produced by the reference itself and
against the numpy oracle: logits within 1e-4 (fp32), top-1 phoneme indices bit-exact."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, split_wg
from oracle import deepspeech2_oracle as DO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _model(cfg, w, dev):
    from artspeech_amd.phoneme_recognition import DeepSpeech2
    c = [int(v) for v in cfg]
    m = DeepSpeech2(c[0], c[1], c[2], c[3], num_classes=c[4], num_features=c[5], adapter_out_features=c[6] or None)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in w.items()}, strict=True)
    return m.to(dev).eval()


@pytest.mark.parametrize("name", ["deepspeech2_small", "deepspeech2_plain"])
def test_scorer_matches_reference_fixture(name, dev):
    from artspeech_amd.phoneme_recognition import top1_phonemes
    g = load_golden(name)
    w, _ = split_wg(g)
    m = _model(g["cfg"], w, dev)
    x = torch.from_numpy(g["x"]).to(dev)
    v = torch.from_numpy(g["voicing"]).to(dev) if "voicing" in g else None
    logits, feats = m(x, v, return_features=True)
    assert logits.shape == g["logits"].shape and feats.shape == g["features"].shape
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < 1e-4 * max(1.0, np.abs(g["logits"]).max())
    assert np.abs(feats.cpu().numpy() - g["features"]).max() < 1e-4
    assert np.array_equal(top1_phonemes(logits).cpu().numpy(), g["top"])  # bit-exact phoneme indices
    assert torch.equal(m(x, v), logits)                                   # deterministic, and the default return


def _random_state(cfg, seed):
    from artspeech_amd.phoneme_recognition import DeepSpeech2
    torch.manual_seed(seed)
    c = cfg
    m = DeepSpeech2(c[0], c[1], c[2], c[3], num_classes=c[4], num_features=c[5], adapter_out_features=c[6] or None)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.LayerNorm):
                mod.weight.uniform_(0.7, 1.3)
                mod.bias.uniform_(-0.2, 0.2)
    return {k: v.numpy() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("cfg,B,T,voiced", [
    ((2, 4, 2, 64, 44, 550, 80), 2, 61, True),    # the thesis scorer: 11 articulators x 50 points, adapter to 80 features
    ((2, 2, 3, 128, 31, 80, 0), 3, 130, False),   # LibriSpeech-style widths (H=128, no adapter); ragged tile edges
    ((1, 1, 1, 32, 5, 100, 0), 1, 1, False),      # one frame, one plane, D between the two register-resident LN widths
    ((3, 1, 1, 32, 5, 200, 0), 2, 7, True),       # D beyond the register-resident LN kernel and the LDS halo tile
    ((4, 1, 1, 32, 5, 24, 0), 2, 11, True),       # plane count without a specialised stem kernel
    ((1, 1, 1, 32, 7, 24, 40), 1, 17, False),     # ONE plane, ONE utterance through the adapter: the transposed input reshapes to a strided VIEW (round-3 sweep)
])
def test_scorer_matches_oracle(cfg, B, T, voiced, dev):
    w = _random_state(cfg, seed=sum(cfg) + T)
    m = _model(cfg, w, dev)
    rng = np.random.default_rng(T)
    x = rng.random((B, cfg[0], cfg[5], T), dtype=np.float32)
    v = (rng.random((B, T)) > 0.5).astype(np.float32) if voiced else None
    want, want_f = DO.forward(w, x, v)
    got, got_f = m(torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev) if voiced else None, return_features=True)
    got, got_f = got.cpu().numpy().astype(np.float64), got_f.cpu().numpy().astype(np.float64)
    assert np.abs(got_f - want_f).max() < 1e-4
    assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max())
    # top-1 indices are bit-exact wherever the oracle's own decision margin exceeds fp32 noise
    srt = np.sort(want, -1)
    decided = (srt[..., -1] - srt[..., -2]) > 1e-4
    assert decided.mean() > 0.9
    assert np.array_equal(got.argmax(-1)[decided], want.argmax(-1)[decided])


@pytest.mark.parametrize("seed", range(int(os.environ.get("AS_FUZZ_SEEDS", "10"))))
def test_scorer_random_configurations_vs_oracle(seed, dev):
    """Seeded random scorer architectures and inputs (1-4 input planes, 1-3 residual CNN and 1-3 GRU layers, hidden 32 / 64 / 128,
    3-60 classes, 8-300 features per plane with or without the adapter, 1-3 utterances of 1-150 frames, voicing or not)
    against the numpy oracle: features and logits within 1e-4, top-1 bit-exact where the oracle's margin exceeds fp32 noise
    (deepspeech2.py:90-195)."""
    r = np.random.RandomState(300 + seed)
    cfg = (int(r.randint(1, 5)), int(r.randint(1, 4)), int(r.randint(1, 4)), int(r.choice([32, 64, 128])), int(r.randint(3, 61)),
           int(r.choice([8, 24, 50, 80, 100, 130, 200, 300])), int(r.choice([0, 0, 40, 80])))
    B, T, voiced = int(r.randint(1, 4)), int(r.choice([1, 2, 17, 64, 65, 150])), bool(r.randint(0, 2))
    w = _random_state(cfg, seed=1000 + seed)
    m = _model(cfg, w, dev)
    x = r.rand(B, cfg[0], cfg[5], T).astype(np.float32)
    v = (r.rand(B, T) > 0.5).astype(np.float32) if voiced else None
    want, want_f = DO.forward(w, x, v)
    got, got_f = m(torch.from_numpy(x).to(dev), torch.from_numpy(v).to(dev) if voiced else None, return_features=True)
    got, got_f = got.cpu().numpy().astype(np.float64), got_f.cpu().numpy().astype(np.float64)
    assert got.shape == want.shape, (cfg, B, T)
    assert np.abs(got_f - want_f).max() < 1e-4, (cfg, B, T, voiced)
    assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max()), (cfg, B, T, voiced)
    srt = np.sort(want, -1)
    decided = (srt[..., -1] - srt[..., -2]) > 1e-4
    assert np.array_equal(got.argmax(-1)[decided], want.argmax(-1)[decided]), (cfg, B, T, voiced)


def test_scorer_accepts_strided_and_degenerate_views(dev):
    """The pipeline hands the scorer a permuted view of the contours (synthetic_shapes.py:133-135); sliced voicing; one plane,
    one utterance, one frame: all exactly what their dense copies give."""
    for cfg, B, T in (((2, 1, 1, 32, 9, 24, 16), 2, 9), ((1, 1, 1, 32, 9, 24, 16), 1, 9), ((1, 1, 1, 32, 9, 24, 0), 1, 1), ((3, 1, 1, 32, 9, 24, 16), 1, 1)):
        w = _random_state(cfg, seed=3)
        m = _model(cfg, w, dev)
        x = torch.rand(T, cfg[0], cfg[5], B, device=dev).permute(3, 1, 2, 0)      # (B, planes, D, T) view
        v = (torch.rand(B, 2 * T, device=dev) > 0.5).float()[:, ::2]
        assert torch.equal(m(x, v), m(x.contiguous(), v.contiguous())), (cfg, B, T)


def test_scorer_rejects_training_mode_and_cpu_inputs(dev):
    w = _random_state((2, 1, 1, 32, 5, 12, 0), 1)
    m = _model((2, 1, 1, 32, 5, 12, 0), w, dev)
    with pytest.raises(Exception):
        m(torch.zeros(1, 2, 12, 3))  # CPU tensor: no CPU path
    m.train()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 2, 12, 3, device=dev))
