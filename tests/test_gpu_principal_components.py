"""GPU parity of the RNNType switch (LSTM / GRU cells) and the principal-components recurrent model: outputs and every
parameter gradient against fixtures produced by the reference itself, larger cases against the numpy oracle, and the
backward at full size against a directional finite difference."""
import os

import numpy as np
import pytest
import torch

from conftest import assert_grad_close, load_golden, split_wg
from oracle import principal_components_oracle as PO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _model(V, comps, E, H, lstm, w, dev):
    from artspeech_amd.phoneme_to_articulation.principal_components.models import PrincipalComponentsArtSpeech
    m = PrincipalComponentsArtSpeech(V, comps, embed_dim=E, hidden_size=H, rnn="lstm" if lstm else "gru")
    if w is not None:
        m.load_state_dict({k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in w.items()}, strict=True)
    return m.to(dev)


@pytest.mark.parametrize("name", ["pc_lstm_small", "pc_gru_small"])
def test_matches_reference_fixture_forward_and_gradients(name, dev):
    g = load_golden(name)
    w, grads = split_wg(g)
    V, E, H, latent, lstm = (int(v) for v in g["cfg"])
    comps = {f"a{i}": int(c) for i, c in enumerate(g["comps"])}
    m = _model(V, comps, E, H, lstm, w, dev)
    tokens = torch.from_numpy(g["tokens"]).to(dev)
    out = m(tokens, torch.from_numpy(g["lengths"]))
    assert out.shape == g["out"].shape
    assert np.abs(out.detach().cpu().numpy() - g["out"]).max() < 1e-5
    (out * torch.from_numpy(g["dout"]).to(dev)).sum().backward()
    for k, p in m.named_parameters():
        assert_grad_close(p.grad.cpu().numpy(), grads[k], f"{name}: {k}")
    with torch.no_grad():  # inference path (no saved gates) gives the same numbers
        assert torch.equal(m(tokens, g["lengths"].tolist()), out.detach())


@pytest.mark.parametrize("lstm,H,B,T", [(True, 128, 6, 37), (False, 128, 3, 20), (True, 64, 2, 5)])
def test_matches_oracle(lstm, H, B, T, dev):
    torch.manual_seed(H + T)
    comps = {"tongue": 6, "lower-lip": 4, "pharynx": 2}
    m = _model(29, comps, 64, H, lstm, None, dev)
    w = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    lengths = sorted(np.random.default_rng(T).integers(1, T + 1, B).tolist(), reverse=True)
    lengths[0] = T
    tokens = torch.randint(1, 29, (B, T))
    want = PO.forward(w, tokens.numpy(), lengths, lstm)
    with torch.no_grad():
        got = m(tokens.to(dev), lengths).cpu().numpy()
    assert got.shape == want.shape and np.abs(got - want).max() < 2e-5


@pytest.mark.parametrize("seed", sorted(set(range(int(os.environ.get("AS_FUZZ_SEEDS", "10")))) | {93}))   # 93: ONE component (a one-row weight gradient)
def test_random_configurations_vs_oracle(seed, dev):
    """Seeded random principal-components models (LSTM or GRU cells, hidden 32 / 64 / 128, embedding widths 8-100, 1-4
    articulators with 1-12 components each, 1-7 utterances of 1-60 frames, ragged) against the numpy oracle
    (principal_components/models/rnn.py:58-109)."""
    r = np.random.RandomState(700 + seed)
    lstm, H, E, V = bool(r.randint(0, 2)), int(r.choice([32, 64, 128])), int(r.choice([8, 20, 64, 100])), int(r.randint(2, 60))
    _check_configuration(r, seed, lstm, H, E, V, dev)


@pytest.mark.parametrize("lstm", [True, False], ids=["lstm", "gru"])
@pytest.mark.parametrize("H", [20, 100, 260])
def test_any_hidden_size_vs_oracle(lstm, H, dev):
    """Hidden sizes outside {32, 64, 128} (nn.LSTM / nn.GRU take any): the plain recurrence kernels (lstm.hip / gru.hip,
    `*_generic_kernel`), forward and backward, against the same oracle and the same bounds."""
    r = np.random.RandomState(900 + H + int(lstm))
    _check_configuration(r, H, lstm, H, int(r.choice([8, 64])), int(r.randint(2, 60)), dev)


def _check_configuration(r, seed, lstm, H, E, V, dev):
    names = ["tongue", "lower-lip", "pharynx", "upper-lip"][:int(r.randint(1, 5))]
    comps = {n: int(r.randint(1, 13)) for n in names}
    B, T = int(r.randint(1, 8)), int(r.randint(1, 61))
    torch.manual_seed(seed)
    m = _model(V, comps, E, H, lstm, None, dev)
    w = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    lengths = sorted(r.randint(1, T + 1, B).tolist(), reverse=True)
    lengths[0] = T
    tokens = torch.from_numpy(r.randint(1, V, (B, T)) if V > 1 else np.zeros((B, T), np.int64))
    want = PO.forward(w, tokens.numpy(), lengths, lstm)
    out = m(tokens.to(dev), lengths)
    got = out.detach().cpu().numpy()
    what = (lstm, H, E, V, comps, B, T, lengths)
    assert got.shape == want.shape and np.abs(got - want).max() < 2e-5, what
    # backward (hand-written LSTM / GRU recurrences, trunk, predictor): <gradient, direction> against central differences of the
    # fp64 oracle along a random direction in parameter space (three step sizes, the nearest counts: a ReLU kink inside the step bends the quotient)
    wgt = torch.randn_like(out)
    (out * wgt).sum().backward()
    w64 = {k: np.asarray(v, np.float64) for k, v in w.items()}
    direction = {k: r.randn(*v.shape) * (np.abs(v).mean() + 1e-3) for k, v in w64.items()}
    analytic = sum(float((p.grad.cpu().numpy().astype(np.float64) * direction[k]).sum()) for k, p in m.named_parameters() if p.grad is not None)
    wg = wgt.cpu().numpy().astype(np.float64)

    def numeric_at(eps):
        f = lambda s: float((PO.forward({k: v + s * eps * direction[k] for k, v in w64.items()}, tokens.numpy(), lengths, lstm) * wg).sum())  # noqa: E731
        return (f(+1) - f(-1)) / (2 * eps)

    numeric = min((numeric_at(e) for e in (1e-6, 1e-7, 1e-8)), key=lambda v: abs(v - analytic))
    assert abs(analytic - numeric) <= 1e-3 * max(abs(numeric), 1.0), (what, analytic, numeric)


def test_raw_lstm_kernels_ragged_and_padded(dev):
    """as_lstm_bidir_fwd/bwd on their own: zeros at padded frames (outputs and gate gradients), batch independence."""
    from artspeech_amd import _lib
    from artspeech_amd.phoneme_to_articulation.rnn_ops import BiRNNLayer
    torch.manual_seed(3)
    B, T, I, H = 4, 11, 20, 32
    lengths = torch.tensor([11, 8, 3, 1], dtype=torch.int32, device=dev)
    x = torch.randn(B, T, I, device=dev, requires_grad=True)
    w_ih, w_hh = torch.randn(2, 4 * H, I, device=dev) * 0.2, torch.randn(2, 4 * H, H, device=dev) * 0.2
    b_ih, b_hh = torch.randn(2, 4 * H, device=dev) * 0.1, torch.randn(2, 4 * H, device=dev) * 0.1
    y = BiRNNLayer.apply(x, w_ih, w_hh, b_ih, b_hh, lengths, "lstm")
    for b, l in enumerate(lengths.tolist()):
        assert torch.all(y[b, l:] == 0)
    y.sum().backward()
    for b, l in enumerate(lengths.tolist()):
        assert torch.all(x.grad[b, l:] == 0)
    # utterance 1 alone gives the same rows (workgroups are independent)
    y1 = BiRNNLayer.apply(x[1:2].detach(), w_ih, w_hh, b_ih, b_hh, lengths[1:2].contiguous(), "lstm")
    assert torch.equal(y1[0], y[1].detach())
    # token-table form of the input projection (embedding folded into W_ih): same rows gathered inside the kernel
    V = 7
    table = torch.randn(V, 2, 4 * H, device=dev)
    tokens = torch.randint(0, V, (B, T), device=dev)
    y_tab, y_ref = torch.empty(B, T, 2 * H, device=dev), torch.empty(B, T, 2 * H, device=dev)
    gi = table[tokens].reshape(B * T, 2, 4 * H).contiguous()
    L = _lib.lib()
    _lib.check(L.as_lstm_bidir_fwd(_lib.ptr(table), _lib.ptr(tokens), T, _lib.ptr(w_hh), _lib.ptr(b_hh), _lib.ptr(lengths), B, T, H,
                                   _lib.ptr(y_tab), None, _lib.stream_ptr()), "as_lstm_bidir_fwd")
    _lib.check(L.as_lstm_bidir_fwd(_lib.ptr(gi), None, 0, _lib.ptr(w_hh), _lib.ptr(b_hh), _lib.ptr(lengths), B, T, H, _lib.ptr(y_ref), None,
                                   _lib.stream_ptr()), "as_lstm_bidir_fwd")
    assert torch.equal(y_tab, y_ref)


@pytest.mark.parametrize("lstm", [True, False])
def test_full_size_backward_matches_finite_difference(lstm, dev):
    """B=32, T=200, H=128: directional derivative of sum(out * dout) along a random parameter direction."""
    torch.manual_seed(7)
    m = _model(45, {"a": 6, "b": 6}, 64, 128, lstm, None, dev)
    B, T = 32, 200
    lengths = torch.linspace(200, 60, B).int().tolist()
    tokens = torch.randint(1, 45, (B, T), device=dev)
    dout = torch.rand(B, T, m.latent_size, device=dev) / (B * T)
    loss = (m(tokens, lengths) * dout).sum()
    loss.backward()
    params = [p for p in m.parameters()]
    dirs = [torch.randn_like(p) for p in params]
    norm = float(sum((d.double() ** 2).sum() for d in dirs)) ** 0.5
    dirs = [d / norm for d in dirs]  # unit-norm direction over all parameters
    analytic = sum(float((p.grad.double() * d.double()).sum()) for p, d in zip(params, dirs))
    eps = 1e-2
    vals = []
    with torch.no_grad():
        for sgn in (1.0, -1.0):
            for p, d in zip(params, dirs):
                p.add_(sgn * eps * d)
            vals.append(float((m(tokens, lengths).double() * dout.double()).sum()))
            for p, d in zip(params, dirs):
                p.sub_(sgn * eps * d)
    numeric = (vals[0] - vals[1]) / (2 * eps)
    assert abs(numeric - analytic) < 1e-2 * max(abs(analytic), 1e-3) + 2e-5, (numeric, analytic)


def test_autoencoder_matches_reference_fixture_forward_and_gradients(dev):
    from artspeech_amd.phoneme_to_articulation.principal_components.models import MultiArticulatorAutoencoder
    g = load_golden("pc_autoencoder")
    w, grads = split_wg(g)
    comps = dict(zip(["tongue", "lower-lip", "upper-lip"], (int(c) for c in g["comps"])))
    m = MultiArticulatorAutoencoder(in_features=20, indices_dict=comps, hidden_features=16)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    m.to(dev)
    out, latent = m(torch.from_numpy(g["x"]).to(dev))
    assert np.abs(out.detach().cpu().numpy() - g["out"]).max() < 2e-6
    assert np.abs(latent.detach().cpu().numpy() - g["latent"]).max() < 2e-6
    ((out * torch.from_numpy(g["dout"]).to(dev)).sum() + (latent * torch.from_numpy(g["dlat"]).to(dev)).sum()).backward()
    for k, p in m.named_parameters():
        assert np.abs(p.grad.cpu().numpy() - grads[k]).max() < 1e-5 * max(1.0, np.abs(grads[k]).max()), k


def test_critical_loss_matches_reference_fixture_and_oracle(dev):
    from artspeech_amd.phoneme_to_articulation.principal_components.losses import CriticalLoss
    c = load_golden("pc_critical_loss")
    arts = ["lower-lip", "tongue", "upper-lip"]
    crit = CriticalLoss(["TTCD", "LA"], list(arts))
    shapes = torch.from_numpy(c["shapes"]).to(dev).requires_grad_(True)
    loss = crit(shapes, torch.from_numpy(c["targets"]).to(dev), torch.from_numpy(c["ref"]).to(dev), torch.from_numpy(c["mask"]).to(dev))
    assert abs(float(loss.detach()) - float(c["loss"])) < 1e-7
    loss.backward()
    assert np.abs(shapes.grad.cpu().numpy() - c["dshapes"]).max() < 1e-6
    assert float(CriticalLoss([], arts)(shapes, shapes, None, None)) == 0.0
    # 50-point contours, four variables, against the oracle (the reference's cdist would use its matmul expansion here)
    rng = np.random.default_rng(5)
    arts = ["lower-lip", "pharynx", "soft-palate", "tongue", "upper-incisor", "upper-lip"]
    big = rng.random((3, 7, 6, 2, 50), dtype=np.float32)
    mask = (rng.random((3, 4, 7)) > 0.3).astype(np.float32)
    got = CriticalLoss(["LA", "TTCD", "TBCD", "VEL"], arts)(torch.from_numpy(big).to(dev), torch.from_numpy(big).to(dev), None,
                                                             torch.from_numpy(mask).to(dev))
    want = PO.critical_loss(big, None, mask, ["LA", "TTCD", "TBCD", "VEL"], arts)
    assert abs(float(got) - want) < 1e-6


def test_wrapper_decodes_components_to_shapes(dev):
    """PrincipalComponentsArtSpeechWrapper: rnn -> MultiDecoder -> per-articulator denormalisation."""
    from artspeech_amd.phoneme_to_articulation.principal_components.models import (MultiDecoder, PrincipalComponentsArtSpeech,
                                                                                    PrincipalComponentsArtSpeechWrapper)
    torch.manual_seed(9)
    comps = {"tongue": 4, "lower-lip": 3}
    rnn = PrincipalComponentsArtSpeech(13, comps, embed_dim=16, hidden_size=32, rnn="lstm").to(dev)
    dec = MultiDecoder(comps, in_features=20, hidden_features=16).to(dev)
    denorm = {"tongue": lambda a: a * 2.0 + 0.5, "lower-lip": lambda a: a - 1.0}
    model = PrincipalComponentsArtSpeechWrapper(rnn, dec, denorm).eval()
    tokens, lengths = torch.randint(1, 13, (3, 6), device=dev), [6, 4, 2]
    with torch.no_grad():
        out = model(tokens, lengths)
        z = rnn(tokens, lengths)
    assert out.shape == (3, 6, 2, 2, 10)
    w = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in dec.state_dict().items()}
    zz = z.cpu().numpy().astype(np.float64)
    for i, (name, idx, f) in enumerate((("lower-lip", [4, 5, 6], lambda a: a - 1.0), ("tongue", [0, 1, 2, 3], lambda a: a * 2.0 + 0.5))):
        want = f(PO._mlp(zz[..., idx], w, f"decoders.{name}.decoder.").reshape(3, 6, 2, 10))
        assert np.abs(out[:, :, i].cpu().numpy() - want).max() < 1e-5
