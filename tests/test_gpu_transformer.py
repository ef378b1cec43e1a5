"""GPU parity of the transformer variant (inference) against fixtures produced by the reference itself and
against the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import assert_grad_close, load_golden, split_wg
from oracle import transformer_oracle as TO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def small(dev):
    from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
    g = load_golden("transformer_small")
    w, _ = split_wg(g)
    V, A, d, h, L, nf = (int(v) for v in g["cfg"])
    model = ArtSpeechTransformer(V, A, embed_dim=d, num_heads=h, num_layers=L, num_feat=nf)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    return model.to(dev).eval(), g, w


def _t(x, dev, dtype=torch.float32):
    return torch.as_tensor(x, dtype=dtype).to(dev)


def test_forward_matches_reference_fixture(small, dev):
    model, g, _ = small
    args = dict(src_key_padding_mask=_t(g["src_kpm"], dev), tgt_key_padding_mask=_t(g["tgt_kpm"], dev),
                src_attn_mask=_t(g["src_mask"], dev), tgt_attn_mask=_t(g["tgt_mask"], dev))
    src, tgt = _t(g["tokens"], dev, torch.int64), _t(g["shifted"], dev)
    with torch.no_grad():  # how the reference's evaluation runs it: encoder fast path, zeros at padded sources
        out = model(src, tgt, **args)
    assert out.shape == g["out_nograd"].shape
    err = np.abs(out.cpu().numpy() - g["out_nograd"])
    assert (err <= 1e-4 * np.abs(g["out_nograd"]) + 1e-6).all(), err.max()
    out = model(src, tgt, **args)  # grad enabled + trainable parameters: standard encoder path (no zeroing)
    err = np.abs(out.detach().cpu().numpy() - g["out_grad"])
    assert (err <= 1e-4 * np.abs(g["out_grad"]) + 1e-6).all(), err.max()


def test_encoder_matches_reference_fixture(small, dev):
    model, g, _ = small
    src, kpm = _t(g["tokens"], dev, torch.int64), _t(g["src_kpm"], dev)
    for zero, key in ((True, "enc_nograd"), (False, "enc_grad")):
        mem = model._encode(src, kpm, zero_padded=zero).view(g[key].shape)
        assert np.abs(mem.detach().cpu().numpy() - g[key]).max() < 5e-6, key


def test_generate_matches_reference_fixture(small, dev):
    model, g, w = small
    src, kpm = _t(g["tokens"], dev, torch.int64), _t(g["src_kpm"], dev)
    gen = model.generate(src, kpm)
    assert gen.shape == g["gen"].shape
    ref = g["gen"]
    # autoregressive feedback amplifies fp32 rounding frame after frame (the fp64 oracle drifts from the
    # reference's own fp32 run by 1.2e-4 at frame 6 too): tight on the first frames, loose later
    e = np.abs(gen.cpu().numpy() - ref)
    assert e[:, 0].max() < 1e-5 and e[:, :3].max() < 1e-4 and e.max() < 2e-3, [float(e[:, t].max()) for t in range(e.shape[1])]
    # teacher-forced single step: the reference's own generated prefix in, next frame out
    B, T, A = ref.shape[:3]
    prefix = torch.cat([torch.zeros(B, 1, A, ref.shape[3] * ref.shape[4], device=dev),
                        _t(ref.reshape(B, T, A, -1), dev)[:, :T - 1]], dim=1)
    with torch.no_grad():
        mem = model._encode(src, kpm, zero_padded=True)
        step = model._generate_one_step(prefix, mem, memory_key_padding_mask=kpm)
    err = np.abs(step[:, -1].cpu().numpy() - ref[:, -1])
    assert err.max() < 2e-5, err.max()


def test_generate_savings_are_exact(small, dev):
    """Memory-side K/V computed once + last layer restricted to the newest frame == re-decoding everything."""
    from artspeech_amd.phoneme_to_articulation.transformer import models as M
    model, g, _ = small
    src, kpm = _t(g["tokens"], dev, torch.int64), _t(g["src_kpm"], dev)
    try:
        M.GENERATE_SAVINGS = False
        plain = model.generate(src, kpm)
        M.GENERATE_SAVINGS = True
        fast = model.generate(src, kpm)
    finally:
        M.GENERATE_SAVINGS = True
    assert (plain - fast).abs().max() < 2e-6  # same arithmetic per element; only the GEMM tile shapes differ


def test_medium_config_vs_oracle(dev):
    """d=64, 4 heads, 2 layers, A=4, T=24, ragged: against the fp64 oracle (both encoder modes)."""
    from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn
    torch.manual_seed(3)
    V, A, d, h, L, nf = 17, 4, 64, 4, 2, 100
    model = ArtSpeechTransformer(V, A, embed_dim=d, num_heads=h, num_layers=L, num_feat=nf)
    with torch.no_grad():  # non-trivial LayerNorm affines
        for k, v in model.named_views().items():
            if k.endswith("bias") and v.dim() == 1:
                v.uniform_(-0.2, 0.2)
    sd = {k: v.numpy().copy() for k, v in model.state_dict().items()}
    model = model.to(dev).eval()
    lens = [24, 17, 9]
    batch = [(f"s{i}", torch.randint(1, V, (l,)), torch.rand(l, A, 2, nf // 2), ["p"] * l, torch.rand(l, 1, 2, nf // 2),
              torch.tensor([], dtype=torch.int), list(range(l)), torch.zeros(l)) for i, l in enumerate(lens)]
    c = pad_sequence_transformer_collate_fn(batch)
    tokens, targets = c[1], c[2]
    B, T = tokens.shape
    shifted = torch.cat([torch.zeros(B, 1, A, nf), targets[:, 1:].reshape(B, T - 1, A, nf)], dim=1)
    for grad_mode in (False, True):
        model.set_encoder_grad_mode(grad_mode)
        out = model(tokens.to(dev), shifted.to(dev), src_key_padding_mask=c[8].to(dev), tgt_key_padding_mask=c[9].to(dev),
                    src_attn_mask=c[10].to(dev), tgt_attn_mask=c[11].to(dev))
        ref = TO.forward(sd, (V, A, d, h, L, nf), tokens.numpy(), shifted.numpy(), c[10].numpy(), c[11].numpy(), c[8].numpy(),
                         c[9].numpy(), grad_mode=grad_mode)
        err = np.abs(out.detach().cpu().numpy() - ref)
        assert (err <= 1e-4 * np.abs(ref) + 1e-6).all(), (grad_mode, err.max())
    model.set_encoder_grad_mode(None)


@pytest.mark.parametrize("seed", range(int(os.environ.get("AS_FUZZ_SEEDS", "10"))))
def test_random_configurations_forward_vs_oracle(dev, seed):
    """Seeded random transformer variants (2-6 channels, widths 12-96 with 1-5 heads incl. odd head widths, 1-3 layers, 2-40 features per channel, 1-4
    ragged utterances of 1-40 frames, both encoder modes) through the trainer's call (train_..._transformer.py:104-111) against the
    fp64 oracle: contours within 1e-4 relative."""
    from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn
    r = np.random.RandomState(4000 + seed)
    A, L, V = int(r.randint(2, 7)), int(r.randint(1, 4)), int(r.randint(3, 50))
    d, h = [(16, 1), (16, 4), (32, 2), (48, 4), (64, 4), (64, 1), (96, 2), (96, 3), (24, 4), (20, 2), (12, 4), (28, 4), (20, 5), (36, 3)][int(r.randint(0, 14))]
    nf = 2 * int(r.randint(1, 21))
    lens = sorted((int(v) for v in r.randint(1, 41, int(r.randint(1, 5)))), reverse=True)
    torch.manual_seed(seed)
    model = ArtSpeechTransformer(V, A, embed_dim=d, num_heads=h, num_layers=L, num_feat=nf)
    with torch.no_grad():
        for k, v in model.named_views().items():
            if k.endswith("bias") and v.dim() == 1:
                v.uniform_(-0.2, 0.2)
    sd = {k: v.numpy().copy() for k, v in model.state_dict().items()}
    model = model.to(dev).eval()
    batch = [(f"s{i}", torch.randint(1, V, (l,)), torch.rand(l, A, 2, nf // 2), ["p"] * l, torch.rand(l, 1, 2, nf // 2),
              torch.tensor([], dtype=torch.int), list(range(l)), torch.zeros(l)) for i, l in enumerate(lens)]
    c = pad_sequence_transformer_collate_fn(batch)
    tokens, targets = c[1], c[2]
    B, T = tokens.shape
    shifted = torch.cat([torch.zeros(B, 1, A, nf), targets[:, 1:].reshape(B, T - 1, A, nf)], dim=1)
    for grad_mode in (False, True):
        model.set_encoder_grad_mode(grad_mode)
        out = model(tokens.to(dev), shifted.to(dev), src_key_padding_mask=c[8].to(dev), tgt_key_padding_mask=c[9].to(dev),
                    src_attn_mask=c[10].to(dev), tgt_attn_mask=c[11].to(dev))
        ref = TO.forward(sd, (V, A, d, h, L, nf), tokens.numpy(), shifted.numpy(), c[10].numpy(), c[11].numpy(), c[8].numpy(),
                         c[9].numpy(), grad_mode=grad_mode)
        err = np.abs(out.detach().cpu().numpy() - ref)
        assert out.shape == ref.shape and (err <= 1e-4 * np.abs(ref) + 1e-6).all(), ((A, d, h, L, nf, lens), grad_mode, err.max())
    model.set_encoder_grad_mode(None)
    if T <= 12:   # free-running generation (transformer/models.py:391-427): the first frames tight, feedback amplifies later ones
        gen = model.generate(tokens.to(dev), c[8].to(dev)).cpu().numpy()
        ref = TO.generate(sd, (V, A, d, h, L, nf), tokens.numpy(), c[8].numpy())
        e = np.abs(gen - ref)
        assert gen.shape == ref.shape and e[:, 0].max() < 1e-5 and e[:, :3].max() < 1e-4, ((A, d, h, L, nf, lens), e[:, :3].max())
        # later frames: the feedback loop amplifies rounding without bound in some draws (3 of 300 reach 4e-3 .. 4e-2 by frame 12,
        # from 1e-6 at frame 1), so the last frame is checked teacher-forced: the oracle's own prefix in, next frame out
        prefix = torch.cat([torch.zeros(B, 1, A, nf), torch.from_numpy(ref.reshape(B, T, A, nf)[:, :T - 1]).float()], dim=1).to(dev)
        with torch.no_grad():
            mem = model._encode(tokens.to(dev), c[8].to(dev), zero_padded=True)
            step = model._generate_one_step(prefix, mem, memory_key_padding_mask=c[8].to(dev))
        err = np.abs(step[:, -1].cpu().numpy() - ref[:, -1])
        assert (err <= 1e-4 * np.abs(ref[:, -1]) + 1e-6).all(), ((A, d, h, L, nf, lens), err.max())


def test_forward_accepts_strided_and_degenerate_views(dev):
    """Row-sliced / step-sliced / permuted tokens, targets and masks, and batches with B = 1 or T = 1 (where reshapes of
    permuted tensors are strided views), give exactly what their dense copies give."""
    from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
    torch.manual_seed(1)
    V, A, d, h, L, nf = 11, 3, 32, 2, 2, 12
    model = ArtSpeechTransformer(V, A, embed_dim=d, num_heads=h, num_layers=L, num_feat=nf).to(dev).eval()
    for B, T in ((3, 7), (1, 6), (4, 1), (1, 1)):
        src = torch.randint(1, V, (B, 2 * T + 1), device=dev)[:, 1:2 * T:2]
        tgt = torch.rand(B, A, T, nf, device=dev).permute(0, 2, 1, 3)
        kpm = torch.zeros(T, B, device=dev).t()
        causal = torch.full((T, T), float("-inf"), device=dev).triu(1)
        mask = causal.expand(B, T, T)
        with torch.no_grad():
            got = model(src, tgt, src_key_padding_mask=kpm, tgt_key_padding_mask=kpm, src_attn_mask=mask, tgt_attn_mask=mask)
            want = model(src.contiguous(), tgt.contiguous(), src_key_padding_mask=kpm.contiguous(), tgt_key_padding_mask=kpm.contiguous(),
                         src_attn_mask=mask.contiguous(), tgt_attn_mask=mask.contiguous())
            assert torch.equal(got, want), (B, T)
            assert torch.equal(model.generate(src, kpm), model.generate(src.contiguous(), kpm.contiguous())), (B, T)


def _drawn_training_configs(n):
    """seeded random (A, d, h, lens, nf) for the backward check below: 2-5 channels, widths 12-64 incl. odd head widths, 1-3
    ragged utterances of 1-14 frames, 2-24 features per channel"""
    out = []
    for s in range(n):
        r = np.random.RandomState(6000 + s)
        d, h = [(12, 4), (16, 1), (20, 5), (28, 4), (32, 2), (48, 4), (64, 4)][int(r.randint(0, 7))]
        lens = sorted((int(v) for v in r.randint(1, 15, int(r.randint(1, 4)))), reverse=True)
        out.append((int(r.randint(2, 6)), d, h, lens, 2 * int(r.randint(1, 13))))
    return out


@pytest.mark.parametrize("A,d,h,lens,nf", [(2, 32, 2, [9, 5], 20), (3, 48, 4, [12, 12, 7], 20), (5, 64, 2, [20, 3], 20), (3, 32, 2, [11, 4], 6),
                                           (2, 64, 4, [7, 7], 10), (3, 28, 4, [9, 6], 6), (2, 12, 4, [8], 2), (3, 20, 2, [5, 5, 2], 10)]
                         + _drawn_training_configs(int(os.environ.get("AS_FUZZ_SEEDS", "8"))))
def test_edge_configs_vs_oracle_with_directional_derivative(dev, A, d, h, lens, nf):
    """Corners of the block-group node (ops.ChannelBlocks): two channels (ONE interaction block per channel: the
    concatenation is a single block wide), a head width the fused attention kernel does not take (48 / 4 = 12: the unfused
    GEMM + softmax path inside the node), d not a multiple of 32 (partial ReLU-bit words) -- forward against the fp64 oracle,
    backward against a central difference of the oracle's loss along a random parameter direction."""
    from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn
    torch.manual_seed(A * 100 + d)
    V, L = 13, 2
    cfg = (V, A, d, h, L, nf)
    model = ArtSpeechTransformer(V, A, embed_dim=d, num_heads=h, num_layers=L, num_feat=nf)
    with torch.no_grad():
        for k, v in model.named_views().items():
            if k.endswith("bias") and v.dim() == 1:
                v.uniform_(-0.2, 0.2)
            if k == "tgt_embedding.1.weight":
                # as in the reference-generated fixtures (make_golden.perturb_by_key): at the default initialisation two
                # decoder layers of this width amplify fp32 rounding into per-cent level gradient differences between ANY two
                # evaluations (measured against this oracle: 1-5 % at L = 2, 1e-6 at L = 1 or with this scaling)
                v.mul_(0.1)
    sd = {k: v.numpy().copy().astype(np.float64) for k, v in model.state_dict().items()}
    model = model.to(dev).eval()
    model.set_encoder_grad_mode(True)
    batch = [(f"s{i}", torch.randint(1, V, (l,)), torch.rand(l, A, 2, nf // 2), ["p"] * l, torch.rand(l, 1, 2, nf // 2),
              torch.tensor([], dtype=torch.int), list(range(l)), torch.zeros(l)) for i, l in enumerate(lens)]
    c = pad_sequence_transformer_collate_fn(batch)
    tokens, targets = c[1], c[2]
    B, T = tokens.shape
    shifted = torch.cat([torch.zeros(B, 1, A, nf), targets[:, 1:].reshape(B, T - 1, A, nf)], dim=1)
    args = (tokens.numpy(), shifted.numpy(), c[10].numpy(), c[11].numpy(), c[8].numpy(), c[9].numpy())
    out = model(tokens.to(dev), shifted.to(dev), src_key_padding_mask=c[8].to(dev), tgt_key_padding_mask=c[9].to(dev),
                src_attn_mask=c[10].to(dev), tgt_attn_mask=c[11].to(dev))
    ref = TO.forward(sd, cfg, *args, grad_mode=True)
    err = np.abs(out.detach().cpu().numpy() - ref)
    assert (err <= 1e-4 * np.abs(ref) + 1e-6).all(), err.max()
    wgt = torch.randn_like(out)
    (out * wgt).sum().backward()
    rng = np.random.RandomState(7)
    direction = {k: rng.randn(*v.shape) * (np.abs(v).mean() + 1e-3) for k, v in sd.items() if not k.endswith(".pe")}
    analytic = sum(float((g.cpu().numpy().astype(np.float64) * direction[k]).sum()) for k, g in model.named_grad_views().items())
    w64 = wgt.cpu().numpy().astype(np.float64)
    # central differences of the fp64 oracle at two step sizes: a ReLU kink inside [-eps, +eps] along the joint direction bends
    # the difference quotient at that step size (one draw of the round-3 sweep: -20.40220 at 1e-8, -20.83 at 1e-7, analytic
    # -20.40221), so the nearer of the two is the yardstick; 1e-5 is 1 % off in most draws
    def numeric_at(eps):
        def loss_at(sign):
            moved = {k: (v + sign * eps * direction[k] if k in direction else v) for k, v in sd.items()}
            return float((TO.forward(moved, cfg, *args, grad_mode=True) * w64).sum())
        return (loss_at(+1) - loss_at(-1)) / (2 * eps)

    numeric = min((numeric_at(e) for e in (1e-7, 1e-8)), key=lambda v: abs(v - analytic))
    # (a wrong term in the hand-written backward shows up at the per-cent level at EVERY step size; fp32 rounding of the analytic
    # side stays below 5e-4: measured 4e-6 .. 4.5e-4 over the first three configurations)
    assert abs(analytic - numeric) <= 1e-3 * max(abs(numeric), 1.0), (analytic, numeric)
    model.set_encoder_grad_mode(None)


def test_full_size_forward_properties(dev):
    """BASELINE configs[3] at full size (d=256, 6 layers, 11 articulators, B=32, T=200, ragged lengths): properties that do not
    need an oracle run -- bitwise run-to-run determinism, batch independence (an utterance's contours do not depend on its
    batch mates), padding invariance (frames beyond an utterance's length do not influence the valid ones), range."""
    from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn
    torch.manual_seed(11)
    V, A, d, h, L, nf = 45, 11, 256, 4, 6, 100
    B, T = 32, 200
    model = ArtSpeechTransformer(V, A, embed_dim=d, num_heads=h, num_layers=L, num_feat=nf).to(dev).eval()
    lens = torch.linspace(T, 60, B).int().tolist()

    def make(idx):
        g = torch.Generator().manual_seed(5)
        items = [(f"s{i}", torch.randint(1, V, (l,), generator=g), torch.rand(l, A, 2, nf // 2, generator=g), ["p"] * l,
                  torch.rand(l, 1, 2, nf // 2, generator=g), torch.tensor([], dtype=torch.int), list(range(l)), torch.zeros(l))
                 for i, l in enumerate(lens)]
        c = pad_sequence_transformer_collate_fn([items[i] for i in idx])
        tokens, targets = c[1].to(dev), c[2].to(dev)
        b, t = tokens.shape
        shifted = torch.cat([torch.zeros(b, 1, A, nf, device=dev), targets[:, 1:].reshape(b, t - 1, A, nf)], dim=1)
        kw = dict(src_key_padding_mask=c[8].to(dev), tgt_key_padding_mask=c[9].to(dev), src_attn_mask=c[10].to(dev),
                  tgt_attn_mask=c[11].to(dev))
        return tokens, shifted, kw, c[3]
    with torch.no_grad():
        tokens, shifted, kw, lengths = make(range(B))
        out = model(tokens, shifted, **kw)
        assert out.shape == (B, T, A, 2, nf // 2)
        assert torch.equal(out, model(tokens, shifted, **kw))                          # deterministic, bit for bit
        valid = torch.arange(T, device=dev)[None, :] < torch.as_tensor(lengths, device=dev)[:, None]
        ov = out[valid]
        assert torch.isfinite(ov).all() and ov.min() >= 0 and ov.max() <= 1            # sigmoid contours
        # batch independence: four of the utterances alone (the collate re-sorts by length: same relative order here)
        sub = [3, 10, 20, 31]
        tokens2, shifted2, kw2, lengths2 = make(sub)
        out2 = model(tokens2, shifted2, **kw2)
        for j, i in enumerate(sub):
            l = int(lengths2[j])
            assert int(lengths[i]) == l
            assert (out2[j, :l] - out[i, :l]).abs().max().item() <= 2e-5, (i, (out2[j, :l] - out[i, :l]).abs().max().item())
        # padding invariance: garbage in the padded decoder inputs of the shorter utterances changes nothing valid
        shifted3 = shifted.clone()
        pad = ~valid
        shifted3[pad] = 123.0
        out3 = model(tokens, shifted3, **kw)
        assert (out3[valid] - out[valid]).abs().max().item() <= 1e-6


def test_full_size_training_step_directional_derivative(dev):
    """configs[3] at full size, training path (fused attention forward with key-major probabilities, dS kernel, grouped
    GEMM backward): the gradient along its own direction against a central finite difference of the masked loss, and bitwise
    determinism of loss and gradients."""
    from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn
    from artspeech_amd.phoneme_to_articulation.metrics import masked_euclidean_loss
    torch.manual_seed(12)
    V, A, d, h, L, nf = 45, 11, 256, 4, 6, 100
    B, T = 32, 200
    model = ArtSpeechTransformer(V, A, embed_dim=d, num_heads=h, num_layers=L, num_feat=nf).to(dev).eval()
    lens = torch.linspace(T, 60, B).int().tolist()
    items = [(f"s{i}", torch.randint(1, V, (l,)), torch.rand(l, A, 2, nf // 2), ["p"] * l, torch.rand(l, 1, 2, nf // 2),
              torch.tensor([], dtype=torch.int), list(range(l)), torch.zeros(l)) for i, l in enumerate(lens)]
    c = pad_sequence_transformer_collate_fn(items)
    tokens, targets, lengths = c[1].to(dev), c[2].to(dev), c[3]
    shifted = torch.cat([torch.zeros(B, 1, A, nf, device=dev), targets[:, 1:].reshape(B, T - 1, A, nf)], dim=1)
    kw = dict(src_key_padding_mask=c[8].to(dev), tgt_key_padding_mask=c[9].to(dev), src_attn_mask=c[10].to(dev), tgt_attn_mask=c[11].to(dev))
    params = [p for p in model.parameters() if p.requires_grad]

    def loss_and_grads():
        for p in params:
            p.grad = None
        loss = masked_euclidean_loss(model(tokens, shifted, **kw), targets, lengths)
        loss.backward()
        return loss.item(), [p.grad.clone() for p in params]
    l0, g0 = loss_and_grads()
    l1, g1 = loss_and_grads()
    assert l0 == l1 and all(torch.equal(a, b) for a, b in zip(g0, g1))                # deterministic, bit for bit
    assert all(torch.isfinite(g).all() for g in g0)
    gnorm = torch.sqrt(sum((g.double() ** 2).sum() for g in g0)).item()
    assert gnorm > 0

    def loss_at(sign):
        with torch.no_grad():
            for p, g in zip(params, g0):
                p.add_(g, alpha=sign * eps)
            val = masked_euclidean_loss(model(tokens, shifted, **kw), targets, lengths).item()
            for p, g in zip(params, g0):
                p.add_(g, alpha=-sign * eps)
        return val
    # the loss moves by ~ +-1e-4 (far above its fp32 rounding, ~3e-8); a 16 times longer step is already 17 % off: curvature
    eps = 1.25e-4 / gnorm
    fd = (loss_at(+1) - loss_at(-1)) / (2 * eps)                                       # = |grad|^2 along the gradient
    assert abs(fd - gnorm ** 2) <= 0.03 * gnorm ** 2, (fd, gnorm ** 2)


# ------------------------------------------------------------------------------------------- training (backward)
def test_backward_matches_reference_fixture(small, dev):
    """eval mode + gradient tracking (dropout off, standard encoder path): d(sum(out * dout)) / d(every parameter)
    against the gradients the reference's autograd produced for the same weights and inputs."""
    model, g, _ = small
    _, grads = split_wg(g)
    model.zero_grad()
    args = dict(src_key_padding_mask=_t(g["src_kpm"], dev), tgt_key_padding_mask=_t(g["tgt_kpm"], dev),
                src_attn_mask=_t(g["src_mask"], dev), tgt_attn_mask=_t(g["tgt_mask"], dev))
    out = model(_t(g["tokens"], dev, torch.int64), _t(g["shifted"], dev), **args)
    assert np.abs(out.detach().cpu().numpy() - g["out_grad"]).max() < 5e-6
    (out * _t(g["dout"], dev)).sum().backward()
    gv = model.named_grad_views()
    assert set(gv) == set(grads), set(gv) ^ set(grads)
    worst = ("", 0.0)
    for k, ref in grads.items():
        got = gv[k].cpu().numpy()
        err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-6)
        worst = max(worst, (k, float(err)), key=lambda t: t[1])
        assert err < 1e-3, (k, err)
    print("worst gradient relative error:", worst)
    model.zero_grad()


def test_split_forward_gemms_opt_in_matches_reference_fixture(small, dev):
    """ARTSPEECH_GEMM_PRECISION=lib / set_gemm_precision("lib"): the forward linears (with their ReLU bit images, grouped
    offsets) on the bf16 matrix instruction with both operands split in the kernel (as_gemm.precision = 3, gemm_s6.hip).  The
    small reference fixture is reproduced at the tolerances of the default path, forward and backward."""
    from artspeech_amd.phoneme_to_articulation.transformer import ops
    model, g, _ = small
    _, grads = split_wg(g)
    keep = ops.GEMM_PRECISION
    ops.set_gemm_precision("lib")
    try:
        model.zero_grad()
        args = dict(src_key_padding_mask=_t(g["src_kpm"], dev), tgt_key_padding_mask=_t(g["tgt_kpm"], dev),
                    src_attn_mask=_t(g["src_mask"], dev), tgt_attn_mask=_t(g["tgt_mask"], dev))
        out = model(_t(g["tokens"], dev, torch.int64), _t(g["shifted"], dev), **args)
        assert np.abs(out.detach().cpu().numpy() - g["out_grad"]).max() < 5e-6
        (out * _t(g["dout"], dev)).sum().backward()
        gv = model.named_grad_views()
        for k, ref in grads.items():
            err = np.abs(gv[k].cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-6)
            assert err < 1e-3, (k, err)
        model.zero_grad()
    finally:
        ops.GEMM_PRECISION = keep


def test_layer_checkpointing_gives_identical_gradients(small, dev):
    """model.checkpoint_layers = True keeps only each decoder layer's input and recomputes the layer in the backward:
    the same kernels on the same inputs in the same order, hence bit-identical outputs and gradients."""
    model, g, _ = small
    args = dict(src_key_padding_mask=_t(g["src_kpm"], dev), tgt_key_padding_mask=_t(g["tgt_kpm"], dev),
                src_attn_mask=_t(g["src_mask"], dev), tgt_attn_mask=_t(g["tgt_mask"], dev))
    res = []
    for ck in (False, True):
        model.checkpoint_layers = ck
        model.zero_grad()
        torch.cuda.reset_peak_memory_stats()
        out = model(_t(g["tokens"], dev, torch.int64), _t(g["shifted"], dev), **args)
        (out * _t(g["dout"], dev)).sum().backward()
        res.append((out.detach().clone(), {k: v.clone() for k, v in model.named_grad_views().items()},
                    torch.cuda.max_memory_allocated()))
    model.checkpoint_layers = False
    model.zero_grad()
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
    # (the memory effect needs more than this fixture's single small layer: 63.7 -> 17.4 GiB at configs[3]'s size,
    #  profiles/r02_transformer_train.log)


def test_grouped_linear_and_attention_ops_vs_torch(dev):
    from artspeech_amd.phoneme_to_articulation.transformer.ops import Attention, FoldLN, GroupedLinear, LayerNormAffine, Normalize
    torch.manual_seed(0)
    C_, R, K, G, N = 3, 40, 16, 5, 24
    x = torch.randn(C_, R, K, device=dev, requires_grad=True)
    W = torch.randn(G, N, K, device=dev, requires_grad=True)
    b = torch.randn(G, N, device=dev, requires_grad=True)
    src = (2, 0, 0, 1, 2)
    out = GroupedLinear.apply(x, W, b, src, True)
    ref = torch.relu(torch.einsum("grk,gnk->grn", x[list(src)], W) + b[:, None])
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-4)
    go = torch.randn_like(out)
    gx, gW, gb = torch.autograd.grad(out, (x, W, b), go)
    rx, rW, rb = torch.autograd.grad(ref, (x, W, b), go)
    for a_, r_ in ((gx, rx), (gW, rW), (gb, rb)):
        assert torch.allclose(a_, r_, rtol=1e-4, atol=2e-4), (a_ - r_).abs().max()
    # attention: G blocks, B sequences, T != Tk, additive float masks
    G, B, T, Tk, d, h = 2, 3, 6, 9, 16, 4
    Q = torch.randn(G, B * T, d, device=dev, requires_grad=True)
    Kt = torch.randn(G, B * Tk, d, device=dev, requires_grad=True)
    V = torch.randn(G, B * Tk, d, device=dev, requires_grad=True)
    am = torch.zeros(B, T, Tk, device=dev).masked_fill(torch.rand(B, T, Tk, device=dev) < 0.2, float("-inf"))
    am[:, :, 0] = 0  # keep one visible key per row
    kpm = torch.zeros(B, Tk, device=dev)
    kpm[1, 6:] = float("-inf")
    out = Attention.apply(Q, Kt, V, am, kpm, B, h)

    def ref_attn(Q, Kt, V):
        q = Q.view(G, B, T, h, d // h).permute(0, 1, 3, 2, 4)
        k = Kt.view(G, B, Tk, h, d // h).permute(0, 1, 3, 2, 4)
        v = V.view(G, B, Tk, h, d // h).permute(0, 1, 3, 2, 4)
        s = q @ k.transpose(-1, -2) / (d // h) ** 0.5 + am[None, :, None] + kpm[None, :, None, None]
        return (torch.softmax(s, -1) @ v).permute(0, 1, 3, 2, 4).reshape(G, B * T, d)
    ref = ref_attn(Q, Kt, V)
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5)
    go = torch.randn_like(out)
    for a_, r_ in zip(torch.autograd.grad(out, (Q, Kt, V), go), torch.autograd.grad(ref, (Q, Kt, V), go)):
        assert torch.allclose(a_, r_, rtol=1e-4, atol=1e-5), (a_ - r_).abs().max()
    # LayerNorm pieces
    x = torch.randn(7, 50, device=dev, requires_grad=True)
    r = torch.randn(7, 50, device=dev, requires_grad=True)
    gm = torch.rand(50, device=dev, requires_grad=True)
    bt = torch.randn(50, device=dev, requires_grad=True)
    y = LayerNormAffine.apply(x, r, gm, bt)
    yr = torch.nn.functional.layer_norm(x + r, (50,), gm, bt)
    assert torch.allclose(y, yr, rtol=1e-5, atol=1e-5)
    go = torch.randn_like(y)
    for a_, r_ in zip(torch.autograd.grad(y, (x, r, gm, bt), go), torch.autograd.grad(yr, (x, r, gm, bt), go)):
        assert torch.allclose(a_, r_, rtol=1e-4, atol=1e-5)
    xh = Normalize.apply(x)
    assert torch.allclose(xh, torch.nn.functional.layer_norm(x, (50,)), rtol=1e-5, atol=1e-5)
    Wl = torch.randn(2, 5, 50, device=dev, requires_grad=True)
    g2 = torch.rand(2, 50, device=dev, requires_grad=True)
    b2 = torch.randn(2, 50, device=dev, requires_grad=True)
    bb = torch.randn(2, 5, device=dev, requires_grad=True)
    Wf, bf = FoldLN.apply(Wl, g2, b2, bb)
    Wr, br = Wl * g2[:, None], bb + torch.einsum("grk,gk->gr", Wl, b2)
    assert torch.allclose(Wf, Wr) and torch.allclose(bf, br, rtol=1e-5, atol=1e-5)
    gW, gbv = torch.randn_like(Wf), torch.randn_like(bf)
    for a_, r_ in zip(torch.autograd.grad((Wf, bf), (Wl, g2, b2, bb), (gW, gbv)), torch.autograd.grad((Wr, br), (Wl, g2, b2, bb), (gW, gbv))):
        assert torch.allclose(a_, r_, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("G,B,T,Tk,d,h", [(3, 4, 200, 200, 256, 4), (2, 3, 7, 13, 32, 2), (2, 2, 50, 64, 64, 2), (1, 2, 33, 200, 256, 4),
                                          (2, 2, 1, 1, 64, 1), (1, 1, 256, 256, 128, 2), (2, 3, 45, 97, 64, 4)])
def test_fused_attention_matches_unfused_and_torch(dev, G, B, T, Tk, d, h):
    """as_attention_fwd (inference: no score tensor) against the unfused GEMM + softmax + GEMM path and a float64 formula:
    causal-style -inf masks, -inf key padding, every head width / key-block count the kernel is built for."""
    from artspeech_amd.phoneme_to_articulation.transformer import ops
    torch.manual_seed(T * 7 + Tk)
    Q = torch.randn(G, B * T, d, device=dev)
    Kt = torch.randn(G, B * Tk, d, device=dev)
    V = torch.randn(G, B * Tk, d, device=dev)
    am = torch.zeros(B, T, Tk, device=dev).masked_fill(torch.rand(B, T, Tk, device=dev) < 0.3, float("-inf"))
    am[:, :, 0] = 0   # keep one visible key per row
    am += 0.25 * torch.randn(B, T, Tk, device=dev).clamp(-1, 1) * torch.isfinite(am)   # general additive values too
    kpm = torch.zeros(B, Tk, device=dev)
    if Tk > 4:
        kpm[B - 1, Tk - Tk // 3:] = float("-inf")
    dh = d // h

    def f64(mask, pad):
        q = Q.double().view(G, B, T, h, dh).permute(0, 1, 3, 2, 4)
        k = Kt.double().view(G, B, Tk, h, dh).permute(0, 1, 3, 2, 4)
        v = V.double().view(G, B, Tk, h, dh).permute(0, 1, 3, 2, 4)
        s = q @ k.transpose(-1, -2) / dh ** 0.5
        if mask is not None:
            s = s + mask.double()[None, :, None]
        if pad is not None:
            s = s + pad.double()[None, :, None, None]
        return (torch.softmax(s, -1) @ v).permute(0, 1, 3, 2, 4).reshape(G, B * T, d)
    assert ops.FUSED_ATTENTION and _lib_supported(T, Tk, d, h)
    for mask, pad in ((am, kpm), (None, kpm), (am, None), (None, None)):
        with torch.no_grad():
            fused = ops.Attention.apply(Q, Kt, V, mask, pad, B, h)
            ops.FUSED_ATTENTION = False
            try:
                unfused = ops.Attention.apply(Q, Kt, V, mask, pad, B, h)
            finally:
                ops.FUSED_ATTENTION = True
        ref = f64(mask, pad)
        scale = ref.abs().max().item()
        assert (fused.double() - ref).abs().max().item() <= 1e-5 * scale, "fused vs float64"
        assert (fused - unfused).abs().max().item() <= 1e-5 * scale, "fused vs unfused"
    # a fully masked query row is NaN in both paths (PyTorch semantics), and only that row
    if T > 1:
        am2 = am.clone()
        am2[0, 1, :] = float("-inf")
        with torch.no_grad():
            fused = ops.Attention.apply(Q, Kt, V, am2, None, B, h)
        bad = torch.isnan(fused.view(G, B, T, d))
        assert bad[:, 0, 1].all() and bad.sum().item() == G * d


@pytest.mark.parametrize("G,B,T,Tk,d,h", [(2, 3, 7, 13, 32, 2), (2, 2, 50, 64, 64, 2), (1, 2, 33, 200, 256, 4), (2, 2, 200, 200, 256, 4),
                                          (1, 1, 2, 3, 16, 1), (1, 2, 40, 256, 64, 1), (1, 2, 36, 5, 32, 2)])
def test_fused_attention_training_gradients(dev, G, B, T, Tk, d, h):
    """Training takes the fused forward too (it also leaves the key-major probabilities) and a backward built on them:
    gradients against float64 autograd of the textbook formula, and against the unfused path."""
    from artspeech_amd.phoneme_to_articulation.transformer import ops
    torch.manual_seed(T + Tk)
    dh = d // h
    Q = torch.randn(G, B * T, d, device=dev, requires_grad=True)
    Kt = torch.randn(G, B * Tk, d, device=dev, requires_grad=True)
    V = torch.randn(G, B * Tk, d, device=dev, requires_grad=True)
    am = torch.zeros(B, T, Tk, device=dev).masked_fill(torch.rand(B, T, Tk, device=dev) < 0.3, float("-inf"))
    am[:, :, 0] = 0
    kpm = torch.zeros(B, Tk, device=dev)
    if Tk > 4:
        kpm[B - 1, Tk - Tk // 3:] = float("-inf")
    go = torch.randn(G, B * T, d, device=dev)

    def ref(Q, Kt, V):
        q = Q.view(G, B, T, h, dh).permute(0, 1, 3, 2, 4)
        k = Kt.view(G, B, Tk, h, dh).permute(0, 1, 3, 2, 4)
        v = V.view(G, B, Tk, h, dh).permute(0, 1, 3, 2, 4)
        s = q @ k.transpose(-1, -2) / dh ** 0.5 + am.to(Q.dtype)[None, :, None] + kpm.to(Q.dtype)[None, :, None, None]
        return (torch.softmax(s, -1) @ v).permute(0, 1, 3, 2, 4).reshape(G, B * T, d)
    Qd, Kd, Vd = (t.detach().double().requires_grad_() for t in (Q, Kt, V))
    gref = torch.autograd.grad(ref(Qd, Kd, Vd), (Qd, Kd, Vd), go.double())
    out = ops.Attention.apply(Q, Kt, V, am, kpm, B, h)
    assert out.grad_fn is not None and len(out.grad_fn.saved_tensors) == 5     # the fused path (Q, K, V, P^T, out)
    gf = torch.autograd.grad(out, (Q, Kt, V), go)
    ops.FUSED_ATTENTION = False
    try:
        gu = torch.autograd.grad(ops.Attention.apply(Q, Kt, V, am, kpm, B, h), (Q, Kt, V), go)
    finally:
        ops.FUSED_ATTENTION = True
    for name, a_, u_, r_ in zip("QKV", gf, gu, gref):
        scale = r_.abs().max().item()
        assert (a_.double() - r_).abs().max().item() <= 2e-5 * scale, f"d{name}: fused vs float64"
        assert (a_ - u_).abs().max().item() <= 2e-5 * scale, f"d{name}: fused vs unfused"


def _lib_supported(T, Tk, d, h):
    from artspeech_amd import _lib
    return bool(_lib.lib().as_attention_supported(T, Tk, d, h))


def test_transformer_training_step_decreases_loss(dev):
    from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn
    from artspeech_amd.phoneme_to_articulation.metrics import masked_euclidean_loss
    torch.manual_seed(0)
    V, A, d, h, L, nf = 12, 3, 32, 4, 2, 100
    model = ArtSpeechTransformer(V, A, embed_dim=d, num_heads=h, num_layers=L, num_feat=nf).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    lens = [14, 10, 6]
    batch = [(f"s{i}", torch.randint(1, V, (l,)), torch.rand(l, A, 2, nf // 2), ["p"] * l, torch.rand(l, 1, 2, nf // 2),
              torch.tensor([], dtype=torch.int), list(range(l)), torch.zeros(l)) for i, l in enumerate(lens)]
    c = pad_sequence_transformer_collate_fn(batch)
    tokens, targets, lengths = c[1].to(dev), c[2].to(dev), c[3]
    B, T = tokens.shape
    shifted = torch.cat([torch.zeros(B, 1, A, nf, device=dev), targets[:, 1:].reshape(B, T - 1, A, nf)], dim=1)
    kw = dict(src_key_padding_mask=c[8].to(dev), tgt_key_padding_mask=c[9].to(dev), src_attn_mask=c[10].to(dev), tgt_attn_mask=c[11].to(dev))
    losses = []
    for _ in range(12):
        opt.zero_grad()
        out = model(tokens, shifted, **kw)
        loss = masked_euclidean_loss(out, targets, lengths)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.02, losses


def test_run_transformer_test_harness(dev, tmp_path):
    from torch.utils.data import DataLoader
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import SyntheticArtSpeechDataset, pad_sequence_transformer_collate_fn
    from artspeech_amd.phoneme_to_articulation.metrics import EuclideanDistance
    from artspeech_amd.phoneme_to_articulation.transformer.evaluation import run_transformer_test
    from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
    arts = ["lower-lip", "pharynx", "soft-palate-midline", "tongue", "upper-lip"]
    voc = {"<blank>": 0, "<unk>": 1, **{f"p{i}": i + 2 for i in range(8)}}
    ds = SyntheticArtSpeechDataset(6, voc, arts, n_samples=50, min_len=3, max_len=8, seed=1)
    loader = DataLoader(ds, batch_size=3, shuffle=False, collate_fn=pad_sequence_transformer_collate_fn)
    torch.manual_seed(0)
    model = ArtSpeechTransformer(len(voc), len(arts), embed_dim=32, num_heads=4, num_layers=1, num_feat=100).to(dev)
    res = run_transformer_test(0, model, loader, EuclideanDistance("none"), str(tmp_path), sorted(arts), device=dev)
    assert set(res) == {"loss", *arts} and np.isfinite(res["loss"])
    assert set(res["tongue"]) == {"x_corr", "y_corr", "p2cp", "p2cp_mm", "med", "med_mm"}
    import os
    assert sum(f == "tract_variables.csv" for _, _, fs in os.walk(tmp_path) for f in fs) == 6


def test_full_width_model_matches_reference_fixture(dev):
    """BASELINE configs[3]'s MODEL at full width (d=256, 6 layers, A=11, 419.6 M parameters) against the reference itself
    on a two-utterance ragged batch of T=200 (tests/golden/make_golden.py gen_transformer_c4): the seeded recipe
    (c4_case, one text for both sides) is re-run with this package's classes, checksums in the fixture prove that
    weights and inputs are the reference run's, then loss, contour slices and norm + strided slice of every one of the
    ~10 000 parameter gradients are compared."""
    import importlib.util
    import os
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn
    from artspeech_amd.phoneme_to_articulation.metrics import masked_euclidean_loss
    from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)     # definitions only: nothing of /root/reference is touched until its main() runs
    g = load_golden("transformer_c4_full")
    model, tokens, targets, lengths, shifted, kw, cfg = mg.c4_case(ArtSpeechTransformer, pad_sequence_transformer_collate_fn)
    assert tuple(int(v) for v in g["cfg"]) == cfg and np.array_equal(lengths.numpy(), g["lengths"])
    assert int(tokens.sum()) == int(g["tok_sum"])
    assert abs(targets.double().sum().item() - float(g["tgt_sum"])) < 1e-6
    w_abs = sum(p.detach().double().abs().sum().item() for k, p in model.state_dict().items() if k != "positional_encoding.pe"
                and not k.endswith(".pe"))
    assert abs(w_abs - float(g["w_abs_sum"])) < 1e-7 * float(g["w_abs_sum"]), (w_abs, float(g["w_abs_sum"]))
    model = model.to(dev).eval()
    out = model(tokens.to(dev), shifted.to(dev), **{k: v.to(dev) for k, v in kw.items()})
    loss = masked_euclidean_loss(out, targets.to(dev), lengths)
    assert abs(loss.item() - float(g["loss"])) < 2e-6, (loss.item(), float(g["loss"]))
    worst_valid = worst_pad = 0.0
    for (b, t), want in zip(g["positions"], g["out_slices"]):
        got = out[int(b), int(t)].detach().cpu().numpy()
        ratio = float((np.abs(got - want) / (1e-4 * np.abs(want) + 2e-6)).max())
        if int(t) < int(lengths[int(b)]):
            worst_valid = max(worst_valid, ratio)
        else:
            worst_pad = max(worst_pad, ratio)
    print(f"full-width transformer contours: worst |got - ref| / (1e-4 |ref| + 2e-6) = {worst_valid:.2f} on valid frames, "
          f"{worst_pad:.2f} on padded frames")
    assert worst_valid <= 1.0, worst_valid      # north_star: 1e-4 relative on contour coordinates
    # frames past the utterance's end enter neither the loss nor any metric (train_..._transformer.py:113-118); there the
    # reference differs from ITSELF (1 thread vs 8) by up to 0.52 of the bound (fixture: self_contour_ratio), this path from
    # the reference by 1.11 (measured, round 3): held to 1.5 x the bound instead of the 3 x of round 2
    assert worst_pad <= 1.5, worst_pad
    loss.backward()
    gv = model.named_grad_views()
    names = str(g["names"]).split("\n")
    assert set(names) == set(gv), set(names) ^ set(gv)
    n = g["gslice"].shape[1]
    norm_err, slice_err = np.zeros(len(names)), np.zeros(len(names))
    for i, k in enumerate(names):
        v = gv[k]
        gn, gm = float(g["gnorm"][i]), float(g["gmax"][i])
        norm_err[i] = abs(float(v.double().norm()) - gn) / max(gn, 1e-30)
        sl = v.reshape(-1)[:: max(1, v.numel() // n)][:n].cpu().numpy()
        slice_err[i] = float(np.abs(sl - g["gslice"][i, :sl.size]).max()) / max(gm, 1e-30)
    # Yardstick: the reference against ITSELF with one thread instead of eight (quantiles 50 / 90 / 99 / 100 % over the
    # tensors, stored by make_golden.py).  ~1e8 ReLU decisions, some within an ulp of zero, sit between the loss and the
    # early layers; each that falls the other way moves gradient elements by a whole frame's term, so two correct fp32
    # evaluations differ by ~5e-3 of max|g| on the median tensor.  This path must be as close to the reference as the
    # reference is to itself: every quantile within 1.5x (the maximum, a single tensor, within 2x).
    q = (0.5, 0.9, 0.99, 1.0)
    got_n, got_s = np.quantile(norm_err, q), np.quantile(slice_err, q)
    # by depth: the heads + output trunk have few ReLU decisions between them and the loss, the last decoder layer a few more
    def group_of(k):   # names are the reference's state_dict keys
        if k.startswith(("predictors.", "linear.")):
            return "heads+trunk"
        if k.startswith("decoder.layers."):
            return "decoder." + k.split(".")[2]
        return k.split(".")[0]
    groups = {}
    for i, k in enumerate(names):
        groups.setdefault(group_of(k), []).append(i)
    group_stats = {gname: {"tensors": len(idx), "norm_err_max": float(norm_err[idx].max()), "slice_err_max": float(slice_err[idx].max()),
                           "slice_err_median": float(np.median(slice_err[idx]))} for gname, idx in sorted(groups.items())}
    for gname, st in group_stats.items():
        print(f"  {gname:14s} {st}")
    print("full-width transformer gradients, quantiles 50/90/99/100 % over", len(names), "tensors")
    print("  norm error          : this path vs reference", got_n, "  reference vs itself", g["self_norm_q"])
    print("  slice error / max|g|: this path vs reference", got_s, "  reference vs itself", g["self_slice_q"])
    try:
        import json
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/c4_full_grad_errors.json", "w") as f:
            json.dump({"quantiles": q, "norm_err": got_n.tolist(), "slice_err": got_s.tolist(),
                       "reference_self_norm_err": g["self_norm_q"].tolist(), "reference_self_slice_err": g["self_slice_q"].tolist(),
                       "contours_valid": worst_valid, "contours_padded": worst_pad,
                       "worst_tensor": names[int(slice_err.argmax())], "by_depth": group_stats,
                       "sample_names": names[:5] + names[-5:]}, f, indent=1)
    except OSError:
        pass
    for i in range(4):
        f = 2.0 if i == 3 else 1.5
        assert got_n[i] <= f * float(g["self_norm_q"][i]), (q[i], got_n[i], float(g["self_norm_q"][i]))
        assert got_s[i] <= f * float(g["self_slice_q"][i]), (q[i], got_s[i], float(g["self_slice_q"][i]))
    # ... and BY DEPTH, against the reference's own per-tensor discrepancy of the same group (fixture: self_norm_err,
    # self_slice_err): the heads and the output trunk have only their own three ReLUs between them and the loss, so their
    # yardstick is ~500 x tighter than the decoder's (median slice error 1e-5 of max|g| instead of 5e-3); a wrong term there
    # no longer hides behind the deep layers' noise.  Quantiles 50 / 90 % within 1.5 x, the group's worst tensor within 2.5 x
    # (+ 5e-5: fp32 summation order on values that small).
    self_n, self_s = g["self_norm_err"].astype(np.float64), g["self_slice_err"].astype(np.float64)
    for gname, idx in sorted(groups.items()):
        if len(idx) < 4:
            continue
        idx = np.array(idx)
        for what, ours, ref in (("norm", norm_err[idx], self_n[idx]), ("slice", slice_err[idx], self_s[idx])):
            for qq, f in ((0.5, 1.5), (0.9, 1.5), (1.0, 2.5)):
                a_, b_ = float(np.quantile(ours, qq)), float(np.quantile(ref, qq))
                assert a_ <= f * b_ + 5e-5, (gname, what, qq, a_, b_)


class _CapturedLoader:
    def __init__(self, batches, dataset_config):
        import types
        self.batches = batches
        self.dataset = types.SimpleNamespace(dataset_config=dataset_config)

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def _perturb_by_key(i, k, v):
    """tests/golden/make_golden.py::perturb_by_key (the fixture model = seeded init + this, by sorted state_dict key)."""
    if k == "pos_encoding.pe":
        return v
    n = v.numel()
    out = v + (0.02 * torch.cos(0.37 * torch.arange(n, dtype=torch.float64) + i)).to(v.dtype).view(v.shape)
    return out * 0.1 if k == "tgt_embedding.1.weight" else out


def test_transformer_loops_match_reference_fixture(dev, tmp_path, monkeypatch, capsys):
    """tests/golden/transformer_loops.npz holds what the REFERENCE's transformer harnesses produced on a captured 6-utterance
    loader: run_epoch(TRAIN) with SGD and run_epoch(VALID) with p2cp_mean (train_phoneme_to_articulation_transformer.py:49-149),
    and run_transformer_test (transformer/evaluation.py:19-191) with ONE utterance whose source is fully masked, so that the
    prediction is NaN and the filter of :69-86 runs.  The drop-in loops must reproduce the info dicts, the parameter update
    of the epoch, the skipped sentence, the directories written (the reference reports kept utterance j under the j-th id of
    the unfiltered batch) and the tract-variable tables."""
    import csv
    import train_phoneme_to_articulation_transformer as tr
    from conftest import load_golden
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.metrics import P2CPDistance
    from artspeech_amd.phoneme_to_articulation.metrics import EuclideanDistance
    from artspeech_amd.phoneme_to_articulation.transformer import models as tmodels
    from artspeech_amd.phoneme_to_articulation.transformer.evaluation import run_transformer_test
    from artspeech_amd.settings import DATASET_CONFIG, TRAIN, VALID
    g = load_golden("transformer_loops")
    V, A, d, heads, L, nf = (int(v) for v in g["cfg"])
    arts = [str(a) for a in g["articulators"]]
    # the fixture model has every dropout probability at 0 (the encoder's library-default 0.1 is a random mask)
    monkeypatch.setattr(tmodels, "ENC_DROPOUT", 0.0)
    torch.manual_seed(33)
    model = tmodels.ArtSpeechTransformer(V, A, embed_dim=d, num_heads=heads, num_layers=L, num_feat=nf)
    sd = model.state_dict()
    assert abs(float(sum(v.double().abs().sum() for v in sd.values())) - float(g["init_abs_sum"])) < 1e-6 * float(g["init_abs_sum"])
    model.load_state_dict({k: _perturb_by_key(i, k, v) for i, (k, v) in enumerate(sorted(sd.items()))})
    w0 = {k: v.detach().clone().double() for k, v in model.state_dict().items()}
    model = model.to(dev)
    items = []
    for i in range(len(g["lens"])):
        items.append((str(g[f"in{i}_id"]), torch.from_numpy(g[f"in{i}_tokens"]), torch.from_numpy(g[f"in{i}_targets"]),
                      [str(p) for p in g[f"in{i}_phonemes"]], torch.from_numpy(g[f"in{i}_refs"]), torch.tensor([], dtype=torch.int),
                      [str(f) for f in g[f"in{i}_frames"]], torch.from_numpy(g[f"in{i}_voicing"])))
    cfg = DATASET_CONFIG["artspeech2"]
    batches = [pad_sequence_transformer_collate_fn(items[:3]), pad_sequence_transformer_collate_fn(items[3:])]
    loader = _CapturedLoader(batches, cfg)
    crit = EuclideanDistance("none")
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    info = tr.run_epoch(TRAIN, 1, model, loader, opt, crit, device=dev)
    assert set(info) == {"loss"}
    assert abs(info["loss"] - float(g["train_loss"])) < 2e-6, (info["loss"], float(g["train_loss"]))
    # what the two SGD steps changed: per tensor, norm and a strided slice of w1 - w0 (fp32 parameters: an ulp of |w| rides on
    # every difference besides the update itself)
    sd1 = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
    keys = [str(k) for k in g["upd_keys"]]
    assert keys == sorted(sd1)
    for i, k in enumerate(keys):
        dlt = (sd1[k] - w0[k]).flatten()
        wmax = max(1.0, float(w0[k].abs().max()))
        assert abs(float(dlt.norm()) - float(g["upd_norm"][i])) <= 1e-3 * float(g["upd_norm"][i]) + 2.5e-7 * wmax * dlt.numel() ** 0.5, k
        sl = dlt[::max(1, dlt.numel() // 33)][:33].numpy()
        assert_grad_close(sl, g["upd_slices"][i, :len(sl)].astype(np.float64), f"transformer_loops: SGD delta of {k}",
                          rtol=1e-3, atol_frac=1e-4, atol_abs=2.5e-7 * wmax + 1e-4 * float(g["upd_max"][i]))
    vinfo = tr.run_epoch(VALID, 1, model, loader, opt, crit, fn_metrics={"p2cp_mean": P2CPDistance(cfg)}, device=dev)
    assert set(vinfo) == {"loss", "p2cp_mean"}
    assert abs(vinfo["loss"] - float(g["valid_loss"])) < 2e-6
    assert abs(vinfo["p2cp_mean"] - float(g["valid_p2cp_mean"])) / float(g["valid_p2cp_mean"]) < 2e-3
    # ---- test loop with the NaN utterance
    tb = [list(b) for b in batches]
    nb, nr = int(g["nan_batch"]), int(g["nan_row"])
    tb[nb][8] = tb[nb][8].clone()
    tb[nb][8][nr] = float("-inf")
    test_loader = _CapturedLoader([tuple(b) for b in tb], cfg)
    capsys.readouterr()
    res = run_transformer_test(7, model, test_loader, crit, str(tmp_path), arts, device=dev, regularize_out=False)
    printed = capsys.readouterr().out
    assert "Invalid outputs produced for sentences:" in printed
    skipped = [ln.strip() for ln in printed.split("Invalid outputs produced for sentences:")[1].strip().splitlines() if ln.strip()]
    assert skipped == [str(s) for s in g["skipped"]] == ["sent2"]
    assert list(res) == ["loss"] + arts
    assert abs(res["loss"] - float(g["test_loss"])) < 2e-6, (res["loss"], float(g["test_loss"]))
    names = [str(n) for n in g["test_metric_names"]]
    # free-running generate() feeds its own 1e-6 differences back for up to 17 frames; correlations of 4..17 samples
    tol = {"x_corr": ("abs", 2e-4), "y_corr": ("abs", 2e-4), "p2cp": ("rel", 2e-3), "p2cp_mm": ("rel", 2e-3), "med": ("rel", 2e-5),
           "med_mm": ("rel", 2e-5)}
    for i, a in enumerate(arts):
        assert list(res[a]) == names
        for j, n in enumerate(names):
            want, got = float(g["test_metrics"][i, j]), res[a][n]
            err = abs(got - want) / (abs(want) if tol[n][0] == "rel" else 1.0)
            assert err < tol[n][1], (a, n, got, want)
    # directories: the kept utterances of the NaN batch are reported under the first ids of the unfiltered batch
    dirs = sorted(os.listdir(os.path.join(str(tmp_path), "7")))
    assert dirs == [str(s) for s in g["sentence_dirs"]]
    first = str(g["first_dir"])
    sdir = os.path.join(str(tmp_path), "7", first)
    assert sorted(os.listdir(os.path.join(sdir, "contours"))) == [str(f) for f in g["contour_files"]]
    with open(os.path.join(sdir, "phonemes.csv")) as f:
        assert [list(r) for r in csv.reader(f)] == [[str(c) for c in r] for r in g["phonemes_csv"]]
    assert np.abs(np.load(os.path.join(sdir, "contours", f"{str(g['first_frame'])}_tongue.npy")) - g["pred_tongue_first"]).max() < 2e-5
    num = [str(c) for c in g["tv_numeric_columns"]]
    for sd_ in dirs:
        with open(os.path.join(str(tmp_path), "7", sd_, "tract_variables.csv")) as f:
            rows = list(csv.reader(f))
        cols = rows[0]
        assert cols == [str(c) for c in g["tv_columns"]]
        assert [r[cols.index("frame")] for r in rows[1:]] == [str(v) for v in g[f"tv_{sd_}_frames"]]
        assert [r[cols.index("phoneme")] for r in rows[1:]] == [str(v) for v in g[f"tv_{sd_}_phonemes"]]
        got = np.array([[float(r[cols.index(c)]) for c in num] for r in rows[1:]])
        want = g[f"tv_{sd_}_values"]
        tcols = [j for j, c in enumerate(num) if "_target_poc_" in c]
        assert np.array_equal(got[:, tcols], want[:, tcols]), sd_      # targets are inputs: the same closest pairs, bit for bit
        dcols = [j for j, c in enumerate(num) if c.endswith("_target")]
        assert (np.abs(got[:, dcols] - want[:, dcols]) <= 2e-7 / np.maximum(want[:, dcols], 1e-4) + 1e-7).all(), sd_
        vcols = [j for j, c in enumerate(num) if c.endswith("_pred")]
        assert (np.abs(got[:, vcols] - want[:, vcols]) <= 2e-7 / np.maximum(want[:, vcols], 1e-4) + 3e-5).all(), sd_
        pcols = [j for j, c in enumerate(num) if "_pred_poc_" in c]
        assert (np.abs(got[:, pcols] - want[:, pcols]) < 3e-5).mean() > 0.95, sd_
