#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs ONLY in the build container (it reads /root/reference, which does not exist on the GPU
box).  It imports the reference's modules by file path, drives them on seeded synthetic inputs and
stores inputs + expected outputs as small ``.npz`` files.  Nothing of the reference's source text is
stored: the fixtures are data only.

    python tests/golden/make_golden.py            # regenerates every fixture

Absent third-party packages (funcy, vt_tools, numba, shapely) are replaced by *name-only* shims
below so that the module-level imports succeed:
  * ``vt_tools`` exports articulator NAME constants (used as dict keys only; the strings are the
    ones the reference's YAML configs use, e.g. thesis_config/.../train_model_free.yaml:13-22);
  * ``vt_tools.metrics.euclidean(u, v)`` / ``distance_matrix`` are ASSUMED to be the L2 norm of the
    difference / pairwise L2 (vt_tools is an un-vendored, un-pinned dependency: requirements.txt:28-38).
    => area_function fixtures are "parity pinned up to that assumption" (see DESIGN.md);
  * ``numba.jit`` -> identity decorator, ``np.float`` -> float (removed from numpy>=1.24,
    area_function.py:130 still uses it).
shapely is absent: evenly_spaced_fx / intersect_semipolar_grid cannot run => no fixture (unpinned).
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- import plumbing
def _shim(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def install_shims():
    _shim("funcy", lmap=lambda f, *s: list(map(f, *s)), lfilter=lambda f, s: list(filter(f, s)),
          flatten=lambda seq: [x for sub in seq for x in sub])
    vt = _shim(
        "vt_tools",
        LOWER_LIP="lower-lip", PHARYNX="pharynx", SOFT_PALATE_MIDLINE="soft-palate-midline",
        TONGUE="tongue", UPPER_LIP="upper-lip", UPPER_INCISOR="upper-incisor", SOFT_PALATE="soft-palate",
    )
    vt.metrics = _shim(
        "vt_tools.metrics",
        euclidean=lambda u, v: float(np.sqrt(np.sum((np.asarray(u, dtype=np.float64) - np.asarray(v, dtype=np.float64)) ** 2))),
        distance_matrix=lambda a, b: np.sqrt(((np.asarray(a)[:, None, :] - np.asarray(b)[None, :, :]) ** 2).sum(-1)),
        p2cp_mean=None,
    )
    _shim("numba", jit=lambda *a, **k: (lambda f: f))
    geom = _shim("shapely.geometry", LineString=None, Point=None)
    _shim("shapely", geometry=geom)
    if not hasattr(np, "float"):
        np.float = float


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def sd_to_np(prefix, sd):
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


# ----------------------------------------------------------------------------- model fixtures
def masked_mean_loss(loss_fn, make_padding_mask, outputs, targets, lengths):
    """train_phoneme_to_articulation.py:86-90, executed through the reference's own objects."""
    loss = loss_fn(outputs, targets)
    padding_mask = make_padding_mask(lengths)
    bs, max_len, num_articulators, features = loss.shape
    loss = loss.view(bs * max_len, num_articulators, features)
    return loss[padding_mask.view(bs * max_len)].mean()


def gen_artspeech(name, models, p2a_metrics, ed_metrics, helpers, settings, *, vocab, n_art, embed_dim,
                  hidden, n_samples, B, T, lengths, seed, database="artspeech2"):
    torch.manual_seed(seed)
    model = models.ArtSpeech(vocab, n_art, embed_dim=embed_dim, hidden_size=hidden, n_samples=n_samples)
    x = torch.randint(0, vocab, (B, T))
    lengths = torch.tensor(lengths, dtype=torch.int)
    tgt = torch.rand(B, T, n_art, 2, n_samples)
    for i, l in enumerate(lengths):
        x[i, l:] = 0
        tgt[i, l:] = 0
    out = model(x, lengths)
    loss = masked_mean_loss(p2a_metrics.EuclideanDistance("none"), helpers.make_padding_mask, out, tgt, lengths)
    loss.backward()
    p2cp = ed_metrics.P2CPDistance(settings.DATASET_CONFIG[database])(out.detach(), tgt, lengths)
    arrays = dict(
        x=x.numpy(), lengths=lengths.numpy(), targets=tgt.numpy(), out=out.detach().numpy(),
        loss=np.float32(loss.item()), p2cp_mm=np.float32(p2cp.item()),
        cfg=np.array([vocab, n_art, embed_dim, hidden, n_samples], dtype=np.int64),
    )
    arrays.update(sd_to_np("w.", model.state_dict()))
    arrays.update({"g." + k: p.grad.numpy() for k, p in model.named_parameters()})
    save(name, **arrays)
    gn = float(torch.sqrt(sum((p.grad ** 2).sum() for p in model.parameters())))
    return dict(out_sum=float(out.sum()), loss=float(loss), grad_norm=gn, p2cp_mm=float(p2cp),
                out_0000_3=[float(v) for v in out[0, 0, 0, 0, :3]])


def gen_simple(name, models, *, vocab, n_art, embed_dim, hidden, n_samples, B, T, seed):
    torch.manual_seed(seed)
    model = models.SimpleArtSpeech(vocab, n_art, embed_dim=embed_dim, hidden_size=hidden, num_samples=n_samples)
    x = torch.randint(0, vocab, (B, T))
    out = model(x, None)
    dout = torch.randn_like(out)
    (out * dout).sum().backward()
    arrays = dict(x=x.numpy(), out=out.detach().numpy(), dout=dout.numpy(),
                  cfg=np.array([vocab, n_art, embed_dim, hidden, n_samples], dtype=np.int64))
    arrays.update(sd_to_np("w.", model.state_dict()))
    arrays.update({"g." + k: p.grad.numpy() for k, p in model.named_parameters()})
    save(name, **arrays)


def gen_predictor(name, models, *, in_features, n_samples, rows, seed):
    """ArticulatorPredictor alone (models.py:7-33): fwd + bwd on a (2, rows/2, in) input."""
    torch.manual_seed(seed)
    head = models.ArticulatorPredictor(in_features, n_samples)
    # non-trivial LayerNorm affine so the fixture exercises gamma/beta
    with torch.no_grad():
        for i in (0, 3, 6):
            head.linear[i].weight.uniform_(0.5, 1.5)
            head.linear[i].bias.uniform_(-0.5, 0.5)
    x = torch.randn(2, rows // 2, in_features, requires_grad=True)
    out = head(x)
    dout = torch.randn_like(out)
    (out * dout).sum().backward()
    arrays = dict(x=x.detach().numpy(), out=out.detach().numpy(), dout=dout.numpy(), dx=x.grad.numpy())
    arrays.update(sd_to_np("w.", head.state_dict()))
    arrays.update({"g." + k: p.grad.numpy() for k, p in head.named_parameters()})
    save(name, **arrays)


# ----------------------------------------------------------------------------- metric fixtures
def gen_metrics(p2a_metrics, ed_metrics, root_metrics, settings):
    torch.manual_seed(7)
    B, T, A, N = 3, 9, 4, 50
    out = torch.rand(B, T, A, 2, N, requires_grad=True)
    tgt = torch.rand(B, T, A, 2, N)
    lengths = torch.tensor([9, 6, 2], dtype=torch.int)
    euc_none = p2a_metrics.EuclideanDistance("none")(out, tgt)
    euc_mean = p2a_metrics.EuclideanDistance("mean")(out, tgt)
    euc_mean.backward()
    p2cp_none = p2a_metrics.MeanP2CPDistance("none")(out.detach().transpose(-1, -2), tgt.transpose(-1, -2))
    p2cp_mean = p2a_metrics.MeanP2CPDistance("mean")(out.detach().transpose(-1, -2), tgt.transpose(-1, -2))
    # N=10 <= 25: torch.cdist takes the direct (non-matmul) path
    u10 = torch.rand(5, 7, 10, 2)
    v12 = torch.rand(5, 7, 12, 2)
    p2cp_small = p2a_metrics.MeanP2CPDistance("none")(u10, v12)
    p2cp_mm = {db: ed_metrics.P2CPDistance(settings.DATASET_CONFIG[db])(out.detach(), tgt, lengths).item()
               for db in ("artspeech2", "gottingen")}
    x_corr, y_corr = root_metrics.pearsons_correlation(out.detach(), tgt)
    save(
        "metrics",
        out=out.detach().numpy(), tgt=tgt.numpy(), lengths=lengths.numpy(),
        euc_none=euc_none.detach().numpy(), euc_mean=np.float32(euc_mean.item()), euc_mean_grad=out.grad.numpy(),
        p2cp_none=p2cp_none.numpy(), p2cp_mean=np.float32(p2cp_mean.item()),
        u10=u10.numpy(), v12=v12.numpy(), p2cp_small=p2cp_small.numpy(),
        p2cp_mm_artspeech2=np.float32(p2cp_mm["artspeech2"]), p2cp_mm_gottingen=np.float32(p2cp_mm["gottingen"]),
        root_p2cp=root_metrics.p2cp_distance(out.detach(), tgt).numpy(),
        root_euclid=root_metrics.euclidean_distance(out.detach(), tgt).numpy(),
        x_corr=x_corr.numpy(), y_corr=y_corr.numpy(),
    )


def gen_tract_variables(tv):
    """tract_variables.py:73-125 on random frames; inputs per articulator are (N, 2)."""
    torch.manual_seed(11)
    names = ["lower-lip", "pharynx", "soft-palate-midline", "tongue", "upper-incisor", "upper-lip"]  # sorted
    F, N = 40, 50
    frames = torch.rand(F, len(names), 2, N)  # model-output layout (A, 2, N) per frame
    values = np.zeros((F, 4), dtype=np.float32)
    poc1 = np.zeros((F, 4, 2), dtype=np.float32)
    poc2 = np.zeros((F, 4, 2), dtype=np.float32)
    tv_names = ["LA", "TTCD", "TBCD", "VEL"]
    for f in range(F):
        inputs = {art: t.T for art, t in zip(names, frames[f])}  # phoneme_to_articulation/__init__.py:253-255
        res = tv.calculate_vocal_tract_variables(inputs)
        assert [k for k, v in res.items() if v is not None] == tv_names
        for j, k in enumerate(tv_names):
            values[f, j] = res[k]["value"]
            poc1[f, j] = res[k]["poc_1"].numpy()
            poc2[f, j] = res[k]["poc_2"].numpy()
    save("tract_variables", frames=frames.numpy(), values=values, poc1=poc1, poc2=poc2,
         articulators=np.array(names), tv_names=np.array(tv_names))


def gen_area_function(af):
    rng = np.random.RandomState(3)
    cases = {}
    for i, nw in enumerate((100, 37, 2, 1)):
        t = np.linspace(0.0, 1.0, nw)
        internal = np.stack([t + 0.02 * rng.randn(nw), 0.3 + 0.05 * rng.randn(nw)], axis=1)
        external = np.stack([t + 0.02 * rng.randn(nw), 0.6 + 0.05 * rng.randn(nw)], axis=1)
        dists, fx = af.area_function(internal, external)
        cases[f"int{i}"], cases[f"ext{i}"] = internal, external
        cases[f"dists{i}"], cases[f"fx{i}"] = np.asarray(dists, dtype=np.float64), np.asarray(fx, dtype=np.float64)
    # non-default alpha/beta
    d, fx = af.area_function(cases["int0"], cases["ext0"], alpha=1.5, beta=1.3)
    cases["dists0_ab"], cases["fx0_ab"] = np.asarray(d), np.asarray(fx)
    grid = af.build_semipolar_grid(np.array([0.5, 0.45]), np.deg2rad(10.0), np.deg2rad(-5.0), 0.05, np.deg2rad(7.5), grid_res=50)
    cases["grid"] = grid
    cases["grid_args"] = np.array([0.5, 0.45, np.deg2rad(10.0), np.deg2rad(-5.0), 0.05, np.deg2rad(7.5), 50.0])
    save("area_function", **cases)


def gen_host(helpers, dataset):
    """make_padding_mask (helpers.py:79-91) and the two collate fns (dataset.py:27-123)."""
    torch.manual_seed(5)
    lens = [5, 9, 3, 9]
    A, N = 2, 6
    batch = []
    for i, l in enumerate(lens):
        batch.append((
            f"s{i}", torch.randint(1, 20, (l,)), torch.rand(l, A, 2, N), [f"p{i}_{j}" for j in range(l)],
            torch.rand(l, 1, 2, N), torch.tensor([], dtype=torch.int), list(range(100 * i, 100 * i + l)),
            (torch.rand(l) > 0.5).float(),
        ))
    c8 = dataset.pad_sequence_collate_fn(batch)
    c12 = dataset.pad_sequence_transformer_collate_fn(batch)
    arrays = {}
    for i, item in enumerate(batch):
        arrays[f"in{i}_tokens"], arrays[f"in{i}_targets"] = item[1].numpy(), item[2].numpy()
        arrays[f"in{i}_refs"], arrays[f"in{i}_voicing"] = item[4].numpy(), item[7].numpy()
    arrays.update(
        ids=np.array(c8[0]), tokens=c8[1].numpy(), targets=c8[2].numpy(), lengths=c8[3].numpy(),
        phonemes0=np.array(c8[4][0]), refs=c8[5].numpy(), frames0=np.array(c8[6][0]), voicing=c8[7].numpy(),
        src_kpm=c12[8].numpy(), tgt_kpm=c12[9].numpy(), src_mask=c12[10].numpy(), tgt_mask=c12[11].numpy(),
        t_tokens=c12[1].numpy(), t_lengths=c12[3].numpy(),
        mask_9_6_2=helpers.make_padding_mask(torch.tensor([9, 6, 2])).numpy(),
    )
    save("host_collate", **arrays)


def gen_transformer(tmod, dataset):
    """ArtSpeechTransformer (transformer/models.py:280-474): forward as the trainer calls it
    (train_phoneme_to_articulation_transformer.py:99-111) and generate() as the test loop calls it
    (transformer/evaluation.py:63-67).  The reference is pinned to torch 2.0.1, whose nn.TransformerDecoder
    simply loops over its layers; torch 2.10's version probes layers[0].self_attn (absent on the custom
    layer), so the decoder's forward is replaced by that 2.0.1 loop here (oracle-side only)."""
    import types as _t
    torch.manual_seed(21)
    V, A, d, heads, L, nf = 13, 3, 32, 4, 2, 20
    model = tmod.ArtSpeechTransformer(V, A, embed_dim=d, num_heads=heads, num_layers=L, num_feat=nf)
    init_abs_sum = float(sum(p.detach().double().abs().sum() for p in model.parameters()))  # seed-for-seed init check (seed 21)
    # decoder layers start as deep copies (identical weights): perturb them so the fixture tells layers apart
    with torch.no_grad():
        for prm in model.decoder.parameters():
            prm.add_(0.02 * torch.randn_like(prm))
        for m in model.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.uniform_(0.7, 1.3)
                m.bias.uniform_(-0.2, 0.2)

    def loop_forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                     memory_key_padding_mask=None, **_):
        out = tgt
        for mod in self.layers:
            out = mod(out, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                      tgt_key_padding_mask=tgt_key_padding_mask, memory_key_padding_mask=memory_key_padding_mask)
        return out
    model.decoder.forward = _t.MethodType(loop_forward, model.decoder)
    model.eval()

    lens = [7, 5]
    batch = []
    for i, l in enumerate(lens):
        batch.append((f"s{i}", torch.randint(1, V, (l,)), torch.rand(l, A, 2, nf // 2), ["p"] * l, torch.rand(l, 1, 2, nf // 2),
                      torch.tensor([], dtype=torch.int), list(range(l)), torch.zeros(l)))
    c = dataset.pad_sequence_transformer_collate_fn(batch)
    tokens, targets, lengths = c[1], c[2], c[3]
    src_kpm, tgt_kpm, src_mask, tgt_mask = c[8], c[9], c[10], c[11]
    bs, T = tokens.shape
    shifted = torch.cat([torch.zeros(bs, 1, A, nf), targets[:, 1:].reshape(bs, T - 1, A, nf)], dim=1)
    captured = {}
    hook = model.encoder.register_forward_hook(lambda m, i, o: captured.__setitem__("enc", o.detach().clone()))
    out_grad = model(tokens, shifted, src_key_padding_mask=src_kpm, tgt_key_padding_mask=tgt_kpm,
                     src_attn_mask=src_mask, tgt_attn_mask=tgt_mask)          # grad enabled: standard encoder path
    enc_grad = captured["enc"]
    dout = torch.rand_like(out_grad)
    (out_grad * dout).sum().backward()                                         # gradients of every parameter
    with torch.no_grad():                                                      # how evaluation runs it
        out_nograd = model(tokens, shifted, src_key_padding_mask=src_kpm, tgt_key_padding_mask=tgt_kpm,
                           src_attn_mask=src_mask, tgt_attn_mask=tgt_mask)
        enc_nograd = captured["enc"]
        gen = model.generate(tokens, src_key_padding_mask=src_kpm)
        enc_gen = captured["enc"]
    hook.remove()
    arrays = dict(tokens=tokens.numpy(), targets=targets.numpy(), lengths=lengths.numpy(), shifted=shifted.numpy(),
                  src_kpm=src_kpm.numpy(), tgt_kpm=tgt_kpm.numpy(), src_mask=src_mask.numpy(), tgt_mask=tgt_mask.numpy(),
                  out_grad=out_grad.detach().numpy(), enc_grad=enc_grad.numpy(), out_nograd=out_nograd.numpy(),
                  enc_nograd=enc_nograd.numpy(), gen=gen.numpy(), enc_gen=enc_gen.numpy(),
                  dout=dout.numpy(), cfg=np.array([V, A, d, heads, L, nf], dtype=np.int64))
    arrays.update(sd_to_np("w.", model.state_dict()))
    arrays.update({"g." + k: p.grad.numpy() for k, p in model.named_parameters()})
    save("transformer_small", **arrays)
    return dict(out_sum=float(out_nograd.sum()), gen_sum=float(gen.nansum()), n_keys=len(model.state_dict()),
                enc_pad_zero_nograd=bool((enc_nograd[1, 5:] == 0).all()), enc_pad_zero_grad=bool((enc_grad[1, 5:] == 0).all()),
                gen_nan=bool(gen.isnan().any()), params=sum(p.numel() for p in model.parameters()), seed=21,
                init_abs_sum=init_abs_sum)


def gen_deepspeech2(ds2):
    """DeepSpeech2 articulatory scorer (phoneme_recognition/deepspeech2.py:90-195) in eval mode, as test() drives it
    (phoneme_recognition/__init__.py:213-236): logits, features and the topk(k=1) phoneme indices."""
    out = {}
    cases = {
        # adapter + voicing, two residual / recurrent layers (the thesis scorer's structure, scaled down)
        "deepspeech2_small": dict(in_channels=2, num_residual_layers=2, num_rnn_layers=2, rnn_hidden_size=64, num_classes=11,
                                  num_features=23, adapter_out_features=16, B=3, T=9, voicing=True, seed=31),
        # no adapter, no voicing, odd sizes, one layer of each
        "deepspeech2_plain": dict(in_channels=2, num_residual_layers=1, num_rnn_layers=1, rnn_hidden_size=32, num_classes=7,
                                  num_features=12, adapter_out_features=None, B=2, T=5, voicing=False, seed=32),
    }
    for name, c in cases.items():
        torch.manual_seed(c["seed"])
        model = ds2.DeepSpeech2(c["in_channels"], c["num_residual_layers"], c["num_rnn_layers"], c["rnn_hidden_size"],
                                num_classes=c["num_classes"], num_features=c["num_features"], dropout=0.1,
                                adapter_out_features=c["adapter_out_features"])
        init_abs_sum = float(sum(p.detach().double().abs().sum() for p in model.parameters()))  # seed-for-seed init check
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.LayerNorm):
                    m.weight.uniform_(0.7, 1.3)
                    m.bias.uniform_(-0.2, 0.2)
        model.eval()
        x = torch.rand(c["B"], c["in_channels"], c["num_features"], c["T"])
        voicing = (torch.rand(c["B"], c["T"]) > 0.5).float() if c["voicing"] else None
        with torch.no_grad():
            logits, features = model(x, voicing, return_features=True)
            top = torch.topk(logits, k=1, dim=-1).indices
        arrays = dict(x=x.numpy(), logits=logits.numpy(), features=features.numpy(), top=top.numpy(),
                      cfg=np.array([c["in_channels"], c["num_residual_layers"], c["num_rnn_layers"], c["rnn_hidden_size"],
                                    c["num_classes"], c["num_features"], c["adapter_out_features"] or 0], dtype=np.int64))
        if voicing is not None:
            arrays["voicing"] = voicing.numpy()
        arrays.update(sd_to_np("w.", model.state_dict()))
        save(name, **arrays)
        srt = logits.sort(dim=-1).values
        out[name] = dict(logit_sum=float(logits.sum()), params=model.total_parameters, n_keys=len(model.state_dict()), seed=c["seed"],
                         init_abs_sum=init_abs_sum, min_top2_gap=float((srt[..., -1] - srt[..., -2]).min()))
    return out


def gen_principal_components(pc_rnn):
    """PrincipalComponentsArtSpeech (principal_components/models/rnn.py:36-109) with both cells of the RNNType switch:
    forward on ragged batches + every parameter gradient of sum(out * dout)."""
    out = {}
    cases = {
        "pc_lstm_small": dict(rnn="lstm", vocab=13, comps={"tongue": 4, "lower-lip": 3}, embed=16, hidden=32, lengths=[9, 7, 4, 1], seed=41),
        "pc_gru_small": dict(rnn="gru", vocab=11, comps={"tongue": 5, "pharynx": 2, "upper-lip": 1}, embed=24, hidden=64, lengths=[6, 6, 2], seed=42),
    }
    for name, c in cases.items():
        torch.manual_seed(c["seed"])
        model = pc_rnn.PrincipalComponentsArtSpeech(c["vocab"], c["comps"], embed_dim=c["embed"], hidden_size=c["hidden"], rnn=c["rnn"])
        init_abs_sum = float(sum(p.detach().double().abs().sum() for p in model.parameters()))  # seed-for-seed init check
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.LayerNorm):
                    m.weight.uniform_(0.7, 1.3)
                    m.bias.uniform_(-0.2, 0.2)
        B, T = len(c["lengths"]), max(c["lengths"])
        tokens = torch.randint(1, c["vocab"], (B, T))
        for i, l in enumerate(c["lengths"]):
            tokens[i, l:] = 0
        lengths = torch.tensor(c["lengths"])
        y = model(tokens, lengths)
        dout = torch.rand_like(y)
        (y * dout).sum().backward()
        arrays = dict(tokens=tokens.numpy(), lengths=lengths.numpy(), out=y.detach().numpy(), dout=dout.numpy(),
                      cfg=np.array([c["vocab"], c["embed"], c["hidden"], model.latent_size, int(c["rnn"] == "lstm")], dtype=np.int64),
                      comps=np.array(list(c["comps"].values()), dtype=np.int64))
        arrays.update(sd_to_np("w.", model.state_dict()))
        arrays.update({"g." + k: p.grad.numpy() for k, p in model.named_parameters()})
        save(name, **arrays)
        out[name] = dict(out_sum=float(y.sum()), params=model.total_parameters, n_keys=len(model.state_dict()), seed=c["seed"],
                         init_abs_sum=init_abs_sum, comps=c["comps"])
    return out


def gen_pc_autoencoder(ae, losses):
    """MultiArticulatorAutoencoder (principal_components/models/autoencoder.py:216-260) forward + parameter gradients, and
    CriticalLoss (losses.py:23-99) value + gradient w.r.t. the predicted shapes (N = 12 points <= 25: cdist computes the
    differences directly, no matmul expansion)."""
    torch.manual_seed(51)
    comps = {"tongue": 4, "lower-lip": 3, "upper-lip": 2}
    model = ae.MultiArticulatorAutoencoder(in_features=20, indices_dict=comps, hidden_features=16)
    init_abs_sum = float(sum(p.detach().double().abs().sum() for p in model.parameters()))
    x = torch.rand(6, 3, 20)
    out, latent = model(x)
    dout, dlat = torch.rand_like(out), torch.rand_like(latent)
    ((out * dout).sum() + (latent * dlat).sum()).backward()
    arrays = dict(x=x.numpy(), out=out.detach().numpy(), latent=latent.detach().numpy(), dout=dout.numpy(), dlat=dlat.numpy(),
                  comps=np.array(list(comps.values()), dtype=np.int64))
    arrays.update(sd_to_np("w.", model.state_dict()))
    arrays.update({"g." + k: p.grad.numpy() for k, p in model.named_parameters()})
    save("pc_autoencoder", **arrays)

    torch.manual_seed(52)
    arts = ["lower-lip", "tongue", "upper-lip"]          # sorted, upper incisor injected from the reference contour
    crit = losses.CriticalLoss(["TTCD", "LA"], list(arts))
    shapes = torch.rand(2, 5, 3, 2, 12, requires_grad=True)
    targets = torch.rand(2, 5, 3, 2, 12)
    ref = torch.rand(2, 5, 1, 2, 12)
    mask = (torch.rand(2, 2, 5) > 0.4).float()
    mask[0, 0, 0] = 1.0
    loss = crit(shapes, targets, ref, mask)
    loss.backward()
    save("pc_critical_loss", shapes=shapes.detach().numpy(), targets=targets.numpy(), ref=ref.numpy(), mask=mask.numpy(),
         loss=np.array(loss.item()), dshapes=shapes.grad.numpy())
    return {"pc_autoencoder": dict(out_sum=float(out.sum()), seed=51, init_abs_sum=init_abs_sum, params=model.total_parameters,
                                   comps=comps),
            "pc_critical_loss": dict(loss=float(loss))}


# ----------------------------------------------------------------------------- test / training loops
TV_ARTS = sorted(["lower-lip", "pharynx", "soft-palate-midline", "tongue", "upper-lip"])  # + injected upper incisor


class _CapturedLoader:
    """A DataLoader stand-in: fixed list of collated batches + `.dataset.dataset_config` (evaluation.py:35)."""

    def __init__(self, batches, dataset_config):
        self.batches = batches
        self.dataset = types.SimpleNamespace(dataset_config=dataset_config)

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def gen_test_loops(models, dataset, p2a_metrics, ed_metrics, settings, ref_eval, ref_train):
    """The reference's own harnesses on a captured 6-utterance loader (2 batches of 3):
      * run_epoch(TRAIN) with SGD (train_phoneme_to_articulation.py:45-121)  -> info + the parameters it leaves,
      * run_epoch(VALID) with fn_metrics = {p2cp_mean: P2CPDistance}        -> info,
      * run_test (encoder_decoder/evaluation.py:17-161, regularize_out=False) -> info dict, tract_variables.csv, contour dumps.
    SGD instead of the trainer's Adam: an Adam step is ~lr * sign(g) for every element, so parameters whose gradient is
    rounding noise differ by 2 * lr between two correct fp32 implementations; with SGD the update is proportional to the
    gradient and the comparison is meaningful."""
    import csv
    import tempfile
    torch.manual_seed(21)
    V, A, E, H, N = 20, len(TV_ARTS), 32, 64, 50
    lens = [13, 9, 11, 4, 17, 1]
    items = []
    for i, l in enumerate(lens):
        items.append((
            f"sent{i}", torch.randint(2, V, (l,)), torch.rand(l, A, 2, N), [f"ph{int(t)}" for t in torch.randint(0, 9, (l,))],
            torch.rand(l, 1, 2, N), torch.tensor([], dtype=torch.int), [f"{1000 * i + j:04d}" for j in range(l)],
            (torch.rand(l) > 0.5).float(),
        ))
    batches = [dataset.pad_sequence_collate_fn(items[:3]), dataset.pad_sequence_collate_fn(items[3:])]
    cfg = settings.DATASET_CONFIG["artspeech2"]
    loader = _CapturedLoader(batches, cfg)
    model = models.ArtSpeech(V, A, embed_dim=E, hidden_size=H, n_samples=N)
    w0 = {k: v.copy() for k, v in sd_to_np("w0.", model.state_dict()).items()}  # copies: SGD updates the parameters in place
    crit = p2a_metrics.EuclideanDistance("none")
    cpu = torch.device("cpu")
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    train_info = ref_train.run_epoch(settings.TRAIN, 1, model, loader, opt, crit, device=cpu)
    w1 = sd_to_np("w1.", model.state_dict())
    valid_info = ref_train.run_epoch(settings.VALID, 1, model, loader, opt, crit,
                                     fn_metrics={"p2cp_mean": ed_metrics.P2CPDistance(cfg)}, device=cpu)
    with tempfile.TemporaryDirectory() as d:
        test_info = ref_eval.run_test(7, model, loader, crit, d, TV_ARTS, device=cpu, regularize_out=False)
        sdir = os.path.join(d, "7", "sent4")
        with open(os.path.join(sdir, "tract_variables.csv")) as f:
            rows = list(csv.reader(f))
        tv_cols, tv_rows = rows[0], rows[1:]
        with open(os.path.join(sdir, "phonemes.csv")) as f:
            ph_rows = list(csv.reader(f))
        contour_files = sorted(os.listdir(os.path.join(sdir, "contours")))
        frame0 = items[4][6][0]
        pred_tongue = np.load(os.path.join(sdir, "contours", f"{frame0}_tongue.npy"))
        true_incisor = np.load(os.path.join(sdir, "contours", f"{frame0}_upper-incisor_true.npy"))
        n_dirs = len(os.listdir(os.path.join(d, "7")))
    num = [c for c in tv_cols if c not in ("sentence", "frame", "phoneme")]
    arrays = dict(
        cfg=np.array([V, A, E, H, N], dtype=np.int64), articulators=np.array(TV_ARTS), lens=np.array(lens),
        train_loss=np.float64(train_info["loss"]), valid_loss=np.float64(valid_info["loss"]),
        valid_p2cp_mean=np.float64(valid_info["p2cp_mean"]), test_loss=np.float64(test_info["loss"]),
        test_metric_names=np.array(["x_corr", "y_corr", "p2cp", "p2cp_mm", "med", "med_mm"]),
        test_metrics=np.array([[test_info[a][k] for k in ("x_corr", "y_corr", "p2cp", "p2cp_mm", "med", "med_mm")] for a in TV_ARTS],
                              dtype=np.float64),
        tv_columns=np.array(tv_cols), tv_numeric_columns=np.array(num),
        tv_values=np.array([[float(r[tv_cols.index(c)]) for c in num] for r in tv_rows], dtype=np.float64),
        tv_frames=np.array([r[tv_cols.index("frame")] for r in tv_rows]),
        tv_phonemes=np.array([r[tv_cols.index("phoneme")] for r in tv_rows]),
        phonemes_csv=np.array(ph_rows), contour_files=np.array(contour_files), n_sentence_dirs=np.int64(n_dirs),
        pred_tongue_frame0=pred_tongue, true_incisor_frame0=true_incisor,
    )
    for i, it in enumerate(items):
        arrays[f"in{i}_id"] = np.array(it[0])
        arrays[f"in{i}_tokens"], arrays[f"in{i}_targets"], arrays[f"in{i}_refs"] = it[1].numpy(), it[2].numpy(), it[4].numpy()
        arrays[f"in{i}_phonemes"], arrays[f"in{i}_frames"], arrays[f"in{i}_voicing"] = np.array(it[3]), np.array(it[6]), it[7].numpy()
    arrays.update(w0)
    arrays.update(w1)
    save("test_loops", **arrays)
    return dict(train_loss=float(train_info["loss"]), valid_loss=float(valid_info["loss"]),
                valid_p2cp_mean=float(valid_info["p2cp_mean"]), test_loss=float(test_info["loss"]),
                tongue=dict(test_info["tongue"]))


def perturb_by_key(i, k, v):
    """Deterministic change of state_dict entry number i (keys sorted): v + 0.02 * cos(0.37 * arange + i), except
      * pos_encoding.pe, the sinusoidal table (a persistent buffer, not a parameter): left alone;
      * tgt_embedding.1.weight: the perturbed value times 0.1.  At its seeded initialisation the free-running generate()
        is chaotic (a 1e-6 difference between two correct fp32 implementations grows ~3.5x per generated frame: 0.47 after
        17 frames, measured against the fp64 oracle), so nothing downstream of it could be pinned; with the fed-back frame
        entering at a tenth of the weight the recursion is contractive and implementations agree to ~2e-6 at every frame."""
    if k == "pos_encoding.pe":
        return v
    n = v.numel()
    out = v + (0.02 * torch.cos(0.37 * torch.arange(n, dtype=torch.float64) + i)).to(v.dtype).view(v.shape)
    return out * 0.1 if k == "tgt_embedding.1.weight" else out


def _patch_decoder_loop(model):
    """torch 2.0.1's nn.TransformerDecoder.forward (the reference's pin, requirements.txt:21): a plain loop over the layers.
    torch 2.10's version probes layers[0].self_attn, absent on the custom layer; oracle-side shim only."""
    import types as _t

    def loop_forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                     memory_key_padding_mask=None, **_):
        out = tgt
        for mod in self.layers:
            out = mod(out, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                      tgt_key_padding_mask=tgt_key_padding_mask, memory_key_padding_mask=memory_key_padding_mask)
        return out
    model.decoder.forward = _t.MethodType(loop_forward, model.decoder)


def load_reference_transformer_harnesses(pkg, settings, root_metrics):
    """transformer/evaluation.py and train_phoneme_to_articulation_transformer.py imported as they are (after
    load_reference_harnesses has put save_outputs / tract_variables on the package and redirected settings.BASE_DIR).
    The training script switches autograd's anomaly detection on at import (:47); it is switched off again here."""
    sys.modules["metrics"] = root_metrics
    tpkg = types.ModuleType("phoneme_to_articulation.transformer")
    tpkg.__path__ = [os.path.join(REF, "phoneme_to_articulation/transformer")]
    sys.modules["phoneme_to_articulation.transformer"] = tpkg
    sys.modules["phoneme_to_articulation.transformer.models"] = sys.modules["ref_transformer_models"]
    ref_teval = _load("phoneme_to_articulation.transformer.evaluation", "phoneme_to_articulation/transformer/evaluation.py")
    ref_ttrain = _load("ref_train_transformer_script", "train_phoneme_to_articulation_transformer.py")
    torch.autograd.set_detect_anomaly(False)
    return ref_teval, ref_ttrain


def gen_transformer_loops(tmod, dataset, p2a_metrics, ed_metrics, settings, ref_teval, ref_ttrain):
    """The reference's transformer harnesses on a captured 6-utterance loader (2 batches of 3):
      * run_epoch(TRAIN) with SGD (train_phoneme_to_articulation_transformer.py:49-149) -> info + the parameters it leaves,
      * run_epoch(VALID) with fn_metrics = {p2cp_mean: P2CPDistance}                   -> info,
      * run_transformer_test (transformer/evaluation.py:19-191, regularize_out=False) on the same loader with ONE utterance
        made invalid -- its source key-padding mask masks every position, so the encoder's softmax sees only -inf and the
        prediction is NaN -- so that the NaN filter (:69-86) executes: info dict, the skipped sentence, CSVs, contour dumps.
    Dropout: the encoder layers keep the library default p = 0.1 (the model's argument does not reach them), a
    random mask that no other implementation can reproduce, so every dropout probability of the fixture model is set to 0
    (nn.Dropout modules and the encoder self-attention's); what is pinned is the deterministic part of the loops.
    NOTE (reference behaviour, kept): after the filter the reference zips the KEPT outputs with the UNFILTERED
    sentence ids / lengths / frames / phonemes (evaluation.py:96, 146-168), so kept utterance j is reported under the j-th
    entry of the unfiltered batch; only the loss uses the filtered padding mask (:81-86)."""
    import csv
    import io
    import tempfile
    from contextlib import redirect_stdout
    V, A, d, heads, L, N = 20, len(TV_ARTS), 32, 4, 2, 50
    # the model first, straight from the seed: the drop-in class reproduces the reference's seeded initialisation (a tested
    # contract), so the 1.3 M initial weights need not be stored; then a perturbation that is a pure function of the
    # state_dict key order (tells the deep-copied decoder layers apart, makes the LayerNorm affines non-trivial)
    torch.manual_seed(33)
    model = tmod.ArtSpeechTransformer(V, A, embed_dim=d, num_heads=heads, num_layers=L, num_feat=2 * N)
    _patch_decoder_loop(model)
    sd = model.state_dict()
    init_abs_sum = float(sum(v.double().abs().sum() for v in sd.values()))
    sd = {k: perturb_by_key(i, k, v) for i, (k, v) in enumerate(sorted(sd.items()))}
    model.load_state_dict(sd)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    torch.manual_seed(34)
    lens = [13, 9, 11, 4, 17, 6]
    items = []
    for i, l in enumerate(lens):
        items.append((
            f"sent{i}", torch.randint(2, V, (l,)), torch.rand(l, A, 2, N), [f"ph{int(t)}" for t in torch.randint(0, 9, (l,))],
            torch.rand(l, 1, 2, N), torch.tensor([], dtype=torch.int), [f"{1000 * i + j:04d}" for j in range(l)],
            (torch.rand(l) > 0.5).float(),
        ))
    batches = [dataset.pad_sequence_transformer_collate_fn(items[:3]), dataset.pad_sequence_transformer_collate_fn(items[3:])]
    cfg = settings.DATASET_CONFIG["artspeech2"]
    loader = _CapturedLoader(batches, cfg)
    w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    crit = p2a_metrics.EuclideanDistance("none")
    cpu = torch.device("cpu")
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    # at the initial weights (what the CPU oracle is checked against: it has no optimizer): VALID info and the test info
    valid0_info = ref_ttrain.run_epoch(settings.VALID, 0, model, loader, opt, crit,
                                       fn_metrics={"p2cp_mean": ed_metrics.P2CPDistance(cfg)}, device=cpu)
    tb0 = [list(b) for b in batches]
    tb0[0][8] = tb0[0][8].clone()
    tb0[0][8][1] = float("-inf")
    with tempfile.TemporaryDirectory() as dd0, redirect_stdout(io.StringIO()):
        test0_info = ref_teval.run_transformer_test(0, model, _CapturedLoader([tuple(b) for b in tb0], cfg), crit, dd0, TV_ARTS,
                                                    device=cpu, regularize_out=False)
    train_info = ref_ttrain.run_epoch(settings.TRAIN, 1, model, loader, opt, crit, device=cpu)
    # the parameter update of the epoch (two SGD steps), per tensor: norm, max and a strided slice of w1 - w0
    keys = sorted(model.state_dict())
    sd1 = model.state_dict()
    upd_slices = np.zeros((len(keys), 33), dtype=np.float32)
    upd_norm, upd_max = np.zeros(len(keys)), np.zeros(len(keys))
    for i, k in enumerate(keys):
        dlt = (sd1[k].detach() - w0[k]).double().flatten()
        upd_norm[i], upd_max[i] = float(dlt.norm()), float(dlt.abs().max())
        sl = dlt[::max(1, dlt.numel() // 33)][:33].numpy()
        upd_slices[i, :len(sl)] = sl
    upd = dict(upd_keys=np.array(keys), upd_norm=upd_norm, upd_max=upd_max, upd_slices=upd_slices)
    valid_info = ref_ttrain.run_epoch(settings.VALID, 1, model, loader, opt, crit,
                                      fn_metrics={"p2cp_mean": ed_metrics.P2CPDistance(cfg)}, device=cpu)
    # test loader: batch 0 row 1 (the middle utterance, "sent2" after the length sort) gets an all-masked source
    tb = [list(b) for b in batches]
    nan_row = 1
    tb[0][8] = tb[0][8].clone()
    tb[0][8][nan_row] = float("-inf")
    test_loader = _CapturedLoader([tuple(b) for b in tb], cfg)
    printed = io.StringIO()
    with tempfile.TemporaryDirectory() as dd, redirect_stdout(printed):
        test_info = ref_teval.run_transformer_test(7, model, test_loader, crit, dd, TV_ARTS, device=cpu, regularize_out=False)
        dirs = sorted(os.listdir(os.path.join(dd, "7")))
        tv = {}
        for sd_ in dirs:
            with open(os.path.join(dd, "7", sd_, "tract_variables.csv")) as f:
                rows = list(csv.reader(f))
            tv[sd_] = rows
        sdir = os.path.join(dd, "7", dirs[0])
        contour_files = sorted(os.listdir(os.path.join(sdir, "contours")))
        with open(os.path.join(sdir, "phonemes.csv")) as f:
            ph_rows = list(csv.reader(f))
        first_frame = tv[dirs[0]][1][tv[dirs[0]][0].index("frame")]
        pred_tongue = np.load(os.path.join(sdir, "contours", f"{first_frame}_tongue.npy"))
    out_text = printed.getvalue()
    skipped = [ln.strip() for ln in out_text.split("Invalid outputs produced for sentences:")[1].strip().splitlines() if ln.strip()] \
        if "Invalid outputs produced" in out_text else []
    tv_cols = tv[dirs[0]][0]
    num = [c for c in tv_cols if c not in ("sentence", "frame", "phoneme")]
    arrays = dict(
        cfg=np.array([V, A, d, heads, L, 2 * N], dtype=np.int64), articulators=np.array(TV_ARTS), lens=np.array(lens),
        train_loss=np.float64(train_info["loss"]), valid_loss=np.float64(valid_info["loss"]),
        valid_p2cp_mean=np.float64(valid_info["p2cp_mean"]), test_loss=np.float64(test_info["loss"]),
        test_metric_names=np.array(["x_corr", "y_corr", "p2cp", "p2cp_mm", "med", "med_mm"]),
        test_metrics=np.array([[test_info[a][k] for k in ("x_corr", "y_corr", "p2cp", "p2cp_mm", "med", "med_mm")] for a in TV_ARTS],
                              dtype=np.float64),
        valid0_loss=np.float64(valid0_info["loss"]), valid0_p2cp_mean=np.float64(valid0_info["p2cp_mean"]),
        test0_loss=np.float64(test0_info["loss"]),
        test0_metrics=np.array([[test0_info[a][k] for k in ("x_corr", "y_corr", "p2cp", "p2cp_mm", "med", "med_mm")] for a in TV_ARTS],
                               dtype=np.float64),
        nan_batch=np.int64(0), nan_row=np.int64(nan_row), skipped=np.array(skipped), sentence_dirs=np.array(dirs),
        tv_columns=np.array(tv_cols), tv_numeric_columns=np.array(num),
        phonemes_csv=np.array(ph_rows), contour_files=np.array(contour_files), pred_tongue_first=pred_tongue,
        first_dir=np.array(dirs[0]), first_frame=np.array(first_frame),
    )
    for sd_ in dirs:
        rows = tv[sd_][1:]
        arrays[f"tv_{sd_}_values"] = np.array([[float(r[tv_cols.index(c)]) for c in num] for r in rows], dtype=np.float64)
        arrays[f"tv_{sd_}_frames"] = np.array([r[tv_cols.index("frame")] for r in rows])
        arrays[f"tv_{sd_}_phonemes"] = np.array([r[tv_cols.index("phoneme")] for r in rows])
    for i, it in enumerate(items):
        arrays[f"in{i}_id"] = np.array(it[0])
        arrays[f"in{i}_tokens"], arrays[f"in{i}_targets"], arrays[f"in{i}_refs"] = it[1].numpy(), it[2].numpy(), it[4].numpy()
        arrays[f"in{i}_phonemes"], arrays[f"in{i}_frames"], arrays[f"in{i}_voicing"] = np.array(it[3]), np.array(it[6]), it[7].numpy()
    arrays.update(upd)
    arrays["init_abs_sum"] = np.float64(init_abs_sum)
    save("transformer_loops", **arrays)
    return dict(train_loss=float(train_info["loss"]), valid_loss=float(valid_info["loss"]),
                valid_p2cp_mean=float(valid_info["p2cp_mean"]), test_loss=float(test_info["loss"]),
                skipped=skipped, sentence_dirs=dirs, tongue=dict(test_info["tongue"]))


def load_reference_harnesses(pkg, settings, models, dataset, p2a_metrics, ed_metrics, root_metrics, helpers):
    """encoder_decoder/evaluation.py and train_phoneme_to_articulation.py, imported as they are.  Name-only shims for the
    absent mlflow / ujson (module-level imports, never called here); settings.BASE_DIR is pointed at a scratch directory
    because the training script creates BASE_DIR/tmp/<random> at import time (the reference tree is read-only)."""
    import tempfile
    _shim("vt_tools.bs_regularization", regularize_Bsplines=None)
    ref_init = _load("ref_p2a_init_for_loops", "phoneme_to_articulation/__init__.py")
    pkg.save_outputs, pkg.tract_variables = ref_init.save_outputs, ref_init.tract_variables
    pkg.REQUIRED_ARTICULATORS_FOR_TVS = ref_init.REQUIRED_ARTICULATORS_FOR_TVS
    sys.modules["metrics"] = root_metrics            # evaluation.py:8 `from metrics import ...`
    ref_eval = _load("phoneme_to_articulation.encoder_decoder.evaluation", "phoneme_to_articulation/encoder_decoder/evaluation.py")
    _shim("mlflow")
    _shim("ujson")
    scratch = tempfile.mkdtemp()
    os.makedirs(os.path.join(scratch, "tmp"), exist_ok=True)
    settings.BASE_DIR = scratch
    sys.modules["phoneme_to_articulation.encoder_decoder.dataset"] = dataset
    sys.modules["phoneme_to_articulation.encoder_decoder.models"] = models
    dataset.ArtSpeechDataset = getattr(dataset, "ArtSpeechDataset", None)
    ref_train = _load("ref_train_script", "train_phoneme_to_articulation.py")
    return ref_eval, ref_train


def gen_artspeech_c2(models, p2a_metrics, helpers):
    """BASELINE configs[1] at FULL size through the reference itself: ArtSpeech(45, 11), B=32, T=200, parity lengths
    linspace(200, 60, 32).  Weights (7.5 MB) and targets (28 MB) are too big for a fixture: both are functions
    of torch's seeded CPU generator, which the test re-runs (seed-for-seed identical initialisation is itself a tested
    contract, tests/test_host.py); the fixture holds checksums of the regenerated inputs, the loss, slices of the
    contours, and per-tensor norms / slices of every gradient."""
    torch.manual_seed(0)
    V, A, E, H, N, B, T = 45, 11, 64, 128, 50, 32, 200
    model = models.ArtSpeech(V, A, embed_dim=E, hidden_size=H, n_samples=N)
    x = torch.randint(1, V, (B, T))
    lengths = torch.linspace(200, 60, B).int()
    tgt = torch.rand(B, T, A, 2, N)
    for i, l in enumerate(lengths):
        x[i, l:] = 0
        tgt[i, l:] = 0
    out = model(x, lengths)
    loss = masked_mean_loss(p2a_metrics.EuclideanDistance("none"), helpers.make_padding_mask, out, tgt, lengths)
    loss.backward()
    pos = [(0, 0), (0, 199), (5, 100), (17, 3), (31, 59), (31, 60), (12, 150)]
    arrays = dict(
        cfg=np.array([V, A, E, H, N, B, T], dtype=np.int64), lengths=lengths.numpy(), positions=np.array(pos),
        x_sum=np.int64(x.sum().item()), tgt_sum=np.float64(tgt.double().sum().item()),
        w_sum=np.float64(sum(p.double().sum().item() for p in model.parameters())),
        loss=np.float64(loss.item()), out_sum=np.float64(out.double().sum().item()),
        out_slices=np.stack([out[b, t].detach().numpy() for b, t in pos]),
    )
    for k, p in model.named_parameters():
        g = p.grad
        arrays["gnorm." + k] = np.float64(g.double().norm().item())
        arrays["gmax." + k] = np.float64(g.abs().max().item())
        arrays["gslice." + k] = g.reshape(-1)[:: max(1, g.numel() // 257)][:257].numpy().copy()
    save("artspeech_c2_full", **arrays)
    return dict(loss=float(loss), out_sum=float(out.double().sum()))


C4_SLICE = 33   # strided gradient elements kept per tensor


def c4_case(model_cls, dataset_collate, seed=0):
    """The seeded recipe of the full-width transformer case (shared text with tests/test_gpu_transformer.py, which re-runs it
    with the build's classes): BASELINE configs[3]'s model (V=45, A=11, d=256, 4 heads, 6 layers, 100 features) on a
    two-utterance ragged batch of T=200.  Decoder layers are deep copies of one layer at construction; layer l's tensors are
    scaled by 1 + 0.01 (l + 1) (by state_dict key, so the order of parameters does not matter) to tell the layers apart."""
    torch.manual_seed(seed)
    V, A, d, heads, L, nf = 45, 11, 256, 4, 6, 100
    model = model_cls(V, A, embed_dim=d, num_heads=heads, num_layers=L, num_feat=nf)
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            if k.startswith("decoder.layers.") and v.dtype.is_floating_point:
                v.mul_(1.0 + 0.01 * (int(k.split(".")[2]) + 1))
    model.load_state_dict(sd)
    lens = [200, 140]
    batch = [(f"s{i}", torch.randint(1, V, (l,)), torch.rand(l, A, 2, nf // 2), ["p"] * l, torch.rand(l, 1, 2, nf // 2),
              torch.tensor([], dtype=torch.int), list(range(l)), torch.zeros(l)) for i, l in enumerate(lens)]
    c = dataset_collate(batch)
    tokens, targets, lengths = c[1], c[2], c[3]
    bs, T = tokens.shape
    shifted = torch.cat([torch.zeros(bs, 1, A, nf), targets[:, 1:].reshape(bs, T - 1, A, nf)], dim=1)
    kw = dict(src_key_padding_mask=c[8], tgt_key_padding_mask=c[9], src_attn_mask=c[10], tgt_attn_mask=c[11])
    return model, tokens, targets, lengths, shifted, kw, (V, A, d, heads, L, nf)


def _c4_run(tmod, dataset, p2a_metrics, helpers):
    """One forward + loss + backward of the full-width case through the reference; returns the arrays of the fixture."""
    import types as _t
    model, tokens, targets, lengths, shifted, kw, cfg = c4_case(tmod.ArtSpeechTransformer, dataset.pad_sequence_transformer_collate_fn)

    def loop_forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                     memory_key_padding_mask=None, **_):
        out = tgt
        for mod in self.layers:
            out = mod(out, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                      tgt_key_padding_mask=tgt_key_padding_mask, memory_key_padding_mask=memory_key_padding_mask)
        return out
    model.decoder.forward = _t.MethodType(loop_forward, model.decoder)   # torch 2.0.1's decoder loop (see gen_transformer)
    model.eval()
    out = model(tokens, shifted, **kw)
    loss = masked_mean_loss(p2a_metrics.EuclideanDistance("none"), helpers.make_padding_mask, out, targets, lengths)
    loss.backward()
    pos = [(0, 0), (0, 199), (0, 77), (1, 0), (1, 139), (1, 140), (1, 199)]
    arrays = dict(
        cfg=np.array(cfg, dtype=np.int64), lengths=lengths.numpy(), positions=np.array(pos),
        tok_sum=np.int64(tokens.sum().item()), tgt_sum=np.float64(targets.double().sum().item()),
        w_abs_sum=np.float64(sum(p.detach().double().abs().sum().item() for p in model.parameters())),
        loss=np.float64(loss.item()), out_sum=np.float64(out.detach().double().sum().item()),
        out_slices=np.stack([out[b, t].detach().numpy() for b, t in pos]),
        params=np.int64(sum(p.numel() for p in model.parameters())),
    )
    # ~10 000 parameter tensors: one row each in packed arrays (an .npz member per tensor would cost 20 MB of zip headers)
    names, gnorm, gmax, gslice = [], [], [], np.zeros((len(list(model.parameters())), C4_SLICE), np.float32)
    for i, (k, p) in enumerate(model.named_parameters()):
        g = p.grad
        names.append(k)
        gnorm.append(g.double().norm().item())
        gmax.append(g.abs().max().item())
        sl = g.reshape(-1)[:: max(1, g.numel() // C4_SLICE)][:C4_SLICE].numpy()
        gslice[i, :sl.size] = sl
    arrays.update(names=np.array("\n".join(names)), gnorm=np.array(gnorm), gmax=np.array(gmax), gslice=gslice)
    return arrays


C4_QUANTILES = (0.5, 0.9, 0.99, 1.0)


def gen_transformer_c4(tmod, dataset, p2a_metrics, helpers):
    """Full-width ArtSpeechTransformer (419.6 M parameters) through the reference: forward as the trainer calls it, the masked
    Euclidean loss of train_phoneme_to_articulation_transformer.py:113-118, backward.  Weights / inputs are functions of the
    seeded generator (the test re-runs c4_case); the fixture holds their checksums, the loss, contour slices, and norm, max
    and a strided slice of every parameter gradient.  eval() mode: the encoder's library-default dropout 0.1 off.

    The reference is then run AGAINST ITSELF with torch.set_num_threads(1) (another summation order inside its matmuls,
    nothing else changes): ~1e8 ReLU decisions stand between the loss and the early layers, a few hundred of them within an
    ulp of zero, and each one that falls the other way moves gradient elements by a whole frame's term (340 valid frames
    only).  The quantiles of that self-discrepancy (per-tensor norm error, per-tensor slice error / max|g|) go into the
    fixture: they are the yardstick the GPU test is held to, instead of a tolerance picked by hand."""
    arrays = _c4_run(tmod, dataset, p2a_metrics, helpers)
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        again = _c4_run(tmod, dataset, p2a_metrics, helpers)
    finally:
        torch.set_num_threads(nt)
    ne = np.abs(again["gnorm"] - arrays["gnorm"]) / arrays["gnorm"]
    se = np.abs(again["gslice"] - arrays["gslice"]).max(1) / arrays["gmax"]
    # per tensor as well (the test groups them by depth: heads, last decoder layer, ..., encoder) and per contour position
    cr = np.abs(again["out_slices"] - arrays["out_slices"]) / (1e-4 * np.abs(arrays["out_slices"]) + 2e-6)
    arrays.update(self_norm_err=ne.astype(np.float32), self_slice_err=se.astype(np.float32),
                  self_contour_ratio=cr.reshape(cr.shape[0], -1).max(1))
    arrays.update(self_norm_q=np.quantile(ne, C4_QUANTILES), self_slice_q=np.quantile(se, C4_QUANTILES),
                  self_contour_diff=np.float64(np.abs(again["out_slices"] - arrays["out_slices"]).max()),
                  self_loss_diff=np.float64(abs(float(again["loss"]) - float(arrays["loss"]))), self_threads=np.array([nt, 1]))
    save("transformer_c4_full", **arrays)
    return dict(loss=float(arrays["loss"]), out_sum=float(arrays["out_sum"]), params=int(arrays["params"]),
                self_norm_q=[float(v) for v in arrays["self_norm_q"]], self_slice_q=[float(v) for v in arrays["self_slice_q"]])


def main():
    only = set(sys.argv[1:])  # e.g. `make_golden.py test_loops`: regenerate just that fixture
    install_shims()
    sys.path.insert(0, REF)  # for `settings`, `helpers`
    settings = _load("settings", "settings.py")
    helpers = _load("helpers", "helpers.py")
    models = _load("ref_ed_models", "phoneme_to_articulation/encoder_decoder/models.py")
    # make `phoneme_to_articulation.metrics` importable without executing the package __init__
    pkg = types.ModuleType("phoneme_to_articulation")
    pkg.__path__ = [os.path.join(REF, "phoneme_to_articulation")]
    pkg.InputLoaderMixin = object
    sys.modules["phoneme_to_articulation"] = pkg
    p2a_metrics = _load("phoneme_to_articulation.metrics", "phoneme_to_articulation/metrics.py")
    edpkg = types.ModuleType("phoneme_to_articulation.encoder_decoder")
    edpkg.__path__ = [os.path.join(REF, "phoneme_to_articulation/encoder_decoder")]
    sys.modules["phoneme_to_articulation.encoder_decoder"] = edpkg
    ed_metrics = _load("phoneme_to_articulation.encoder_decoder.metrics", "phoneme_to_articulation/encoder_decoder/metrics.py")
    root_metrics = _load("ref_root_metrics", "metrics.py")
    tv = _load("tract_variables", "tract_variables.py")
    af = _load("ref_area_function", "area_function.py")
    # dataset.py imports a few absent things at module level; give it empty shells
    _shim("vt_shape_gen"); _shim("vt_shape_gen.helpers", load_articulator_array=None)
    _shim("database_collector", DATABASE_COLLECTORS={})
    _shim("phoneme_to_articulation.tail_clipper", TailClipper=None)
    dataset = _load("ref_ed_dataset", "phoneme_to_articulation/encoder_decoder/dataset.py")

    if only == {"c2_full"}:
        res = gen_artspeech_c2(models, p2a_metrics, helpers)
        path = os.path.join(OUT, "checksums.json")
        with open(path) as f:
            allc = json.load(f)
        allc["cases"]["artspeech_c2_full"] = res
        with open(path, "w") as f:
            json.dump(allc, f, indent=1)
        print(json.dumps(res, indent=1))
        return
    if only == {"c4_full"}:
        sys.modules["phoneme_to_articulation.encoder_decoder.models"] = models
        tmod = _load("ref_transformer_models", "phoneme_to_articulation/transformer/models.py")
        res = gen_transformer_c4(tmod, dataset, p2a_metrics, helpers)
        path = os.path.join(OUT, "checksums.json")
        with open(path) as f:
            allc = json.load(f)
        allc["cases"]["transformer_c4_full"] = res
        with open(path, "w") as f:
            json.dump(allc, f, indent=1)
        print(json.dumps(res, indent=1))
        return
    if only == {"transformer_loops"}:
        sys.modules["phoneme_to_articulation.encoder_decoder.models"] = models
        tmod = _load("ref_transformer_models", "phoneme_to_articulation/transformer/models.py")
        load_reference_harnesses(pkg, settings, models, dataset, p2a_metrics, ed_metrics, root_metrics, helpers)
        ref_teval, ref_ttrain = load_reference_transformer_harnesses(pkg, settings, root_metrics)
        res = gen_transformer_loops(tmod, dataset, p2a_metrics, ed_metrics, settings, ref_teval, ref_ttrain)
        path = os.path.join(OUT, "checksums.json")
        with open(path) as f:
            allc = json.load(f)
        allc["cases"]["transformer_loops"] = res
        with open(path, "w") as f:
            json.dump(allc, f, indent=1)
        print(json.dumps(res, indent=1))
        return
    if only == {"test_loops"}:
        ref_eval, ref_train = load_reference_harnesses(pkg, settings, models, dataset, p2a_metrics, ed_metrics, root_metrics, helpers)
        res = gen_test_loops(models, dataset, p2a_metrics, ed_metrics, settings, ref_eval, ref_train)
        path = os.path.join(OUT, "checksums.json")
        with open(path) as f:
            allc = json.load(f)
        allc["cases"]["test_loops"] = res
        with open(path, "w") as f:
            json.dump(allc, f, indent=1)
        print(json.dumps(res, indent=1))
        return
    checks = {}
    # C1 (SURVEY 8c recipe): ArtSpeech(45, 2), B=4 T=50 lengths [50,40,30,20]
    checks["artspeech_c1"] = gen_artspeech(
        "artspeech_c1", models, p2a_metrics, ed_metrics, helpers, settings,
        vocab=45, n_art=2, embed_dim=64, hidden=128, n_samples=50, B=4, T=50, lengths=[50, 40, 30, 20], seed=0)
    # small, odd sizes, ragged down to length 1, max(len) == T
    checks["artspeech_small"] = gen_artspeech(
        "artspeech_small", models, p2a_metrics, ed_metrics, helpers, settings,
        vocab=13, n_art=3, embed_dim=16, hidden=32, n_samples=10, B=5, T=12, lengths=[12, 12, 9, 4, 1], seed=1)
    # hidden 64 (the other width the reference's configs use), single utterance
    checks["artspeech_h64"] = gen_artspeech(
        "artspeech_h64", models, p2a_metrics, ed_metrics, helpers, settings,
        vocab=9, n_art=1, embed_dim=24, hidden=64, n_samples=7, B=1, T=5, lengths=[5], seed=2)
    gen_simple("simple_small", models, vocab=11, n_art=2, embed_dim=16, hidden=32, n_samples=10, B=3, T=7, seed=3)
    gen_predictor("predictor_in128", models, in_features=128, n_samples=50, rows=70, seed=4)
    gen_predictor("predictor_in32", models, in_features=32, n_samples=10, rows=18, seed=5)
    gen_metrics(p2a_metrics, ed_metrics, root_metrics, settings)
    gen_tract_variables(tv)
    gen_area_function(af)
    gen_host(helpers, dataset)
    sys.modules["phoneme_to_articulation.encoder_decoder.models"] = models  # transformer/models.py:6 imports it by this name
    tmod = _load("ref_transformer_models", "phoneme_to_articulation/transformer/models.py")
    checks["transformer_small"] = gen_transformer(tmod, dataset)
    checks["transformer_c4_full"] = gen_transformer_c4(tmod, dataset, p2a_metrics, helpers)
    ds2 = _load("ref_deepspeech2", "phoneme_recognition/deepspeech2.py")  # file-path import: the package __init__ needs funcy/seaborn
    checks.update(gen_deepspeech2(ds2))
    # the genuine RNNType enum: execute the reference package __init__ under another name (its absent imports are shimmed)
    _shim("vt_tools.bs_regularization", regularize_Bsplines=None)
    pkg.RNNType = _load("ref_p2a_init", "phoneme_to_articulation/__init__.py").RNNType
    pc_rnn = _load("ref_pc_rnn", "phoneme_to_articulation/principal_components/models/rnn.py")
    checks.update(gen_principal_components(pc_rnn))
    # autoencoders + critical loss: register the reference's package modules under their own names, then load them
    pcpkg = types.ModuleType("phoneme_to_articulation.principal_components")
    pcpkg.__path__ = [os.path.join(REF, "phoneme_to_articulation/principal_components")]
    sys.modules["phoneme_to_articulation.principal_components"] = pcpkg
    mpkg = types.ModuleType("phoneme_to_articulation.principal_components.models")
    mpkg.__path__ = [os.path.join(REF, "phoneme_to_articulation/principal_components/models")]
    sys.modules["phoneme_to_articulation.principal_components.models"] = mpkg
    ae = _load("phoneme_to_articulation.principal_components.models.autoencoder",
               "phoneme_to_articulation/principal_components/models/autoencoder.py")
    for name in ("Encoder", "Decoder", "MultiEncoder", "MultiDecoder"):
        setattr(mpkg, name, getattr(ae, name))
    _load("phoneme_to_articulation.principal_components.transforms", "phoneme_to_articulation/principal_components/transforms.py")
    pc_losses = _load("phoneme_to_articulation.principal_components.losses", "phoneme_to_articulation/principal_components/losses.py")
    checks.update(gen_pc_autoencoder(ae, pc_losses))
    ref_eval, ref_train = load_reference_harnesses(pkg, settings, models, dataset, p2a_metrics, ed_metrics, root_metrics, helpers)
    checks["artspeech_c2_full"] = gen_artspeech_c2(models, p2a_metrics, helpers)
    checks["test_loops"] = gen_test_loops(models, dataset, p2a_metrics, ed_metrics, settings, ref_eval, ref_train)
    ref_teval, ref_ttrain = load_reference_transformer_harnesses(pkg, settings, root_metrics)
    checks["transformer_loops"] = gen_transformer_loops(tmod, dataset, p2a_metrics, ed_metrics, settings, ref_teval, ref_ttrain)
    with open(os.path.join(OUT, "checksums.json"), "w") as f:
        json.dump({"torch": torch.__version__, "numpy": np.__version__, "cases": checks}, f, indent=1)
    print(json.dumps(checks, indent=1))


if __name__ == "__main__":
    main()
