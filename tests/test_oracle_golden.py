"""Pin the CPU oracle against fixtures produced by the reference itself (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from conftest import load_golden, split_wg
from oracle import artspeech_oracle as O


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("name", ["artspeech_c1", "artspeech_small", "artspeech_h64"])
def test_artspeech_fwd_bwd(name):
    g = load_golden(name)
    w, grads = split_wg(g)
    n_art = int(g["cfg"][1])
    out, cache = O.artspeech_fwd(w, g["x"], g["lengths"], n_art)
    assert out.shape == g["out"].shape
    assert np.abs(out - g["out"]).max() < 2e-6
    loss, dout = O.masked_euclid_loss(out, g["targets"], g["lengths"])
    assert abs(loss - g["loss"]) < 1e-6
    og = O.artspeech_bwd(dout, cache, n_art)
    assert set(og) == set(grads)
    for k in grads:
        assert relerr(og[k], grads[k]) < 2e-4, k
    p2cp = O.p2cp_distance_mm(out, g["targets"], g["lengths"], 136 * 1.6176470518112)
    assert abs(p2cp - g["p2cp_mm"]) / g["p2cp_mm"] < 2e-3  # reference's cdist uses the fp32 matmul expansion


def test_artspeech_full_size_oracle_vs_reference_fixture():
    """BASELINE configs[1] at full size (B=32, T=200, A=11): the oracle on the regenerated seeded inputs against what the
    reference itself produced (tests/golden/artspeech_c2_full.npz) - loss, contour slices, gradient norms and slices."""
    import torch
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    g = load_golden("artspeech_c2_full")
    V, A, E, H, N, B, T = (int(v) for v in g["cfg"])
    torch.manual_seed(0)
    model = ArtSpeech(V, A, embed_dim=E, hidden_size=H, n_samples=N)
    x = torch.randint(1, V, (B, T))
    lengths = torch.linspace(200, 60, B).int()
    tgt = torch.rand(B, T, A, 2, N)
    for i, l in enumerate(lengths):
        x[i, l:] = 0
        tgt[i, l:] = 0
    assert int(x.sum()) == int(g["x_sum"]) and abs(tgt.double().sum().item() - float(g["tgt_sum"])) < 1e-6
    w = {k: v.numpy() for k, v in model.state_dict().items()}
    assert abs(sum(v.astype(np.float64).sum() for v in w.values()) - float(g["w_sum"])) < 1e-6
    out, cache = O.artspeech_fwd(w, x.numpy(), lengths.numpy(), A)
    loss, dout = O.masked_euclid_loss(out, tgt.numpy(), lengths.numpy())
    assert abs(loss - float(g["loss"])) < 1e-6
    assert abs(out.sum() - float(g["out_sum"])) < 1e-6 * float(g["out_sum"])
    for (b, t), want in zip(g["positions"], g["out_slices"]):
        assert np.abs(out[b, t] - want).max() < 2e-6
    og = O.artspeech_bwd(dout, cache, A)
    for k, v in og.items():
        assert abs(np.linalg.norm(v) - float(g["gnorm." + k])) < 2e-4 * float(g["gnorm." + k]), k
        sl = v.reshape(-1)[:: max(1, v.size // 257)][:257]
        # 18 M ReLU decisions per head layer at this size: a handful of pre-activations sit within an ulp of zero and
        # are decided differently by any two evaluation orders (4 differ between this oracle in fp64 and in fp32); each
        # moves a bias gradient by one full frame term, measured up to 2.5e-3 of max|g| against the reference's fp32 run
        assert np.abs(sl - g["gslice." + k]).max() < 5e-3 * float(g["gmax." + k]), k


def test_artspeech_fp32_mode():
    g = load_golden("artspeech_small")
    w, _ = split_wg(g)
    out, _ = O.artspeech_fwd(w, g["x"], g["lengths"], int(g["cfg"][1]), dtype=np.float32)
    assert out.dtype == np.float32
    assert np.abs(out - g["out"]).max() < 5e-6


def test_padded_frames_are_not_zero():
    # heads run on padded frames too (GRU output is zero there, LayerNorm/bias make outputs non-zero)
    g = load_golden("artspeech_c1")
    assert np.abs(g["out"][3, 25:]).min() > 0


def test_simple_artspeech():
    g = load_golden("simple_small")
    w, grads = split_wg(g)
    n_art = int(g["cfg"][1])
    out, cache = O.simple_artspeech_fwd(w, g["x"], n_art)
    assert np.abs(out - g["out"]).max() < 2e-6
    og = O.simple_artspeech_bwd(g["dout"].astype(np.float64), cache, n_art)
    for k in grads:
        assert relerr(og[k], grads[k]) < 2e-4, k


@pytest.mark.parametrize("name", ["predictor_in128", "predictor_in32"])
def test_predictor(name):
    g = load_golden(name)
    w, grads = split_wg(g)
    p = {k: v.astype(np.float64) for k, v in w.items()}
    out, cache = O.predictor_fwd(g["x"].astype(np.float64), p)
    assert relerr(out, g["out"]) < 2e-6
    dx, og = O.predictor_bwd(g["dout"].astype(np.float64), cache, p)
    assert relerr(dx, g["dx"]) < 1e-5
    for k in grads:
        assert relerr(og[k], grads[k]) < 1e-5, k


def test_metrics():
    g = load_golden("metrics")
    out, tgt = g["out"], g["tgt"]
    assert relerr(O.euclidean_distance(out.astype(np.float64), tgt.astype(np.float64)), g["euc_none"]) < 1e-6
    full = np.full(out.shape[0], out.shape[1])
    loss, grad = O.masked_euclid_loss(out, tgt, full)
    assert abs(loss - g["euc_mean"]) < 1e-6
    assert relerr(grad, g["euc_mean_grad"]) < 1e-5
    # direct formula vs the reference (which uses the fp32 matmul expansion for N=50): loose
    assert relerr(O.p2cp_distance(out, tgt), g["p2cp_none"]) < 2e-3
    assert relerr(O.p2cp_distance(out, tgt), g["root_p2cp"]) < 2e-3
    # the fp32 expansion restated: closer to the reference than the direct formula is
    mm = O.mean_p2cp_mm(np.swapaxes(out, -1, -2), np.swapaxes(tgt, -1, -2))
    assert relerr(mm, g["p2cp_none"]) < 5e-4
    # N, M <= 25: the reference takes the direct path -> tight
    assert relerr(O.mean_p2cp(g["u10"], g["v12"]), g["p2cp_small"]) < 1e-6
    for db, to_mm in (("artspeech2", 136 * 1.6176470518112), ("gottingen", 136 * 1.4117647409439)):
        v = O.p2cp_distance_mm(out, tgt, g["lengths"], to_mm)
        assert abs(v - g[f"p2cp_mm_{db}"]) / g[f"p2cp_mm_{db}"] < 2e-3
    assert relerr(O.euclidean_distance_metric(out.astype(np.float64), tgt.astype(np.float64)), g["root_euclid"]) < 1e-6
    xc, yc = O.pearsons_correlation(out.astype(np.float64), tgt.astype(np.float64))
    assert np.abs(xc - g["x_corr"]).max() < 1e-5 and np.abs(yc - g["y_corr"]).max() < 1e-5


def test_tract_variables():
    g = load_golden("tract_variables")
    arts = [str(a) for a in g["articulators"]]
    for f in range(g["frames"].shape[0]):
        vals, p1, p2, _ = O.tract_variables(g["frames"][f], arts, dtype=np.float32)
        # TTCD (15x25) takes torch.cdist's direct path -> tight; LA/TBCD/VEL have a side > 25 and go
        # through the reference's fp32 matmul expansion, whose error grows as 1/d for small d
        assert abs(vals[1] - g["values"][f, 1]) < 1e-6
        assert np.abs(vals - g["values"][f]).max() < 1e-4
        # closest-point pairs must be the very same points (bit-exact coordinates)
        assert np.array_equal(p1.astype(np.float32), g["poc1"][f])
        assert np.array_equal(p2.astype(np.float32), g["poc2"][f])


def test_area_function():
    g = load_golden("area_function")
    for i in range(4):
        d, fx = O.area_function(g[f"int{i}"], g[f"ext{i}"])
        assert d.shape == g[f"dists{i}"].shape
        assert np.abs(d - g[f"dists{i}"]).max() < 1e-13 and np.abs(fx - g[f"fx{i}"]).max() < 1e-13
    d, fx = O.area_function(g["int0"], g["ext0"], alpha=1.5, beta=1.3)
    assert np.abs(d - g["dists0_ab"]).max() < 1e-13 and np.abs(fx - g["fx0_ab"]).max() < 1e-13
    a = g["grid_args"]
    grid = O.build_semipolar_grid(a[:2], a[2], a[3], a[4], a[5], int(a[6]))
    assert grid.shape == g["grid"].shape and np.abs(grid - g["grid"]).max() < 1e-13


def test_evenly_spaced_fx_properties():
    # unpinned by the reference (shapely absent): check the defining properties instead
    g = load_golden("area_function")
    x, fx = g["dists0"], g["fx0"]
    xfx = O.evenly_spaced_fx(x, fx, 200)
    assert xfx.shape == (2, 200)
    assert np.allclose(np.diff(xfx[0]), (x[-1] - x[0]) / 199)
    assert xfx[1, 0] == fx[0] and abs(xfx[1, -1] - fx[-1]) < 1e-12


def test_padding_mask():
    g = load_golden("host_collate")
    assert np.array_equal(O.make_padding_mask(np.array([9, 6, 2])), g["mask_9_6_2"])


def test_torch_port_matches_reference_fixture():
    """The CPU-baseline port (stock PyTorch CPU kernels) reproduces the reference's outputs and loss."""
    import torch
    from oracle.torch_port import CpuPort
    g = load_golden("artspeech_c1")
    w, grads = split_wg(g)
    port = CpuPort({k: torch.from_numpy(v) for k, v in w.items()}, int(g["cfg"][1]), int(g["cfg"][3]))
    loss, out = port.step(torch.from_numpy(g["x"]), torch.from_numpy(g["lengths"]).long(), torch.from_numpy(g["targets"]))
    assert np.abs(out.numpy() - g["out"]).max() < 1e-6
    assert abs(loss - float(g["loss"])) < 1e-6
    assert relerr(port.p["linear.0.weight"].grad.numpy(), grads["linear.0.weight"]) < 1e-4
    assert relerr(port.rnn.weight_hh_l0.grad.numpy(), grads["rnn.weight_hh_l0"]) < 1e-4


def test_transformer_oracle_matches_reference_fixture():
    from oracle import transformer_oracle as TO
    g = load_golden("transformer_small")
    w, _ = split_wg(g)
    cfg = tuple(int(v) for v in g["cfg"])
    args = (g["tokens"], g["shifted"], g["src_mask"], g["tgt_mask"], g["src_kpm"], g["tgt_kpm"])
    out = TO.forward(w, cfg, *args, grad_mode=True)            # standard encoder path
    assert out.shape == g["out_grad"].shape and np.abs(out - g["out_grad"]).max() < 2e-6
    out = TO.forward(w, cfg, *args, grad_mode=False)           # nested-tensor fast path: zeros at padded sources
    assert np.abs(out - g["out_nograd"]).max() < 2e-6
    assert np.abs(g["out_grad"] - g["out_nograd"]).max() > 1e-4  # the two modes really differ
    gen = TO.generate(w, cfg, g["tokens"], g["src_kpm"])
    # autoregressive feedback amplifies the fixture's own fp32 rounding frame after frame (fp64 oracle here)
    assert gen.shape == g["gen"].shape and np.abs(gen - g["gen"]).max() < 2e-4


@pytest.mark.parametrize("name", ["deepspeech2_small", "deepspeech2_plain"])
def test_deepspeech2_oracle_matches_reference_fixture(name):
    from oracle import deepspeech2_oracle as DO
    g = load_golden(name)
    w, _ = split_wg(g)
    logits, features = DO.forward(w, g["x"], g.get("voicing"))
    assert logits.shape == g["logits"].shape and np.abs(logits - g["logits"]).max() < 5e-6
    assert np.abs(features - g["features"]).max() < 5e-6
    assert np.array_equal(logits.argmax(-1)[..., None], g["top"])  # topk(k=1) phoneme indices, bit-exact


@pytest.mark.parametrize("name", ["pc_lstm_small", "pc_gru_small"])
def test_principal_components_oracle_matches_reference_fixture(name):
    from oracle import principal_components_oracle as PO
    g = load_golden(name)
    w, _ = split_wg(g)
    out = PO.forward(w, g["tokens"], g["lengths"], lstm=bool(g["cfg"][4]))
    assert out.shape == g["out"].shape and np.abs(out - g["out"]).max() < 2e-6


def test_pc_autoencoder_and_critical_loss_oracle_match_reference_fixtures():
    from oracle import principal_components_oracle as PO
    g = load_golden("pc_autoencoder")
    w, _ = split_wg(g)
    comps = dict(zip(["tongue", "lower-lip", "upper-lip"], (int(c) for c in g["comps"])))
    out, latent = PO.autoencoder_forward(w, g["x"], comps)
    assert np.abs(out - g["out"]).max() < 2e-6 and np.abs(latent - g["latent"]).max() < 2e-6
    c = load_golden("pc_critical_loss")
    loss = PO.critical_loss(c["shapes"], c["ref"], c["mask"], ["TTCD", "LA"], ["lower-lip", "tongue", "upper-lip"])
    assert abs(loss - float(c["loss"])) < 1e-7


def test_semipolar_grid_intersection_oracle_properties():
    """intersect_semipolar_grid restatement (parity unpinned: the reference needs shapely): known answers on a square tube."""
    # grid lines: horizontal segments x in [0, 1] at y = 0.25, 0.5, 2.0; walls: vertical polylines x = 0.3 (internal), x = 0.8 (external)
    grid = np.array([[[x, y] for x in np.linspace(0, 1, 5)] for y in (0.25, 0.5, 2.0)])
    internal = np.array([[0.3, y] for y in np.linspace(0, 1, 6)])
    external = np.array([[0.8, y] for y in np.linspace(0, 0.4, 6)])       # too short for the line at y = 0.5
    flags, pi, pe = O.intersect_semipolar_grid(internal, external, grid)
    assert flags.tolist() == [3, 1, 0]
    assert np.allclose(pi[0], [0.3, 0.25]) and np.allclose(pe[0], [0.8, 0.25])
    # line 1 crosses only the internal wall: the external wall's end nearer to the crossing stands in (its last point)
    assert np.allclose(pi[1], [0.3, 0.5]) and np.array_equal(pe[1], external[-1])
    # a junction hit (y = 0.2 is a vertex of the internal wall) is reported once
    grid2 = np.array([[[x, 0.2] for x in np.linspace(0, 1, 3)]])
    assert len(O._polyline_intersections(grid2[0], internal)) == 1
    # several crossings of a wall: the one closest to the other wall's crossing is chosen
    zig = np.array([[0.2, 0.0], [0.4, 1.0], [0.6, 0.0], [0.7, 1.0]])       # crosses y = 0.5 at x = 0.3, 0.5, 0.65
    line = np.array([[[x, 0.5] for x in np.linspace(0, 1, 4)]])
    ext = np.array([[0.9, 0.0], [0.9, 1.0]])
    flags, pi, pe = O.intersect_semipolar_grid(zig, ext, line)
    assert flags.tolist() == [3] and np.allclose(pi[0], [0.65, 0.5]) and np.allclose(pe[0], [0.9, 0.5])


def test_oracle_reproduces_the_reference_training_and_test_loops():
    """tests/golden/test_loops.npz: what the reference's run_epoch / run_test produced on a captured 6-utterance loader
    (make_golden.py::gen_test_loops).  The oracle, stepped by hand the way run_epoch steps the model (forward, masked mean,
    backward, SGD update, twice), must land on the same losses and the same parameters; its metric functions on the trained
    parameters must give the test loop's info dict."""
    g = load_golden("test_loops")
    V, A, E, H, N = (int(v) for v in g["cfg"])
    n_utt = len(g["lens"])
    w = {k[3:]: v.astype(np.float64) for k, v in g.items() if k.startswith("w0.")}
    w1 = {k[3:]: v for k, v in g.items() if k.startswith("w1.")}

    def collate(idx):  # pad_sequence_collate_fn: sort by length, descending (stable), zero padding
        order = sorted(idx, key=lambda i: -len(g[f"in{i}_tokens"]))
        lens = np.array([len(g[f"in{i}_tokens"]) for i in order])
        T = lens.max()
        x = np.zeros((len(order), T), np.int64)
        tgt = np.zeros((len(order), T, A, 2, N), np.float32)
        for r, i in enumerate(order):
            x[r, :lens[r]], tgt[r, :lens[r]] = g[f"in{i}_tokens"], g[f"in{i}_targets"]
        return x, tgt, lens

    batches = [collate(range(0, 3)), collate(range(3, n_utt))]
    losses = []
    for x, tgt, lens in batches:
        out, cache = O.artspeech_fwd(w, x, lens, A)
        loss, dout = O.masked_euclid_loss(out, tgt, lens)
        grads = O.artspeech_bwd(dout, cache, A)
        losses.append(loss)
        w = {k: w[k] - 0.05 * grads[k] for k in w}
    assert abs(np.mean(losses) - float(g["train_loss"])) < 1e-6
    for k in w1:
        d_ref, d_or = w1[k].astype(np.float64) - g["w0." + k], w[k] - g["w0." + k]
        # the fixture's parameters are fp32: w1 - w0 carries up to an ulp of |w| of rounding on top of the update itself
        assert np.abs(d_or - d_ref).max() <= 1e-3 * np.abs(d_ref).max() + 2.5e-7 * max(1.0, np.abs(w1[k]).max()), k
    # evaluation on the trained parameters (the fixture's own, so that this half does not inherit the first half's error)
    w1d = {k: v.astype(np.float64) for k, v in w1.items()}
    vloss, per_art = [], {n: [[] for _ in range(A)] for n in ("x_corr", "y_corr", "med")}
    for x, tgt, lens in batches:
        out, _ = O.artspeech_fwd(w1d, x, lens, A)
        vloss.append(O.masked_euclid_loss(out, tgt, lens)[0])
        for b, l in enumerate(lens):
            xc, yc = O.pearsons_correlation(out[b:b + 1, :l], tgt[b:b + 1, :l].astype(np.float64))
            med = O.euclidean_distance_metric(out[b:b + 1, :l], tgt[b:b + 1, :l].astype(np.float64)).mean(1)[0]
            for a in range(A):
                per_art["x_corr"][a].append(xc.mean(-1)[0, a])
                per_art["y_corr"][a].append(yc.mean(-1)[0, a])
                per_art["med"][a].append(med[a])
    assert abs(np.mean(vloss) - float(g["valid_loss"])) < 1e-6 and abs(np.mean(vloss) - float(g["test_loss"])) < 1e-6
    names = [str(n) for n in g["test_metric_names"]]
    for a in range(A):
        for n in ("x_corr", "y_corr", "med"):
            assert abs(np.mean(per_art[n][a]) - g["test_metrics"][a, names.index(n)]) < 2e-6, (a, n)


def test_transformer_oracle_reproduces_the_reference_loops_incl_nan_filter():
    """tests/golden/transformer_loops.npz (the reference's run_epoch(VALID) and run_transformer_test at the fixture's
    initial weights): the fp64 oracle, stepped by hand, gives the same VALID loss, marks the same utterance NaN when its
    source is fully masked, and -- with the reference's pairing of KEPT predictions with the UNFILTERED lengths
    (transformer/evaluation.py:96) -- the same test loss and mean Euclidean distances."""
    import torch
    from oracle import transformer_oracle as TO
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn
    from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
    g = load_golden("transformer_loops")
    cfg = tuple(int(v) for v in g["cfg"])
    V, A, d, heads, L, nf = cfg
    torch.manual_seed(33)                       # seeded init of the drop-in class == the reference's (tested contract) ...
    sd = ArtSpeechTransformer(V, A, embed_dim=d, num_heads=heads, num_layers=L, num_feat=nf).state_dict()
    w = {}
    for i, (k, v) in enumerate(sorted(sd.items())):   # ... + make_golden.py::perturb_by_key
        if k != "pos_encoding.pe":
            v = v + (0.02 * torch.cos(0.37 * torch.arange(v.numel(), dtype=torch.float64) + i)).to(v.dtype).view(v.shape)
            v = v * 0.1 if k == "tgt_embedding.1.weight" else v
        w[k] = v.numpy()
    items = []
    for i in range(len(g["lens"])):
        items.append((str(g[f"in{i}_id"]), torch.from_numpy(g[f"in{i}_tokens"]), torch.from_numpy(g[f"in{i}_targets"]),
                      [str(p) for p in g[f"in{i}_phonemes"]], torch.from_numpy(g[f"in{i}_refs"]), torch.tensor([], dtype=torch.int),
                      [str(f) for f in g[f"in{i}_frames"]], torch.from_numpy(g[f"in{i}_voicing"])))
    batches = [pad_sequence_transformer_collate_fn(items[:3]), pad_sequence_transformer_collate_fn(items[3:])]
    losses = []
    for c in batches:   # run_epoch(VALID): eval + no grad -> the encoder's nested-tensor path (zeros at padded sources)
        tokens, targets, lengths = c[1].numpy(), c[2].numpy(), c[3].numpy()
        bs, T = tokens.shape
        shifted = np.concatenate([np.zeros((bs, 1, A, nf)), targets[:, 1:].reshape(bs, T - 1, A, nf)], 1)
        out = TO.forward(w, cfg, tokens, shifted, c[10].numpy(), c[11].numpy(), c[8].numpy(), c[9].numpy(), grad_mode=False)
        losses.append(O.masked_euclid_loss(out, targets, lengths)[0])
    assert abs(np.mean(losses) - float(g["valid0_loss"])) < 2e-6
    # run_transformer_test with the NaN utterance
    nb, nr = int(g["nan_batch"]), int(g["nan_row"])
    losses, med = [], [[] for _ in range(A)]
    for ib, c in enumerate(batches):
        tokens, targets, lengths, kpm = c[1].numpy(), c[2].numpy(), c[3].numpy(), c[8].numpy().copy()
        if ib == nb:
            kpm[nr] = -np.inf
        with np.errstate(invalid="ignore"):
            gen = TO.generate(w, cfg, tokens, kpm)
        nan = np.isnan(gen).reshape(gen.shape[0], -1).any(1)
        assert list(np.nonzero(nan)[0]) == ([nr] if ib == nb else [])
        keep = np.nonzero(~nan)[0]
        gen, tgt = gen[keep], targets[keep]
        losses.append(O.masked_euclid_loss(gen, tgt, lengths[keep])[0])        # the loss: filtered padding mask (:81-86)
        for j in range(len(keep)):                                             # the metrics: unfiltered lengths (:96)
            l = int(lengths[j])
            dist = O.euclidean_distance(gen[j:j + 1, :l], tgt[j:j + 1, :l])    # (1, l, A, N)
            for a in range(A):
                med[a].append(dist[0, :, a].mean(-1).mean())
    assert abs(np.mean(losses) - float(g["test0_loss"])) < 5e-6
    names = [str(n) for n in g["test_metric_names"]]
    want = g["test0_metrics"][:, names.index("med")]
    assert np.abs(np.array([np.mean(m) for m in med]) - want).max() < 5e-6
