"""GPU tests of the harness around the kernels: run_epoch / run_test / the TrainStep engine."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_grad_close

pytestmark = pytest.mark.gpu

ARTS = ["lower-lip", "pharynx", "soft-palate-midline", "tongue", "upper-lip"]  # + upper-incisor injected at test time


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _loaders(n, bs, seed):
    from torch.utils.data import DataLoader
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import SyntheticArtSpeechDataset, pad_sequence_collate_fn
    voc = {"<blank>": 0, "<unk>": 1, **{f"p{i}": i + 2 for i in range(10)}}
    ds = SyntheticArtSpeechDataset(n, voc, ARTS, n_samples=50, min_len=5, max_len=24, seed=seed)
    return voc, DataLoader(ds, batch_size=bs, shuffle=False, collate_fn=pad_sequence_collate_fn)


def test_run_epoch_and_run_test(dev, tmp_path):
    import train_phoneme_to_articulation as tr
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.metrics import P2CPDistance
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.evaluation import run_test
    from artspeech_amd.phoneme_to_articulation.metrics import EuclideanDistance
    from artspeech_amd.settings import DATASET_CONFIG, TRAIN, VALID
    torch.manual_seed(0)
    voc, loader = _loaders(24, 8, seed=0)
    model = ArtSpeech(len(voc), len(ARTS)).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = EuclideanDistance("none")
    first = tr.run_epoch(TRAIN, 1, model, loader, opt, crit, device=dev)["loss"]
    for ep in range(2, 6):
        last = tr.run_epoch(TRAIN, ep, model, loader, opt, crit, device=dev)["loss"]
    assert np.isfinite(last) and last < first            # it learns
    info = tr.run_epoch(VALID, 1, model, loader, opt, crit, fn_metrics={"p2cp_mean": P2CPDistance(DATASET_CONFIG["artspeech2"])},
                        device=dev)
    assert set(info) == {"loss", "p2cp_mean"} and info["p2cp_mean"] > 0
    # generic (unfused) criterion path gives the same loss as the fused one
    class Wrapped(torch.nn.Module):
        def forward(self, o, t):
            return EuclideanDistance("none")(o, t)
    info2 = tr.run_epoch(VALID, 1, model, loader, opt, Wrapped(), device=dev)
    assert abs(info2["loss"] - info["loss"]) < 1e-6
    res = run_test(0, model, loader, crit, str(tmp_path), sorted(ARTS), device=dev)
    assert set(res) == {"loss", *ARTS}
    assert set(res["tongue"]) == {"x_corr", "y_corr", "p2cp", "p2cp_mm", "med", "med_mm"}
    assert abs(res["loss"] - info["loss"]) < 1e-6
    csvs = [f for _, _, fs in os.walk(tmp_path) for f in fs if f == "tract_variables.csv"]
    assert len(csvs) == 24                               # one per sentence (upper incisor injected)
    # run_epoch defers the token-id check for ITS loop only: afterwards nothing is pending (run_test above went through
    # immediate checks, not through a growing list of workspaces) and an id outside the vocabulary raises at the forward again
    assert model.defer_token_check is False and not getattr(model, "_pending_ws", [])
    bad = torch.full((2, 9), len(voc) + 3, dtype=torch.int64, device=dev)
    with pytest.raises(IndexError, match="out of range"):
        model(bad, torch.tensor([9, 9]))


class _CapturedLoader:
    """Fixed list of collated batches + `.dataset.dataset_config`, as the fixture generator fed the reference's loops."""

    def __init__(self, batches, dataset_config):
        import types
        self.batches = batches
        self.dataset = types.SimpleNamespace(dataset_config=dataset_config)

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def test_run_epoch_and_run_test_match_reference_fixture(dev, tmp_path):
    """tests/golden/test_loops.npz holds what the REFERENCE's run_epoch (train_phoneme_to_articulation.py:45-121) and run_test
    (encoder_decoder/evaluation.py:17-161) produced on a captured 6-utterance loader: the TRAIN info + the parameters two SGD
    steps leave, the VALID info with p2cp_mean, the test info dict, one tract_variables.csv, phonemes.csv and contour dumps.
    The drop-in loops must reproduce all of it from the same state_dict and items."""
    import csv
    import train_phoneme_to_articulation as tr
    from conftest import load_golden
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_collate_fn
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.evaluation import run_test
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.metrics import P2CPDistance
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    from artspeech_amd.phoneme_to_articulation.metrics import EuclideanDistance
    from artspeech_amd.settings import DATASET_CONFIG, TRAIN, VALID
    g = load_golden("test_loops")
    V, A, E, H, N = (int(v) for v in g["cfg"])
    arts = [str(a) for a in g["articulators"]]
    items = []
    for i in range(len(g["lens"])):
        items.append((str(g[f"in{i}_id"]), torch.from_numpy(g[f"in{i}_tokens"]), torch.from_numpy(g[f"in{i}_targets"]),
                      [str(p) for p in g[f"in{i}_phonemes"]], torch.from_numpy(g[f"in{i}_refs"]), torch.tensor([], dtype=torch.int),
                      [str(f) for f in g[f"in{i}_frames"]], torch.from_numpy(g[f"in{i}_voicing"])))
    cfg = DATASET_CONFIG["artspeech2"]
    loader = _CapturedLoader([pad_sequence_collate_fn(items[:3]), pad_sequence_collate_fn(items[3:])], cfg)
    model = ArtSpeech(V, A, embed_dim=E, hidden_size=H, n_samples=N)
    w0 = {k[3:]: v for k, v in g.items() if k.startswith("w0.")}
    w1 = {k[3:]: v for k, v in g.items() if k.startswith("w1.")}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()}, strict=True)
    model = model.to(dev)
    crit = EuclideanDistance("none")
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    info = tr.run_epoch(TRAIN, 1, model, loader, opt, crit, device=dev)
    assert set(info) == {"loss"}
    assert abs(info["loss"] - float(g["train_loss"])) < 2e-6, (info["loss"], float(g["train_loss"]))
    for k, v in model.state_dict().items():   # what two optimizer steps changed, element by element
        # both sides are fp32 parameters: each difference carries up to an ulp of |w| of rounding besides the update itself
        assert_grad_close(v.cpu().numpy().astype(np.float64) - w0[k], w1[k].astype(np.float64) - w0[k], f"test_loops: SGD delta of {k}",
                          rtol=1e-3, atol_frac=1e-4, atol_abs=2.5e-7 * max(1.0, float(np.abs(w1[k]).max())))
    vinfo = tr.run_epoch(VALID, 1, model, loader, opt, crit, fn_metrics={"p2cp_mean": P2CPDistance(cfg)}, device=dev)
    assert set(vinfo) == {"loss", "p2cp_mean"}
    assert abs(vinfo["loss"] - float(g["valid_loss"])) < 2e-6
    # the reference's torch.cdist takes the fp32 matmul expansion for 50-point contours: ~1e-3 off the direct formula
    assert abs(vinfo["p2cp_mean"] - float(g["valid_p2cp_mean"])) / float(g["valid_p2cp_mean"]) < 2e-3
    res = run_test(7, model, loader, crit, str(tmp_path), arts, device=dev, regularize_out=False)
    assert list(res) == ["loss"] + arts
    assert abs(res["loss"] - float(g["test_loss"])) < 2e-6
    names = [str(n) for n in g["test_metric_names"]]
    tol = {"x_corr": ("abs", 2e-5), "y_corr": ("abs", 2e-5), "p2cp": ("rel", 2e-3), "p2cp_mm": ("rel", 2e-3), "med": ("rel", 1e-5),
           "med_mm": ("rel", 1e-5)}
    for i, a in enumerate(arts):
        assert list(res[a]) == names
        for j, n in enumerate(names):
            want, got = float(g["test_metrics"][i, j]), res[a][n]
            err = abs(got - want) / (abs(want) if tol[n][0] == "rel" else 1.0)
            assert err < tol[n][1], (a, n, got, want)
    # files of one sentence: same names, same table layout
    sdir = os.path.join(str(tmp_path), "7", "sent4")
    assert len(os.listdir(os.path.join(str(tmp_path), "7"))) == int(g["n_sentence_dirs"])
    assert sorted(os.listdir(os.path.join(sdir, "contours"))) == [str(f) for f in g["contour_files"]]
    with open(os.path.join(sdir, "phonemes.csv")) as f:
        assert [list(r) for r in csv.reader(f)] == [[str(c) for c in r] for r in g["phonemes_csv"]]
    with open(os.path.join(sdir, "tract_variables.csv")) as f:
        rows = list(csv.reader(f))
    cols = rows[0]
    assert cols == [str(c) for c in g["tv_columns"]]
    assert [r[cols.index("frame")] for r in rows[1:]] == [str(v) for v in g["tv_frames"]]
    assert [r[cols.index("phoneme")] for r in rows[1:]] == [str(v) for v in g["tv_phonemes"]]
    num = [str(c) for c in g["tv_numeric_columns"]]
    got = np.array([[float(r[cols.index(c)]) for c in num] for r in rows[1:]])
    want = g["tv_values"]
    # targets are inputs: the same closest pairs, their coordinates bit for bit
    tcols = [j for j, c in enumerate(num) if "_target_poc_" in c]
    assert np.array_equal(got[:, tcols], want[:, tcols])
    # the distance itself: the reference takes it from torch.cdist, which for 50-point sets evaluates |u|^2 + |v|^2 - 2 u.v in
    # fp32 (error ~ 2e-7 / d in d); the HIP kernel takes the direct difference (error ~ 1e-7 * d)
    dcols = [j for j, c in enumerate(num) if c.endswith("_target")]
    assert (np.abs(got[:, dcols] - want[:, dcols]) <= 2e-7 / np.maximum(want[:, dcols], 1e-4) + 1e-7).all()
    vcols = [j for j, c in enumerate(num) if c.endswith("_pred")]
    assert (np.abs(got[:, vcols] - want[:, vcols]) <= 2e-7 / np.maximum(want[:, vcols], 1e-4) + 1e-5).all()
    # predicted closest points: the same points except where two candidate pairs tie within the 1e-6 the contours differ by
    pcols = [j for j, c in enumerate(num) if "_pred_poc_" in c]
    same = np.abs(got[:, pcols] - want[:, pcols]) < 1e-5
    assert same.mean() > 0.97, same.mean()
    frame0 = str(g["in4_frames"][0])
    assert np.abs(np.load(os.path.join(sdir, "contours", f"{frame0}_tongue.npy")) - g["pred_tongue_frame0"]).max() < 1e-5
    assert np.array_equal(np.load(os.path.join(sdir, "contours", f"{frame0}_upper-incisor_true.npy")), g["true_incisor_frame0"])


@pytest.mark.parametrize("n_samp", [50, 1, 70])
def test_train_step_engine_equals_module_path(dev, n_samp):
    """n_samp = 1 and 70: 2 and 140 outputs per head, which the output-layer kernel with the fused criterion does not take (4 ..
    128): the library then runs the criterion as its own kernel and the engine steps as before (it used to fail with
    AS_ERR_UNSUPPORTED for these)."""
    from artspeech_amd.engine import TrainStep
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    from artspeech_amd.phoneme_to_articulation.metrics import masked_euclidean_loss
    torch.manual_seed(1)
    B, T, A = 6, 30, 3
    model = ArtSpeech(20, A, n_samples=n_samp).to(dev)
    lengths = torch.tensor([30, 28, 20, 11, 4, 1], dtype=torch.int32)
    x = torch.randint(1, 20, (B, T), device=dev)
    tgt = torch.rand(B, T, A, 2, n_samp, device=dev)
    loss = masked_euclidean_loss(model(x, lengths), tgt, lengths)
    loss.backward()
    ref_grad, ref_loss = model.flat.grad.clone(), loss.item()
    ref_param = model.flat.data.clone()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-6)
    opt.step()
    after_torch = model.flat.data.clone()
    model.flat.data.copy_(ref_param)
    step = TrainStep(model, B, T, lr=1e-3, weight_decay=1e-6)
    scale = 1.0 / (int(lengths.sum()) * A * n_samp)
    step.step(x, lengths.to(dev), tgt, scale)
    torch.cuda.synchronize()
    assert abs(step.loss.item() - ref_loss) < 1e-7
    if n_samp == 50:
        assert torch.equal(step.grads, ref_grad)          # same kernels, same order: bit-identical
    else:   # (the module path's criterion is a separate autograd node with its own rounding of the sigmoid's backward)
        assert torch.allclose(step.grads, ref_grad, rtol=1e-4, atol=1e-6 * float(ref_grad.abs().max()))
    assert torch.allclose(model.flat.data, after_torch, rtol=1e-5, atol=1e-7)  # fused Adam == torch.optim.Adam


def test_two_models_interleaved_on_two_streams(dev):
    """The library's side stream, fork/join events and heads-done event belong to the (device, caller stream) pair: two
    models stepped alternately on two streams (no host sync in between) leave exactly the losses and gradients that each
    leaves when it runs alone."""
    from artspeech_amd.engine import TrainStep
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    B, T, A = 8, 48, 3
    lengths = torch.tensor([48, 45, 40, 33, 21, 12, 5, 1], dtype=torch.int32)
    cases = []
    for seed in (11, 12):
        torch.manual_seed(seed)
        model = ArtSpeech(20, A).to(dev)
        x = torch.randint(1, 20, (B, T), device=dev)
        tgt = torch.rand(B, T, A, 2, 50, device=dev)
        cases.append((model, x, tgt, TrainStep(model, B, T, optimizer=False)))
    scale = 1.0 / (int(lengths.sum()) * A * 50)
    ldev = lengths.to(dev)
    alone = []
    for model, x, tgt, step in cases:            # reference: each alone on the default stream
        step.forward_backward(x, ldev, tgt, scale)
        torch.cuda.synchronize()
        alone.append((step.loss.clone(), step.grads.clone()))
        step.grads.zero_()
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    torch.cuda.synchronize()
    for _ in range(3):                            # interleaved, no synchronisation between the launches
        for (model, x, tgt, step), st in zip(cases, streams):
            with torch.cuda.stream(st):
                step.forward_backward(x, ldev, tgt, scale)
    torch.cuda.synchronize()
    for (model, x, tgt, step), (loss, grads) in zip(cases, alone):
        assert torch.equal(step.loss, loss)
        assert torch.equal(step.grads, grads)


def test_gru_dropout_training_mode(dev):
    """nn.GRU(dropout=p) inter-layer dropout: mask statistics, and exact forward/backward parity with the
    oracle GIVEN the mask the library generated (the mask is a pure function of the seed)."""
    from artspeech_amd import _lib
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    from artspeech_amd.phoneme_to_articulation.metrics import masked_euclidean_loss
    from oracle import artspeech_oracle as O
    L = _lib.lib()
    p = 0.3
    ones = torch.ones(1 << 20, device=dev)
    m = torch.empty_like(ones)
    _lib.check(L.as_dropout_fwd(_lib.ptr(ones), _lib.ptr(m), ones.numel(), p, 1234, _lib.stream_ptr()))
    keep = (m > 0).float().mean().item()
    assert abs(keep - (1 - p)) < 3e-3 and torch.allclose(m[m > 0], torch.tensor(1 / (1 - p), device=dev))
    m2 = torch.empty_like(ones)
    _lib.check(L.as_dropout_fwd(_lib.ptr(ones), _lib.ptr(m2), ones.numel(), p, 1235, _lib.stream_ptr()))
    assert not torch.equal(m, m2)                       # another seed, another mask
    # model in training mode
    torch.manual_seed(5)
    B, T, A, H = 4, 16, 2, 128
    model = ArtSpeech(20, A, dropout=p).to(dev)
    sd = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
    lengths = np.array([16, 11, 7, 2])
    rng = np.random.RandomState(0)
    x = rng.randint(1, 20, (B, T))
    tgt = rng.rand(B, T, A, 2, 50).astype(np.float32)
    for b, l in enumerate(lengths):
        x[b, l:] = 0
        tgt[b, l:] = 0
    model.train()
    torch.manual_seed(77)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())   # what forward() will draw
    torch.manual_seed(77)
    out = model(torch.from_numpy(x).to(dev), torch.from_numpy(lengths))
    loss = masked_euclidean_loss(out, torch.from_numpy(tgt).to(dev), lengths)
    loss.backward()
    scale = torch.empty(B * T * 2 * H, device=dev)
    one = torch.ones_like(scale)
    _lib.check(L.as_dropout_fwd(_lib.ptr(one), _lib.ptr(scale), scale.numel(), p, seed, _lib.stream_ptr()))
    scale = scale.view(B, T, 2 * H).cpu().numpy()
    o_out, cache = O.artspeech_fwd(sd, x, lengths, A, interlayer_scale=scale)
    assert np.abs(out.detach().cpu().numpy() - o_out).max() < 1e-5
    o_loss, o_dout = O.masked_euclid_loss(o_out, tgt, lengths)
    og = O.artspeech_bwd(o_dout, cache, A)
    for k, v in model.named_grad_views().items():
        assert_grad_close(v.cpu().numpy(), og[k], f"gru dropout vs oracle: {k}")
    # eval mode ignores dropout
    model.eval()
    with torch.no_grad():
        out_eval = model(torch.from_numpy(x).to(dev), torch.from_numpy(lengths))
    o_eval, _ = O.artspeech_fwd(sd, x, lengths, A)
    assert np.abs(out_eval.cpu().numpy() - o_eval).max() < 1e-5


def test_simple_artspeech_dropout_training_mode(dev):
    """SimpleArtSpeech(dropout=p).train(): nn.Dropout on the embedded frames (reference models.py:64,85).  Exact forward /
    backward parity with the oracle GIVEN the mask the library generated; eval mode ignores p."""
    from artspeech_amd import _lib
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import SimpleArtSpeech
    from artspeech_amd.phoneme_to_articulation.metrics import masked_euclidean_loss
    from oracle import artspeech_oracle as O
    L = _lib.lib()
    p, B, T, A, E = 0.25, 3, 21, 2, 64
    torch.manual_seed(3)
    model = SimpleArtSpeech(20, A, dropout=p).to(dev)
    sd = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
    rng = np.random.RandomState(1)
    x = rng.randint(0, 20, (B, T))
    tgt = rng.rand(B, T, A, 2, 50).astype(np.float32)
    lengths = np.array([T, T, T])
    model.train()
    torch.manual_seed(99)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())   # what forward() will draw
    torch.manual_seed(99)
    out = model(torch.from_numpy(x).to(dev), torch.from_numpy(lengths))
    loss = masked_euclidean_loss(out, torch.from_numpy(tgt).to(dev), lengths)
    loss.backward()
    scale = torch.empty(B * T * E, device=dev)
    _lib.check(L.as_dropout_fwd(_lib.ptr(torch.ones_like(scale)), _lib.ptr(scale), scale.numel(), p, seed, _lib.stream_ptr()))
    scale = scale.view(B, T, E).cpu().numpy()
    assert 0.6 < (scale > 0).mean() < 0.9
    o_out, cache = O.simple_artspeech_fwd(sd, x, A, embed_scale=scale)
    err = np.abs(out.detach().cpu().numpy() - o_out).max()
    assert err < 1e-5, err
    o_loss, o_dout = O.masked_euclid_loss(o_out, tgt, lengths)
    assert abs(loss.item() - o_loss) < 1e-6
    og = O.simple_artspeech_bwd(o_dout, cache, A)
    for k, v in model.named_grad_views().items():
        a, b = v.cpu().numpy().astype(np.float64), og[k]
        viol = np.abs(a - b) - (1e-4 * np.abs(b) + 1e-6 * np.abs(b).max() + 1e-9)
        assert viol.max() <= 0, (k, float(np.abs(a - b).max()), float(np.abs(b).max()))
    # another draw, another mask; eval mode is deterministic and ignores p
    out2 = model(torch.from_numpy(x).to(dev), torch.from_numpy(lengths))
    assert not torch.equal(out2, out)
    model.eval()
    with torch.no_grad():
        out_eval = model(torch.from_numpy(x).to(dev), torch.from_numpy(lengths))
    o_eval, _ = O.simple_artspeech_fwd(sd, x, A)
    assert np.abs(out_eval.cpu().numpy() - o_eval).max() < 1e-5
    # dropout follows the MODULE's mode, not grad mode (nn.Dropout drops under torch.no_grad() while model.training is set)
    model.train()
    torch.manual_seed(99)
    with torch.no_grad():
        out_ng = model(torch.from_numpy(x).to(dev), torch.from_numpy(lengths))
    assert torch.equal(out_ng, out.detach())          # same seed, same mask as the graded call above


def test_evaluation_entry_points_write_reference_outputs(dev, tmp_path):
    """test_phoneme_to_articulation*.py: checkpoint -> test split -> test_results.{json,csv} + per-sentence outputs."""
    import json
    import sys
    import yaml
    sys.path.insert(0, ROOT)
    import test_phoneme_to_articulation as cli
    import test_phoneme_to_articulation_transformer as cli_t
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    with open(os.path.join(ROOT, "configs", "test_synthetic.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg.update(test_seq_dict={"num_sentences": 6}, batch_size=3, save_to=str(tmp_path / "gru"), synthetic={"min_len": 5, "max_len": 12})
    torch.manual_seed(0)
    ckpt = str(tmp_path / "best_model.pt")
    torch.save(ArtSpeech(45, len(cfg["articulators"])).state_dict(), ckpt)  # reference-keyed checkpoint
    cfg["state_dict_fpath"] = ckpt
    res = cli.main(**cfg)
    with open(tmp_path / "gru" / "test_results.json") as f:
        assert json.load(f)["loss"] == pytest.approx(res["loss"])
    with open(tmp_path / "gru" / "test_results.csv") as f:
        header, row = f.read().strip().split("\n")
    arts = sorted(cfg["articulators"])
    assert header.split(",")[:3] == ["exp", "loss", f"p2cp_{arts[0]}"] and len(header.split(",")) == 2 + 4 * len(arts)
    assert float(row.split(",")[1]) == pytest.approx(res["loss"])
    sentence_dirs = os.listdir(tmp_path / "gru" / "test_outputs" / "0")
    assert len(sentence_dirs) == 6
    first = tmp_path / "gru" / "test_outputs" / "0" / sentence_dirs[0]
    assert os.path.exists(first / "phonemes.csv") and os.path.exists(first / "tract_variables.csv") and os.listdir(first / "contours")

    with open(os.path.join(ROOT, "configs", "test_transformer_synthetic.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg.update(test_seq_dict={"num_sentences": 2}, batch_size=2, save_to=str(tmp_path / "tf"), synthetic={"min_len": 4, "max_len": 6},
               model_kwargs={"embed_dim": 32, "num_heads": 4, "num_layers": 1, "num_feat": 100})
    res = cli_t.main(**cfg)
    assert np.isfinite(res["loss"]) and os.path.exists(tmp_path / "tf" / "test_results.csv")


def test_overlapped_gradient_all_reduce_equals_single_all_reduce(dev):
    """Two ranks (gloo, sharing the GPU): the two-piece all-reduce that overlaps the GRU backward leaves bit-identical
    parameters and keeps the ranks in sync (tools/check_dp_overlap.py exits non-zero otherwise)."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, ARTSPEECH_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "check_dp_overlap.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert res.stdout.count("overlap == plain all-reduce: True") == 2
    assert res.stdout.count("pipelined == plain: True") == 2     # the pipelined engine's late all-reduce under data parallelism
    # transformer trainer: all-reduce hooks fired during the backward == full-batch gradients
    cmd[-1] = os.path.join(ROOT, "tools", "check_dp_transformer.py")
    cmd[cmd.index("--master-port") + 1] = str(port + 1 if port < 65000 else port - 1)
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0 and res.stdout.count("ok = True") == 2, res.stdout[-2000:] + res.stderr[-2000:]


@pytest.mark.parametrize("bad", ["V", "negative"])
def test_out_of_range_token_ids_raise_like_nn_embedding(dev, bad):
    """nn.Embedding raises IndexError for an id outside [0, V) (reference models.py:135).  Here the kernels clamp such ids
    into their tables (the layer-0 backward keeps per-token sums in LDS: an id >= V must never address memory outside them)
    and as_artspeech_fwd counts them for the host: the drop-in forward raises at once, the deferred form and the training
    engine raise where they synchronise, and a whole step with a bad id runs to completion without touching other memory."""
    from artspeech_amd import engine
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech, SimpleArtSpeech
    from artspeech_amd.phoneme_to_articulation.metrics import masked_euclidean_loss
    torch.manual_seed(0)
    V, A, B, T = 45, 3, 4, 64          # V <= T: the token-sum variant of the layer-0 backward recurrence
    lengths = torch.tensor([64, 50, 33, 7])
    x = torch.randint(1, V, (B, T))
    tgt = torch.rand(B, T, A, 2, 50)
    for b, l in enumerate(lengths):
        x[b, l:] = 0
        tgt[b, l:] = 0
    x_bad = x.clone()
    x_bad[1, 5] = V if bad == "V" else -3
    x_bad[0, 63] = V + 100000 if bad == "V" else -(1 << 40)
    for cls in (ArtSpeech, SimpleArtSpeech):
        model = cls(V, A).to(dev)
        out = model(x.to(dev), lengths)          # valid ids pass
        assert torch.isfinite(out).all()
        with pytest.raises(IndexError, match="out of range"):
            model(x_bad.to(dev), lengths)
        # deferred: forward + loss + backward run (clamped ids, no fault), the check raises afterwards
        model.defer_token_check = True
        out = model(x_bad.to(dev), lengths)
        loss = masked_euclidean_loss(out, tgt.to(dev), lengths.numpy())
        loss.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(model.flat.grad).all()
        with pytest.raises(IndexError, match="2 token ids outside"):
            model.check_tokens()
        model.check_tokens()                      # the pending list is consumed
    # training engine: the count is read with the loss
    model = ArtSpeech(V, A).to(dev)
    step = engine.TrainStep(model, B, T)
    ld = lengths.to(torch.int32).to(dev)
    scale = 1.0 / (float(lengths.sum()) * A * 50)
    step.step(x.to(dev), ld, tgt.to(dev), scale)
    assert np.isfinite(step.loss_value())
    step.step(x_bad.to(dev), ld, tgt.to(dev), scale)
    with pytest.raises(IndexError, match="out of range"):
        step.loss_value()
    step.step(x.to(dev), ld, tgt.to(dev), scale)   # the count is per batch, not sticky
    assert np.isfinite(step.loss_value())


@pytest.mark.parametrize("shape", ["small", "configs1"])
def test_pipelined_engine_is_bit_identical_to_the_unpipelined_one(dev, shape):
    """TrainStep(pipeline=True) carries the weight gradient + Adam update of the heads' second Linear into the next step's
    forward (beside its recurrences).  After flush() the parameters, both Adam moments and every step's loss must equal the
    unpipelined engine's BIT FOR BIT: same operands, same arithmetic, another schedule.  B*T is a multiple of 32 and >= 512 so
    that the fused weight-gradient launches are the ones exercised; ragged lengths; batches change from step to step."""
    from artspeech_amd.engine import TrainStep
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    if shape == "small":
        V, A, B, T = 45, 3, 8, 96
        lengths = torch.tensor([96, 90, 77, 64, 40, 33, 8, 1], dtype=torch.int32)
    else:       # BASELINE configs[1]: B = 32, T = 200, A = 11, ragged
        V, A, B, T = 45, 11, 32, 200
        lengths = torch.linspace(200, 60, 32).int()
    scale = 1.0 / (float(lengths.sum()) * A * 50)
    g = torch.Generator().manual_seed(3)
    batches = []
    for _ in range(5):
        x = torch.randint(1, V, (B, T), generator=g)
        tgt = torch.rand(B, T, A, 2, 50, generator=g)
        for b, l in enumerate(lengths):
            x[b, l:] = 0
            tgt[b, l:] = 0
        batches.append((x.to(dev), tgt.to(dev)))
    ld = lengths.to(dev)
    results = []
    for pipeline in (False, True):
        torch.manual_seed(11)
        model = ArtSpeech(V, A).to(dev)
        step = TrainStep(model, B, T, lr=1e-3, weight_decay=1e-6, pipeline=pipeline)
        assert step.pipeline == pipeline
        losses = []
        for x, tgt in batches:
            step.step(x, ld, tgt, scale)
            losses.append(step.loss.clone())
        if pipeline:
            assert step.pending is not None
            # before the flush the late slice still holds the previous step's parameters
            assert not torch.equal(model.flat.data[step.late_off:], results[0][0][step.late_off:])
            assert torch.equal(model.flat.data[:step.late_off], results[0][0][:step.late_off])
        step.flush()
        torch.cuda.synchronize()
        assert step.pending is None
        results.append((model.flat.data.clone(), step.exp_avg.clone(), step.exp_avg_sq.clone(), torch.stack(losses)))
    for a, b, what in zip(results[0], results[1], ("parameters", "exp_avg", "exp_avg_sq", "losses")):
        assert torch.equal(a, b), f"pipelined {what} differ: max |diff| {(a - b).abs().max().item():.3e}"
    assert torch.isfinite(results[0][3]).all() and results[0][3][-1] < results[0][3][0]


@pytest.mark.parametrize("seed", range(int(os.environ.get("AS_FUZZ_SEEDS", "10"))))
def test_engine_random_configurations(dev, seed):
    """The training engine (one C call per step: fused criterion, fused weight-gradient launches, side streams, optional
    pipelining across the step boundary) on seeded random architectures and batch shapes -- vocabulary beyond and within the
    in-kernel token table, hidden sizes of both recurrence families, 1-12 articulators, odd contour sizes, B * T on both
    sides of the fused launches' thresholds: (1) loss and gradients of the first step equal the module path's (autograd over the
    drop-in modules, itself swept against the oracle in test_gpu_parity.py); (2) three steps over changing batches leave the
    pipelined and the unpipelined engine with bit-identical parameters, Adam moments and losses."""
    from artspeech_amd.engine import TrainStep
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    from artspeech_amd.phoneme_to_articulation.metrics import masked_euclidean_loss
    r = np.random.RandomState(8000 + seed)
    V, A = int(r.choice([5, 45, 100, 200])), int(r.randint(1, 13))
    E, H, N = int(r.choice([8, 24, 64])), int(r.choice([32, 64, 128, 128, 44])), int(r.choice([3, 25, 50, 50, 64]))
    B, T = int(r.randint(1, 13)), int(r.choice([7, 32, 50, 96, 130]))
    lengths = torch.from_numpy(np.sort(r.randint(1, T + 1, B))[::-1].copy()).int()
    lengths[0] = T
    scale = 1.0 / (float(lengths.sum()) * A * N)
    g = torch.Generator().manual_seed(seed)
    batches = []
    for _ in range(3):
        x = torch.randint(1, V, (B, T), generator=g)
        tgt = torch.rand(B, T, A, 2, N, generator=g)
        for b, l in enumerate(lengths):
            x[b, l:] = 0
            tgt[b, l:] = 0
        batches.append((x.to(dev), tgt.to(dev)))
    ld = lengths.to(dev)
    what = dict(V=V, A=A, E=E, H=H, N=N, B=B, T=T, lengths=lengths.tolist())

    def make():
        torch.manual_seed(seed)
        return ArtSpeech(V, A, embed_dim=E, hidden_size=H, n_samples=N).to(dev)

    model = make()
    loss = masked_euclidean_loss(model(batches[0][0], lengths), batches[0][1], lengths)
    loss.backward()
    ref_grad, ref_loss = model.flat.grad.clone(), loss.item()
    model = make()
    step = TrainStep(model, B, T, optimizer=False)
    step.forward_backward(batches[0][0], ld, batches[0][1], scale)
    torch.cuda.synchronize()
    assert abs(step.loss.item() - ref_loss) < 1e-6, what
    gmax = ref_grad.abs().max().item()
    assert (step.grads - ref_grad).abs().max().item() <= 2e-6 * gmax, (what, (step.grads - ref_grad).abs().max().item() / gmax)
    results = []
    for pipeline in (False, True):
        model = make()
        step = TrainStep(model, B, T, lr=1e-3, weight_decay=1e-6, pipeline=pipeline)
        losses = []
        for x, tgt in batches:
            step.step(x, ld, tgt, scale)
            losses.append(step.loss.clone())
        step.flush()
        torch.cuda.synchronize()
        results.append((model.flat.data.clone(), step.exp_avg.clone(), step.exp_avg_sq.clone(), torch.stack(losses)))
    for a, b, name in zip(results[0], results[1], ("parameters", "exp_avg", "exp_avg_sq", "losses")):
        assert torch.equal(a, b), (what, f"pipelined {name} differ: max |diff| {(a - b).abs().max().item():.3e}")
    assert torch.isfinite(results[0][3]).all()


@pytest.mark.parametrize("G", [2, 4, 8])
def test_shards_of_the_benchmark_batch_sum_to_the_full_batch(dev, G):
    """BASELINE configs[2] on one device: the length-sorted batch of 32 ragged utterances dealt round-robin over G ranks
    (distributed.shard_batch), each rank's engine step with the GLOBAL valid-frame count in its loss scale; the G losses and
    the G flat gradient buffers must SUM to the full batch's (what the single all-reduce of SURVEY 8e delivers), for the rank
    batch sizes 16, 8 and 4 of the 2-, 4- and 8-GPU runs."""
    from artspeech_amd import distributed as dp
    from artspeech_amd.engine import TrainStep
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    V, A, B, T, N = 45, 11, 32, 200, 50
    lengths = torch.linspace(200, 60, B).int()
    gen = torch.Generator().manual_seed(5)
    x = torch.randint(1, V, (B, T), generator=gen)
    tgt = torch.rand(B, T, A, 2, N, generator=gen)
    for b, l in enumerate(lengths):
        x[b, l:] = 0
        tgt[b, l:] = 0
    torch.manual_seed(2)
    model = ArtSpeech(V, A).to(dev)
    full = TrainStep(model, B, T, optimizer=False)
    full.forward_backward(x.to(dev), lengths.to(dev), tgt.to(dev), dp.loss_scale(int(lengths.sum()), A, N))
    torch.cuda.synchronize()
    full_loss, full_grad = full.loss.item(), full.grads.clone()
    loss_sum, grad_sum = 0.0, torch.zeros_like(full_grad)
    for r in range(G):
        xs, ts, ls, n_valid = dp.shard_batch(x, tgt, lengths, r, G)
        assert n_valid == int(lengths.sum()) and xs.shape[0] == B // G and xs.shape[1] == int(ls[0])
        shard = TrainStep(model, xs.shape[0], xs.shape[1], optimizer=False)
        shard.forward_backward(xs.contiguous().to(dev), ls.to(dev), ts.contiguous().to(dev), dp.loss_scale(n_valid, A, N))
        torch.cuda.synchronize()
        loss_sum += shard.loss.item()
        grad_sum += shard.grads
    assert abs(loss_sum - full_loss) < 1e-6
    err = (grad_sum - full_grad).abs().max().item() / full_grad.abs().max().item()
    assert err < 1e-5, err


def test_hbm_resident_dataset_collates_like_the_host_collate(dev):
    """HBMResidentDataset.collate (device-side gather / pad, as_gather_pad_rows) returns the tuple of pad_sequence_collate_fn:
    same order, dtypes, padding values (0 / -1), with the tensor fields on the device; run_epoch gives the same loss over it."""
    import train_phoneme_to_articulation as tr
    from torch.utils.data import DataLoader
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import (
        HBMResidentDataset, SyntheticArtSpeechDataset, pad_sequence_collate_fn)
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    from artspeech_amd.phoneme_to_articulation.metrics import EuclideanDistance
    from artspeech_amd.settings import VALID
    voc = {"<blank>": 0, "<unk>": 1, **{f"p{i}": i + 2 for i in range(10)}}
    ds = SyntheticArtSpeechDataset(21, voc, ARTS, n_samples=50, min_len=1, max_len=33, seed=4, voiced_tokens=["p1", "p4"])
    rds = HBMResidentDataset(ds, dev)
    assert len(rds) == len(ds) and rds.dataset_config is ds.dataset_config
    r = np.random.RandomState(0)   # + 25 seeded index lists: any size, any order, repeats allowed (a sampler with replacement)
    drawn = [r.randint(0, 21, int(r.randint(1, 22))).tolist() for _ in range(25)]
    for idx in [[3, 0, 7, 12, 20], [5], list(range(21))] + drawn:
        want = pad_sequence_collate_fn([ds[i] for i in idx])
        got = rds.collate(idx)
        assert got[0] == want[0] and got[4] == want[4] and got[6] == want[6]
        assert torch.equal(got[3], want[3]) and got[3].dtype == want[3].dtype and not got[3].is_cuda
        for f in (1, 2, 5, 7):
            assert got[f].is_cuda and got[f].dtype == want[f].dtype and got[f].shape == want[f].shape
            assert torch.equal(got[f].cpu(), want[f]), f
    torch.manual_seed(0)
    model = ArtSpeech(len(voc), len(ARTS)).to(dev)
    crit = EuclideanDistance("none")
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    a = tr.run_epoch(VALID, 1, model, DataLoader(ds, batch_size=8, shuffle=False, collate_fn=pad_sequence_collate_fn), opt, crit, device=dev)
    b = tr.run_epoch(VALID, 1, model, DataLoader(rds, batch_size=8, shuffle=False, collate_fn=rds.collate), opt, crit, device=dev)
    assert a["loss"] == b["loss"]


def test_criterion_fused_into_the_output_layer_equals_the_separate_kernel(dev):
    """TrainStep asks as_artspeech_fwd to fuse the masked Euclidean criterion (and its gradient through the sigmoid) into the
    epilogue of the heads' output layer (as_opts.loss_*).  Against the same engine with the separate criterion kernel:
    contours and every gradient bit for bit (same arithmetic per element), the loss to summation order (1e-7)."""
    from artspeech_amd.engine import TrainStep
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    V, A, B, T = 45, 4, 8, 100                      # 800 frames: 64-row tiles + a ragged 32-row tail per head
    lengths = torch.tensor([100, 97, 64, 63, 40, 33, 8, 1], dtype=torch.int32)
    g = torch.Generator().manual_seed(5)
    x = torch.randint(1, V, (B, T), generator=g)
    tgt = torch.rand(B, T + 3, A, 2, 50, generator=g)       # targets padded beyond T: tgt_T > T
    for b, l in enumerate(lengths):
        x[b, l:] = 0
    x, tgt, ld = x.to(dev), tgt.to(dev), lengths.to(dev)
    scale = 1.0 / (float(lengths.sum()) * A * 50)
    res = []
    for fused in (False, True):
        torch.manual_seed(2)
        model = ArtSpeech(V, A).to(dev)
        step = TrainStep(model, B, T, optimizer=False)
        assert step.fuse_loss
        step.fuse_loss = fused
        step.forward_backward(x, ld, tgt, scale)
        torch.cuda.synchronize()
        res.append((step.out.clone(), step.dout.clone(), step.grads.clone(), float(step.loss)))
    assert torch.equal(res[0][0], res[1][0]), "contours differ"
    assert torch.equal(res[0][1], res[1][1]), "d loss / d(pre-sigmoid) differs"
    assert torch.equal(res[0][2], res[1][2]), "gradients differ"
    assert abs(res[0][3] - res[1][3]) < 1e-7 and np.isfinite(res[0][3])


def test_rccl_single_rank_engine_is_bit_identical(dev):
    """The engine's data-parallel step over a ONE-rank RCCL group (child process, tests/_rccl_single_rank_worker.py): the
    overlapped two-piece all-reduce on the communication stream and the late slice's all-reduce run through the real backend
    (a one-GPU box cannot hold two RCCL ranks; the world-2 arithmetic is covered by the gloo test in tests/test_host.py)."""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_single_rank_worker.py")
    r = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "rccl single rank ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])

