"""CPU tests of the host side: C-ABI surface, flat-parameter modules (state_dict contract, seed-for-seed
initialisation), collate functions, loud failure without a GPU, and the data-parallel scheme rehearsed
with gloo (world_size 2) using the CPU oracle as the per-rank compute."""
import os
import re
import socket

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, split_wg


def test_header_symbols_are_exported_and_bound():
    from artspeech_amd import _lib
    header = open(os.path.join(ROOT, "include", "artspeech_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(as_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    L = _lib.lib()  # loads without a GPU; resolves every prototype
    for name in declared:
        assert hasattr(L, name), f"{name} declared in the header but not exported by the library"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert L.as_arch() == b"gfx950"


def test_product_library_has_no_ablation_switches():
    """The shipped library reads no AS_* environment variable: ablations, tuning aids and legacy kernels exist only in the
    -DAS_DIAG flavour (libartspeech_hip_diag.so, `python -m artspeech_amd.build --diag`) that tools/ load on request."""
    from artspeech_amd import _lib
    assert _lib.LIB_PATH.endswith("libartspeech_hip.so")
    blob = open(_lib.LIB_PATH, "rb").read()
    names = set(re.findall(rb"AS_[A-Z][A-Z0-9_]{2,}", blob))
    switches = {n for n in names if not n.startswith((b"AS_ERR_", b"AS_HEAD_", b"AS_WAVE"))}
    assert not switches, f"environment switches compiled into the product library: {sorted(switches)}"
    csrc = os.path.join(ROOT, "artspeech_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".h")):
            for line in open(os.path.join(csrc, f)):
                if "getenv(" in line and "#define AS_DIAG_" not in line:
                    assert "ARTSPEECH_" in line, f"{f}: raw getenv outside the AS_DIAG macros: {line.strip()}"


def test_layout_is_disjoint_and_counts_parameters():
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech, SimpleArtSpeech
    for cls, kw in ((ArtSpeech, {}), (SimpleArtSpeech, {})):
        m = cls(45, 11, **kw)
        spans = sorted((off, off + int(np.prod(shape))) for off, shape in m._views.values())
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
        assert spans[-1][1] <= m.flat.numel()
    assert ArtSpeech(45, 11).total_parameters == 1864972   # SURVEY 2.3 (reference instantiation)
    assert ArtSpeech(45, 2).total_parameters == 732808
    assert ArtSpeech(45, 10).total_parameters == 1739176


def test_state_dict_contract_and_seed_for_seed_init():
    """Same seed => the very same initial weights as the reference (fixture artspeech_c1 was created by
    torch.manual_seed(0); ArtSpeech(45, 2) in the reference) under the reference's key names."""
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    g = load_golden("artspeech_c1")
    w, _ = split_wg(g)
    torch.manual_seed(0)
    model = ArtSpeech(45, 2)
    sd = model.state_dict()
    assert set(sd) == set(w)
    for k, v in w.items():
        assert tuple(sd[k].shape) == v.shape, k
        assert np.array_equal(sd[k].numpy(), v), k
    # round trip through load_state_dict
    other = ArtSpeech(45, 2)
    other.load_state_dict(sd, strict=True)
    assert torch.equal(other.flat, model.flat)
    # strict errors like nn.Module
    bad = dict(sd)
    bad.pop("embedding.weight")
    bad["extra.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="Missing key|Unexpected key"):
        other.load_state_dict(bad, strict=True)
    bad = dict(sd)
    bad["linear.0.bias"] = torch.zeros(3)
    with pytest.raises(RuntimeError, match="size mismatch"):
        other.load_state_dict(bad)


def test_simple_model_state_dict_keys():
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import SimpleArtSpeech
    g = load_golden("simple_small")
    w, _ = split_wg(g)
    V, A, E, H, N = (int(v) for v in g["cfg"])
    m = SimpleArtSpeech(V, A, embed_dim=E, hidden_size=H, num_samples=N)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: v.shape for k, v in w.items()}


def test_cpu_tensors_fail_loudly():
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    from artspeech_amd.phoneme_to_articulation.metrics import EuclideanDistance
    m = ArtSpeech(9, 1, embed_dim=16, hidden_size=32, n_samples=5)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(2, 3, dtype=torch.long), torch.tensor([3, 2]))
    with pytest.raises(RuntimeError, match="no CPU path"):
        EuclideanDistance("none")(torch.rand(1, 2, 1, 2, 5), torch.rand(1, 2, 1, 2, 5))


def test_product_code_never_imports_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "artspeech_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                assert "oracle" not in src, f"{f} mentions the oracle"
    assert "oracle" not in open(os.path.join(ROOT, "train_phoneme_to_articulation.py")).read()


def test_collate_functions_match_reference_fixture():
    from artspeech_amd.helpers import make_padding_mask
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import (
        pad_sequence_collate_fn, pad_sequence_transformer_collate_fn)
    g = load_golden("host_collate")
    batch = []
    for i in range(4):
        l = len(g[f"in{i}_tokens"])
        batch.append((f"s{i}", torch.from_numpy(g[f"in{i}_tokens"]), torch.from_numpy(g[f"in{i}_targets"]),
                      [f"p{i}_{j}" for j in range(l)], torch.from_numpy(g[f"in{i}_refs"]), torch.tensor([], dtype=torch.int),
                      list(range(100 * i, 100 * i + l)), torch.from_numpy(g[f"in{i}_voicing"])))
    c8 = pad_sequence_collate_fn(batch)
    assert len(c8) == 8
    assert list(c8[0]) == [str(s) for s in g["ids"]]
    assert torch.equal(c8[1], torch.from_numpy(g["tokens"])) and c8[1].dtype == torch.int64
    assert torch.equal(c8[2], torch.from_numpy(g["targets"]))
    assert torch.equal(c8[3], torch.from_numpy(g["lengths"])) and c8[3].dtype == torch.int32
    assert c8[4][0] == [str(s) for s in g["phonemes0"]]
    assert torch.equal(c8[5], torch.from_numpy(g["refs"]))
    assert c8[6][0] == list(g["frames0"])
    assert torch.equal(c8[7], torch.from_numpy(g["voicing"]))
    c12 = pad_sequence_transformer_collate_fn(batch)
    assert len(c12) == 12
    for got, key in zip(c12[8:], ("src_kpm", "tgt_kpm", "src_mask", "tgt_mask")):
        assert torch.equal(got, torch.from_numpy(g[key])), key
    assert torch.equal(make_padding_mask(torch.tensor([9, 6, 2])), torch.from_numpy(g["mask_9_6_2"]))


def test_synthetic_dataset_items():
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import SyntheticArtSpeechDataset
    voc = {"<blank>": 0, "<unk>": 1, "a": 2, "b": 3}
    ds = SyntheticArtSpeechDataset(5, voc, ["tongue", "lower-lip"], n_samples=7, min_len=2, max_len=6, seed=3)
    assert ds.articulators == ["lower-lip", "tongue"] and len(ds) == 5
    item = ds[2]
    assert len(item) == 8 and item[1].dtype == torch.long and item[2].shape[1:] == (2, 2, 7)
    assert item[1].min() >= 2 and torch.equal(ds[2][2], item[2])  # pad id 0 never drawn; deterministic


def test_round_robin_sharding():
    from artspeech_amd import distributed as dp
    lengths = torch.tensor([9, 8, 7, 5, 3, 1], dtype=torch.int32)
    tokens = torch.arange(6 * 9).view(6, 9)
    targets = torch.rand(6, 9, 2, 2, 4)
    seen = []
    for r in range(4):
        tok, tgt, ln, n_valid = dp.shard_batch(tokens, targets, lengths, r, 4)
        assert n_valid == 33 and tok.shape[1] == int(ln.max()) and tgt.shape[:2] == tok.shape
        assert bool((ln[1:] <= ln[:-1]).all())  # shard stays sorted descending
        seen += dp.shard_indices(6, r, 4)
    assert sorted(seen) == list(range(6))
    with pytest.raises(ValueError):
        dp.shard_batch(tokens[:2], targets[:2], lengths[:2], 3, 4)


# ------------------------------------------------------------------------------------------- DP rehearsal (gloo)
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, out_path):
    import torch.distributed as dist
    from artspeech_amd import distributed as dp
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech
    from oracle import artspeech_oracle as O  # the per-rank compute of this CPU rehearsal
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)             # ranks start different ...
    model = ArtSpeech(13, 2, embed_dim=16, hidden_size=32, n_samples=6)
    dp.broadcast_parameters(model)            # ... and are made identical by one broadcast
    rng = np.random.RandomState(0)            # every rank builds the same global batch
    B, T = 5, 9
    lengths = torch.tensor([9, 7, 6, 3, 1], dtype=torch.int32)
    tokens = torch.from_numpy(rng.randint(1, 13, (B, T)))
    targets = torch.from_numpy(rng.rand(B, T, 2, 2, 6).astype(np.float32))
    for b, l in enumerate(lengths):
        tokens[b, l:] = 0
        targets[b, l:] = 0
    sd = {k: v.numpy() for k, v in model.state_dict().items()}

    def grads_of(tok, tgt, ln, n_valid):
        out, cache = O.artspeech_fwd(sd, tok.numpy(), ln.numpy(), 2)
        dist_ = O.euclidean_distance(out, tgt.numpy().astype(np.float64)[:, :out.shape[1]])
        mask = O.make_padding_mask(ln.numpy())
        scale = dp.loss_scale(n_valid, 2, 6)
        loss = (dist_ * mask[:, :, None, None]).sum() * scale
        _, dout = O.masked_euclid_loss(out, tgt.numpy()[:, :out.shape[1]], ln.numpy())
        dout = dout * (mask.sum() * 2 * 6) * scale   # re-normalise the oracle's local mean to the global count
        g = O.artspeech_bwd(dout, cache, 2)
        flat = torch.zeros_like(model.flat.data)
        for k, (off, shape) in model._views.items():
            flat[off:off + int(np.prod(shape))] = torch.from_numpy(np.asarray(g[k], np.float32).reshape(-1))
        return loss, flat

    tok, tgt, ln, n_valid = dp.shard_batch(tokens, targets, lengths, rank, world)
    loss, flat = grads_of(tok, tgt, ln, n_valid)
    dp.all_reduce_flat(flat)
    loss_t = dp.all_reduce_flat(torch.tensor([loss], dtype=torch.float64))
    if rank == 0:
        full_loss, full_flat = grads_of(tokens, targets, lengths, int(lengths.sum()))
        torch.save({"loss": float(loss_t), "full_loss": float(full_loss),
                    "err": float((flat - full_flat).abs().max() / full_flat.abs().max())}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_matches_full_batch_gloo(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "dp.pt")
    mp.spawn(_dp_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    assert abs(res["loss"] - res["full_loss"]) < 1e-9      # shard losses SUM to the full-batch mean
    assert res["err"] < 1e-5                               # summed shard gradients == full-batch gradients


@pytest.mark.parametrize("name", ["deepspeech2_small", "deepspeech2_plain"])
def test_scorer_state_dict_keys_and_seeded_init_match_reference(name):
    """Same constructor, key names/shapes and (same seed) the same initial weights as the reference DeepSpeech2."""
    import json
    from conftest import GOLDEN, load_golden, split_wg
    from artspeech_amd.phoneme_recognition import DeepSpeech2
    g = load_golden(name)
    w, _ = split_wg(g)
    c = [int(v) for v in g["cfg"]]
    with open(os.path.join(GOLDEN, "checksums.json")) as f:
        chk = json.load(f)["cases"][name]
    torch.manual_seed(chk["seed"])
    m = DeepSpeech2(c[0], c[1], c[2], c[3], num_classes=c[4], num_features=c[5], dropout=0.1, adapter_out_features=c[6] or None)
    sd = m.state_dict()
    assert list(sd.keys()) == list(w.keys())
    assert all(tuple(sd[k].shape) == w[k].shape for k in w)
    assert m.total_parameters == chk["params"]
    init = float(sum(p.detach().double().abs().sum() for p in m.parameters()))
    assert abs(init - chk["init_abs_sum"]) < 1e-6 * chk["init_abs_sum"]
    with pytest.raises(RuntimeError):  # inference only, and never on the CPU
        m(torch.zeros(1, c[0], c[5], 4))


@pytest.mark.parametrize("name", ["pc_lstm_small", "pc_gru_small"])
def test_principal_components_model_keys_and_seeded_init_match_reference(name):
    """RNNType switch: same constructor, state_dict keys/shapes and (same seed) initial weights as the reference."""
    import json
    from conftest import GOLDEN
    from artspeech_amd.phoneme_to_articulation import RNNType
    from artspeech_amd.phoneme_to_articulation.principal_components.models import PrincipalComponentsArtSpeech
    g = load_golden(name)
    w, _ = split_wg(g)
    V, E, H, latent, lstm = (int(v) for v in g["cfg"])
    with open(os.path.join(GOLDEN, "checksums.json")) as f:
        chk = json.load(f)["cases"][name]
    torch.manual_seed(chk["seed"])
    m = PrincipalComponentsArtSpeech(V, chk["comps"], embed_dim=E, hidden_size=H, rnn=RNNType.LSTM if lstm else "gru")
    sd = m.state_dict()
    assert m.latent_size == latent and list(sd.keys()) == list(w.keys())
    assert all(tuple(sd[k].shape) == w[k].shape for k in w)
    init = float(sum(p.detach().double().abs().sum() for p in m.parameters()))
    assert abs(init - chk["init_abs_sum"]) < 1e-6 * chk["init_abs_sum"]
    with pytest.raises(RuntimeError):  # never on the CPU
        m(torch.zeros(2, 3, dtype=torch.long), [3, 2])


def test_xarticul_text_round_trip(tmp_path):
    from artspeech_amd.helpers import npy_to_xarticul, xarticul_to_npy
    arr = np.array([[0.25, 0.5], [1.0, -2.125], [3.0, 4.0]])
    path = os.path.join(tmp_path, "contour.txt")
    lines = npy_to_xarticul(arr, path)
    assert lines == ["0.25 0.5", "1.0 -2.125", "3.0 4.0", "-1 -1"]  # known answer: str() of the coordinates + end tag
    with open(path) as f:
        assert f.read() == "\n".join(lines)
    assert np.array_equal(xarticul_to_npy(path), arr)


def test_save_outputs_writes_reference_layout(tmp_path):
    """<sentence>/contours/<frame>_<articulator>[_true].npy for valid frames only + phonemes.csv (pandas layout)."""
    from artspeech_amd.phoneme_to_articulation import save_outputs
    rng = np.random.default_rng(0)
    out, tgt = rng.random((2, 3, 2, 2, 5), dtype=np.float32), rng.random((2, 3, 2, 2, 5), dtype=np.float32)
    save_outputs(["s0", "s1"], [[10, 11, 12], [20, 21]], torch.from_numpy(out), torch.from_numpy(tgt), [3, 2],
                 [["a", "b", "c"], ["d", "e"]], ["tongue", "lower-lip"], str(tmp_path))
    names = sorted(["tongue", "lower-lip"])  # channel order = sorted articulator names
    assert np.array_equal(np.load(os.path.join(tmp_path, "s1", "contours", "21_tongue.npy")), out[1, 1, names.index("tongue")])
    assert np.array_equal(np.load(os.path.join(tmp_path, "s0", "contours", "12_lower-lip_true.npy")), tgt[0, 2, names.index("lower-lip")])
    assert not os.path.exists(os.path.join(tmp_path, "s1", "contours", "22_tongue.npy"))  # padded frame: nothing written
    assert len(os.listdir(os.path.join(tmp_path, "s1", "contours"))) == 2 * 2 * 2
    with open(os.path.join(tmp_path, "s1", "phonemes.csv")) as f:
        assert f.read() == "sentence,frame,phoneme\ns1,20,d\ns1,21,e\n"
    import pandas as pd
    expect = pd.DataFrame([{"sentence": "s0", "frame": 10 + i, "phoneme": p} for i, p in enumerate("abc")]).to_csv(index=False)
    with open(os.path.join(tmp_path, "s0", "phonemes.csv")) as f:
        assert f.read() == expect
    with pytest.raises(NotImplementedError):
        save_outputs(["s0"], [[1]], out[:1], tgt[:1], [1], [["a"]], ["tongue", "lower-lip"], str(tmp_path), regularize_out=True)


def test_air_column_files_round_trip(tmp_path):
    """<frame>.npy = (2 walls, 2 coordinates, Nw) with the internal wall first, stacked to (frames, 2, 2, Nw) float64."""
    from artspeech_amd.area_function import load_air_columns, save_air_column
    rng = np.random.default_rng(1)
    walls = [(rng.random((7, 2)), rng.random((7, 2))) for _ in range(3)]
    for f, (inner, outer) in zip((101, 102, 103), walls):
        arr = save_air_column(os.path.join(tmp_path, f"{f}.npy"), inner, outer)
        assert arr.shape == (2, 2, 7)
    air = load_air_columns(str(tmp_path), [101, 103])
    assert air.shape == (2, 2, 2, 7) and air.dtype == torch.float64
    assert np.array_equal(air[1, 0].numpy(), walls[2][0].T) and np.array_equal(air[0, 1].numpy(), walls[0][1].T)


def test_transformer_seeded_init_matches_reference():
    """Same torch seed => the reference's initial weights (construction-order draws; layers of a stack start identical)."""
    import json
    from conftest import GOLDEN
    from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
    g = load_golden("transformer_small")
    V, A, d, h, L, nf = (int(v) for v in g["cfg"])
    with open(os.path.join(GOLDEN, "checksums.json")) as f:
        chk = json.load(f)["cases"]["transformer_small"]
    torch.manual_seed(chk["seed"])
    m = ArtSpeechTransformer(V, A, embed_dim=d, num_heads=h, num_layers=L, num_feat=nf)
    sd = m.state_dict()
    init = float(sum(v.double().abs().sum() for k, v in sd.items() if k != "pos_encoding.pe"))
    assert abs(init - chk["init_abs_sum"]) < 1e-9 * chk["init_abs_sum"]
    assert torch.equal(sd["decoder.layers.0.feed_forward.1.weight"], sd["decoder.layers.1.feed_forward.1.weight"])
    assert torch.equal(sd["encoder.layers.0.linear1.weight"], sd["encoder.layers.1.linear1.weight"])


def test_pc_autoencoder_keys_and_seeded_init_match_reference():
    import json
    from conftest import GOLDEN
    from artspeech_amd.phoneme_to_articulation.principal_components.models import MultiArticulatorAutoencoder
    g = load_golden("pc_autoencoder")
    w, _ = split_wg(g)
    with open(os.path.join(GOLDEN, "checksums.json")) as f:
        chk = json.load(f)["cases"]["pc_autoencoder"]
    torch.manual_seed(chk["seed"])
    m = MultiArticulatorAutoencoder(in_features=20, indices_dict=chk["comps"], hidden_features=16)
    sd = m.state_dict()
    assert list(sd.keys()) == list(w.keys()) and all(tuple(sd[k].shape) == w[k].shape for k in w)
    assert m.total_parameters == chk["params"] and m.latent_size == 9
    init = float(sum(p.detach().double().abs().sum() for p in m.parameters()))
    assert abs(init - chk["init_abs_sum"]) < 1e-9 * chk["init_abs_sum"]


def test_ctypes_structs_match_the_header_layout():
    """The ctypes mirrors in _lib.py against a C compiler's view of include/artspeech_hip.h (sizes and the offsets of the
    fields appended this round): a drifted struct would silently shift every later argument."""
    import ctypes as C
    import subprocess
    import tempfile
    from artspeech_amd import _lib
    src = r"""
#include <stddef.h>
#include <stdio.h>
#include "artspeech_hip.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(as_opts), offsetof(as_opts, dout_presigmoid), sizeof(as_gemm), offsetof(as_gemm, precision),
           offsetof(as_gemm, b_kshift_batch), sizeof(as_dims), offsetof(as_gemm, cu_budget), offsetof(as_gemm, res_off),
           offsetof(as_gemm, mask_batch), offsetof(as_gemm, k_tri));
    return 0;
}
"""
    inc = os.path.join(ROOT, "include")
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", inc, c, "-o", exe])
        got = [int(x) for x in subprocess.check_output([exe]).split()]
    want = [C.sizeof(_lib.Opts), _lib.Opts.dout_presigmoid.offset, C.sizeof(_lib.Gemm), _lib.Gemm.precision.offset,
            _lib.Gemm.b_kshift_batch.offset, C.sizeof(_lib.Dims), _lib.Gemm.cu_budget.offset, _lib.Gemm.res_off.offset,
            _lib.Gemm.mask_batch.offset, _lib.Gemm.k_tri.offset]
    assert got == want, (got, want)


def test_key_major_mask_is_transposed_padded_and_cached():
    """Host side of as_attention_fwd's mask contract: (B, T, Tk) -> (B, Tk rounded up to 32, T), finite padding, one
    transpose per mask tensor and version."""
    from artspeech_amd.phoneme_to_articulation.transformer import ops
    B, T, Tk = 2, 5, 37
    m = torch.randn(B, T, Tk)
    m[0, 1, 3] = float("-inf")
    mt = ops._key_major_mask(m, Tk, T)
    assert mt.shape == (B, 64, T) and torch.equal(mt[:, :Tk], m.transpose(1, 2)) and torch.equal(mt[:, Tk:], torch.zeros(B, 64 - Tk, T))
    assert ops._key_major_mask(m, Tk, T) is mt            # same tensor, same version: cached
    m[0, 0, 0] = 7.0                                      # in-place change bumps the version
    mt2 = ops._key_major_mask(m, Tk, T)
    assert mt2 is not mt and mt2[0, 0, 0] == 7.0


def test_bench_self_launch_builds_a_child_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` outside torchrun: the parent spawns `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>` as a CHILD (subprocess, never exec), falls back to the
    shared-card gloo rehearsal when fewer cards than ranks are visible, and refuses a rehearsal that would put too many
    processes on one card.  No GPU is touched: device_count() is the only torch.cuda call on this path."""
    import importlib
    import sys as _sys
    import types
    monkeypatch.setattr(_sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    bench = importlib.import_module("bench")
    calls = {}

    def fake_call(cmd, env=None):
        calls["cmd"], calls["env"] = cmd, env
        return 7
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    args = types.SimpleNamespace(gpus=2)
    assert bench.self_launch(args) == 7                       # the children's exit code is the parent's
    cmd = calls["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert calls["env"]["ARTSPEECH_DIST_BACKEND"] == "gloo"   # one card, two ranks: rehearsal
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    args.gpus = 8
    bench.self_launch(args)
    assert "ARTSPEECH_DIST_BACKEND" not in calls["env"] or calls["env"].get("ARTSPEECH_DIST_BACKEND") == os.environ.get("ARTSPEECH_DIST_BACKEND")
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit):
        bench.self_launch(args)                               # 8 ranks on one card: refused
