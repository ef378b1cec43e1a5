"""End-to-end training epoch through the reference-shaped host loop (DataLoader -> collate -> H2D -> model -> loss -> backward
-> torch Adam, one .item() per step as the reference does) on synthetic utterances of 200 frames, 11 articulators: what the
data path costs on top of the resident-input step that bench.py measures; then the same loop over an HBMResidentDataset (the
whole data set uploaded once, batches collated on the device) and the bare module-path step for comparison.
usage: python tools/bench_epoch.py [num_workers]"""
import os
import sys
import time

import torch
from torch.optim import Adam
from torch.utils.data import DataLoader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import (  # noqa: E402
    HBMResidentDataset, SyntheticArtSpeechDataset, pad_sequence_collate_fn)
from artspeech_amd.phoneme_to_articulation.encoder_decoder.models import ArtSpeech  # noqa: E402
from artspeech_amd.phoneme_to_articulation.metrics import EuclideanDistance  # noqa: E402
from artspeech_amd.settings import TRAIN  # noqa: E402
from train_phoneme_to_articulation import build_vocabulary, run_epoch  # noqa: E402

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 0
arts = ["arytenoid-cartilage", "epiglottis", "lower-incisor", "lower-lip", "pharynx", "soft-palate-midline", "thyroid-cartilage",
        "tongue", "upper-incisor", "upper-lip", "vocal-folds"]
vocab = build_vocabulary(None)
ds = SyntheticArtSpeechDataset(512, vocab, arts, seed=0, min_len=200, max_len=200)
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ArtSpeech(len(vocab), len(arts)).to(dev)
opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-6)
crit = EuclideanDistance("none")
for pin in (False, True):
    dl = DataLoader(ds, batch_size=32, shuffle=False, num_workers=workers, collate_fn=pad_sequence_collate_fn, pin_memory=pin,
                    persistent_workers=workers > 0)
    run_epoch(TRAIN, 0, model, dl, opt, crit, device=dev)  # warm-up epoch (workers start, caches fill)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = run_epoch(TRAIN, 1, model, dl, opt, crit, device=dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = len(dl)
    print(f"num_workers={workers} pin_memory={pin}: {dt / steps * 1e3:.2f} ms/step -> {512 * 200 / dt:.0f} frames/s end to end "
          f"(loss {info['loss']:.4f}); resident-input step: see bench.py", flush=True)

# ---- the same entry point over the HBM-resident data set: one upload, device-side collate, no per-step host copies
t0 = time.perf_counter()
rds = HBMResidentDataset(ds, dev)
torch.cuda.synchronize()
print(f"HBMResidentDataset: {len(rds)} utterances uploaded in {time.perf_counter() - t0:.2f} s "
      f"({rds._targets.numel() * 4 / 1e6:.0f} MB of contours)", flush=True)
dl = DataLoader(rds, batch_size=32, shuffle=True, num_workers=0, collate_fn=rds.collate, generator=torch.Generator().manual_seed(0))
run_epoch(TRAIN, 0, model, dl, opt, crit, device=dev)
torch.cuda.synchronize()
best = 1e9
for ep in range(3):
    t0 = time.perf_counter()
    info = run_epoch(TRAIN, 1 + ep, model, dl, opt, crit, device=dev)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
steps = len(dl)
ms_res = best / steps * 1e3
print(f"HBM-resident data set: {ms_res:.2f} ms/step -> {512 * 200 / best:.0f} frames/s end to end (loss {info['loss']:.4f})", flush=True)
# the engine's step on resident inputs (what bench.py times), same model size, for the ratio
from artspeech_amd.engine import TrainStep  # noqa: E402
batch = next(iter(dl))
tokens, targets, lengths = batch[1], batch[2], batch[3]
step = TrainStep(model, 32, 200, lr=1e-4, weight_decay=1e-6, pipeline=True)
ld = lengths.to(torch.int32).to(dev)
scale = 1.0 / (float(lengths.sum()) * len(arts) * 50)
for _ in range(20):
    step.step(tokens, ld, targets, scale)
step.flush()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    step.step(tokens, ld, targets, scale)
step.flush()
torch.cuda.synchronize()
ms_eng = (time.perf_counter() - t0) / 100 * 1e3
print(f"engine step on resident inputs: {ms_eng:.3f} ms/step; run_epoch over the resident data set reaches {100 * ms_eng / ms_res:.0f} % of "
      f"that rate", flush=True)
