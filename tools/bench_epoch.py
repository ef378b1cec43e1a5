"""End-to-end training epoch through the reference-shaped host loop (DataLoader -> collate -> H2D -> model -> loss -> backward
-> torch Adam, one .item() per step as the reference does) on synthetic utterances of 200 frames, 11 articulators: what the
data path costs on top of the resident-input step that bench.py measures.  usage: python tools/bench_epoch.py [num_workers]"""
import os
import sys
import time

import torch
from torch.optim import Adam
from torch.utils.data import DataLoader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import SyntheticArtSpeechDataset, pad_sequence_collate_fn  # noqa: E402
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
