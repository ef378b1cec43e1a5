"""BASELINE configs[4] on one GPU (the 8-GPU run shards utterances, no exchange): phoneme -> contour (transformer variant,
teacher-forced forward, d=256 L=6 A=11 N=50) -> tract variables + vocal-tract area function of every frame -> DeepSpeech2
articulatory scorer -> top-1 phoneme indices, B=32, T=200, synthetic inputs, random-init weights.
usage: python tools/bench_pipeline.py [B] [T] [iters]             (bench.py imports `run` for its `pipeline_c5_1gpu` key)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

ARTS = sorted(["arytenoid-cartilage", "epiglottis", "lower-incisor", "lower-lip", "pharynx", "soft-palate-midline", "thyroid-cartilage",
               "tongue", "upper-incisor", "upper-lip", "vocal-folds"])
STAGES = ["phoneme->contour (transformer fwd)", "tract variables", "area function + resampling", "scorer + top-1"]


def run(B=32, T=200, iters=3, dev=None, log=print):
    from artspeech_amd.area_function import area_function_batched, evenly_spaced_fx_batched
    from artspeech_amd.phoneme_recognition import DeepSpeech2, top1_phonemes
    from artspeech_amd.tract_variables import tract_variables_batched
    from bench_transformer import A, NFEAT, V, make_case
    dev = dev or torch.device("cuda:0")
    p2a, tokens, _targets, shifted, _lengths, kw = make_case(B, T, dev)
    scorer = DeepSpeech2(2, 4, 2, 64, num_classes=V, num_features=A * NFEAT // 2, adapter_out_features=80).to(dev).eval()
    tongue, pharynx = ARTS.index("tongue"), ARTS.index("pharynx")
    nf = NFEAT

    def stage_times():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        with torch.no_grad():
            ev[0].record()
            contours = p2a(tokens, shifted, **kw)                                          # (B, T, A, 2, N)
            ev[1].record()
            tv, _poc1, _poc2, _ = tract_variables_batched(contours.reshape(B * T, A, 2, nf // 2), ARTS)
            ev[2].record()
            # two predicted contours stand in for the internal / external walls of the tube (vt_shape_gen builds the real ones)
            air = torch.stack([contours[:, :, tongue], contours[:, :, pharynx]], dim=2).reshape(B * T, 2, 2, nf // 2).double()
            dists, fx = area_function_batched(air)
            af = evenly_spaced_fx_batched(dists, fx, 200)
            ev[3].record()
            x = contours.permute(0, 3, 2, 4, 1).reshape(B, 2, A * nf // 2, T)              # (B, 2, A*N, T): planes x features x time
            top = top1_phonemes(scorer(x))
            ev[4].record()
        torch.cuda.synchronize()
        return [ev[i].elapsed_time(ev[i + 1]) for i in range(4)], (contours, tv, af, top)

    stage_times()
    t0 = time.perf_counter()
    acc = [0.0] * 4
    for _ in range(iters):
        ts, outs = stage_times()
        acc = [a + t for a, t in zip(acc, ts)]
    wall = (time.perf_counter() - t0) / iters
    contours, tv, af, top = outs
    assert torch.isfinite(contours).all() and torch.isfinite(tv).all() and torch.isfinite(af).all() and top.shape == (B, T, 1)
    res = {"ms_per_batch": round(wall * 1e3, 2), "frames_s": round(B * T / wall, 1),
           "stages_ms": {n: round(a / iters, 3) for n, a in zip(STAGES, acc)}}
    for n, a in zip(STAGES, acc):
        log(f"{n:36s} {a / iters:9.3f} ms")
    log(f"pipeline B={B} T={T}: {wall * 1e3:.1f} ms per batch -> {B * T / wall:.0f} frames/s on one GPU")
    del p2a, scorer
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    t = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    run(b, t, n, log=lambda m: print(m, flush=True))
