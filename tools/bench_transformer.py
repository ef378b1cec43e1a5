"""Throughput of the transformer variant at BASELINE configs[3] (d=256, L=6, A=11, B=32, T=200): forward only and
forward + masked Euclidean loss + backward to every parameter gradient.
usage: python tools/bench_transformer.py [B] [T] [iters]          (bench.py imports `run` for its `transformer_c4` key)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

V, A, D_MODEL, HEADS, LAYERS, NFEAT = 45, 11, 256, 4, 6, 100
FWD_FLOPS_PER_FRAME = 2 * 0.503e9   # ~0.50 GMAC per frame forward; fwd+bwd = 3x
F32_MFMA_PEAK_TFLOPS = 157.3


def issued_fraction(T):
    """Share of the dense-equivalent FLOPs the kernels really issue.  Every one of the 132 blocks of a decoder layer is causally
    masked in training and the attention kernels skip key blocks that lie wholly above the diagonal (32 keys per
    block: attention.hip), in the forward (QK^T, PV) and in the backward (dS, dQ, dK, dV) alike: of the nq x nq (query block, key
    block) pairs, nq (nq + 1) / 2 are computed.  The attention products are 2 T d of a block's 7 d^2 + 2 T d MACs per frame."""
    nq = (T + 31) // 32
    computed = (nq + 1) / (2.0 * nq)
    attn_mac = 132 * LAYERS * 2 * T * D_MODEL                      # per frame, decoder blocks (the encoder's self-attention is not causal)
    return 1.0 - (1.0 - computed) * attn_mac / (FWD_FLOPS_PER_FRAME / 2)


def make_case(B, T, dev, seed=0):
    from artspeech_amd.phoneme_to_articulation.encoder_decoder.dataset import pad_sequence_transformer_collate_fn
    from artspeech_amd.phoneme_to_articulation.transformer.models import ArtSpeechTransformer
    torch.manual_seed(seed)
    model = ArtSpeechTransformer(V, A, embed_dim=D_MODEL, num_heads=HEADS, num_layers=LAYERS, num_feat=NFEAT).to(dev).eval()
    batch = [(f"s{i}", torch.randint(1, V, (T,)), torch.rand(T, A, 2, NFEAT // 2), ["p"] * T, torch.rand(T, 1, 2, NFEAT // 2),
              torch.tensor([], dtype=torch.int), list(range(T)), torch.zeros(T)) for i in range(B)]
    c = pad_sequence_transformer_collate_fn(batch)
    tokens, targets = c[1].to(dev), c[2].to(dev)
    shifted = torch.cat([torch.zeros(B, 1, A, NFEAT, device=dev), targets[:, 1:].reshape(B, T - 1, A, NFEAT)], dim=1)
    kw = dict(src_key_padding_mask=c[8].to(dev), tgt_key_padding_mask=c[9].to(dev), src_attn_mask=c[10].to(dev),
              tgt_attn_mask=c[11].to(dev))
    return model, tokens, targets, shifted, c[3], kw


def run(B=32, T=200, iters=3, dev=None, log=print):
    """Returns {"fwd": {...}, "fwd_bwd": {...}} (ms per pass, frames/s, TFLOP/s, fraction of the fp32 MFMA peak, peak GiB)."""
    from artspeech_amd.phoneme_to_articulation.metrics import masked_euclidean_loss
    dev = dev or torch.device("cuda:0")
    t0 = time.time()
    model, tokens, targets, shifted, lengths, kw = make_case(B, T, dev)
    log(f"transformer: {model.total_parameters} parameters, built in {time.time() - t0:.1f} s")
    flops = FWD_FLOPS_PER_FRAME * B * T
    res = {"params": int(model.total_parameters)}
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        out = model(tokens, shifted, **kw)
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = model(tokens, shifted, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
    res["fwd"] = {"ms_per_step": round(dt * 1e3, 2), "frames_s": round(B * T / dt, 1), "tflops": round(flops / dt / 1e12, 1),
                  "frac_of_157.3": round(flops / dt / 1e12 / F32_MFMA_PEAK_TFLOPS, 3),
                  "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}
    log(f"transformer forward B={B} T={T}: {res['fwd']}")

    # training step: forward + masked Euclidean loss + backward (no optimizer), dropout 0 like the parity runs;
    # eval() keeps the encoder's library-default dropout off (deterministic), gradients still flow
    torch.cuda.reset_peak_memory_stats()

    def step():
        for p_ in model.parameters():
            p_.grad = None
        loss = masked_euclidean_loss(model(tokens, shifted, **kw), targets, lengths)
        loss.backward()
        return loss

    loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    assert torch.isfinite(loss)
    fi = issued_fraction(T)
    res["fwd_bwd"] = {"ms_per_step": round(dt * 1e3, 2), "frames_s": round(B * T / dt, 1), "tflops": round(3 * flops / dt / 1e12, 1),
                      "frac_of_157.3": round(3 * flops / dt / 1e12 / F32_MFMA_PEAK_TFLOPS, 3), "loss": round(float(loss), 5),
                      "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                      # `tflops` prices the DENSE-EQUIVALENT work (SURVEY's 3 x 2 x 0.503 GMAC per frame); the kernels skip the
                      # causally masked key blocks of the decoder's attention: what they really issue is this share of it
                      "issued_share_of_dense": round(fi, 4), "tflops_issued": round(3 * flops * fi / dt / 1e12, 1),
                      "frac_of_157.3_issued": round(3 * flops * fi / dt / 1e12 / F32_MFMA_PEAK_TFLOPS, 3)}
    log(f"transformer fwd+bwd B={B} T={T}: {res['fwd_bwd']}")
    if os.environ.get("ARTSPEECH_BENCH_CHECKPOINT"):   # the same step with per-layer activation checkpointing
        model.checkpoint_layers = True
        torch.cuda.reset_peak_memory_stats()
        loss = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        res["fwd_bwd_checkpointed"] = {"ms_per_step": round(dt * 1e3, 2), "loss": round(float(loss), 5),
                                       "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}
        log(f"transformer fwd+bwd, decoder layers checkpointed: {res['fwd_bwd_checkpointed']}")
    del model
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    t = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    run(b, t, n, log=lambda m: print(m, flush=True))
