"""Attention core and grouped projections of one ChannelInteractionsLayer call of the transformer variant, alone:
G = 110 (target, source) channel pairs, B = 32, T = 200, d = 256, 4 heads.  usage: python tools/bench_attention.py [G] [B] [T]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd.phoneme_to_articulation.transformer import ops  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 110
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
d, heads = 256, 4
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


Q = torch.randn(G, B * T, d, device=dev, requires_grad=True)
K = torch.randn(G, B * T, d, device=dev, requires_grad=True)
V = torch.randn(G, B * T, d, device=dev, requires_grad=True)
mask = torch.zeros(B, T, T, device=dev)
kpm = torch.zeros(B, T, device=dev)
flop_f = 2.0 * G * B * heads * (T * T * (d // heads)) * 2
ms = timed(lambda: ops.Attention.apply(Q, K, V, mask, kpm, B, heads))
print(f"attention forward  G={G} B={B} T={T}: {ms:8.2f} ms  {flop_f / ms / 1e9:6.1f} TFLOP/s   (training: unfused, keeps P)")
with torch.no_grad():
    cmask = torch.triu(torch.full((T, T), float("-inf"), device=dev), 1).expand(B, T, T).contiguous()
    Qd, Kd, Vd = Q.detach(), K.detach(), V.detach()  # nothing asks for a gradient: the fused kernel
    for name, mk in (("no mask", None), ("causal mask", cmask)):
        ms_f = timed(lambda: ops.Attention.apply(Qd, Kd, Vd, mk, kpm, B, heads))
        ops.FUSED_ATTENTION = False
        ms_u = timed(lambda: ops.Attention.apply(Qd, Kd, Vd, mk, kpm, B, heads))
        ops.FUSED_ATTENTION = True
        print(f"attention inference ({name}): fused {ms_f:7.2f} ms  {flop_f / ms_f / 1e9:6.1f} TFLOP/s   unfused {ms_u:7.2f} ms")
g = torch.randn(G, B * T, d, device=dev)


def fb():
    for t in (Q, K, V):
        t.grad = None
    ops.Attention.apply(Q, K, V, mask, kpm, B, heads).backward(g)


ms2 = timed(fb)
print(f"attention fwd+bwd: {ms2:8.2f} ms  {3 * flop_f / ms2 / 1e9:6.1f} TFLOP/s   (backward alone {ms2 - ms:.2f} ms)")

x = torch.randn(G, B * T, d, device=dev, requires_grad=True)
W = torch.randn(G, d, d, device=dev, requires_grad=True)
b = torch.randn(G, d, device=dev, requires_grad=True)
flop_l = 2.0 * G * B * T * d * d
ms = timed(lambda: ops.GroupedLinear.apply(x, W, b, tuple(range(G)), True))
print(f"grouped linear forward (one of 7 per block): {ms:8.2f} ms  {flop_l / ms / 1e9:6.1f} TFLOP/s")


def fbl():
    for t in (x, W, b):
        t.grad = None
    ops.GroupedLinear.apply(x, W, b, tuple(range(G)), True).backward(g)


ms2 = timed(fbl)
print(f"grouped linear fwd+bwd: {ms2:8.2f} ms  {3 * flop_l / ms2 / 1e9:6.1f} TFLOP/s   (backward alone {ms2 - ms:.2f} ms)")
