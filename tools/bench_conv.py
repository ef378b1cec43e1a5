"""Microbenchmark of the scorer's 32->32 3x3 convolution (as_conv3x3_c32) at the thesis shape: B=32, T=200, D=80.
usage: python tools/bench_conv.py [B] [T] [D] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402

B, T, D, iters = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 32), (2, 200), (3, 80), (4, 50)))
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(B, T, D, 32, device=dev)
res = torch.randn_like(x)
w = torch.randn(9, 32, 32, device=dev) * 0.05
bias = torch.randn(32, device=dev)
y = torch.empty_like(x)
L, st = _lib.lib(), _lib.stream_ptr()
flops = 2 * 9 * 32 * 32 * B * T * D
for name, r in (("plain", None), ("skip-input", res)):
    rp = _lib.ptr(r) if r is not None else None
    for _ in range(5):
        _lib.check(L.as_conv3x3_c32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), rp, _lib.ptr(y), B, T, D, st), "conv")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _lib.check(L.as_conv3x3_c32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), rp, _lib.ptr(y), B, T, D, st), "conv")
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"conv3x3 32->32 {name:10s} B={B} T={T} D={D}: {us:7.1f} us  {flops / us / 1e6:6.1f} TFLOP/s (fp32 MFMA peak 157.3)  "
          f"{(2 + (r is not None)) * x.numel() * 4 / us / 1e3:6.0f} GB/s algorithmic", flush=True)
