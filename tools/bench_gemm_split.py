"""Accuracy and speed of the split-precision (bf16 MFMA) forward GEMM against the exact fp32-MFMA kernel.
usage: python tools/bench_gemm_split.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
st = _lib.stream_ptr()


def make(M, N, K, batch, precision, a, b, c, bias=None, act=0):
    g = _lib.Gemm()
    g.A, g.B, g.C = a.data_ptr(), b.data_ptr(), c.data_ptr()
    g.bias = bias.data_ptr() if bias is not None else None
    g.M, g.N, g.K = M, N, K
    g.a_i, g.a_k, g.b_j, g.b_k, g.ldc = K, 1, K, 1, N
    g.batch, g.a_batch, g.b_batch, g.c_batch, g.bias_batch = batch, M * K, N * K, M * N, N
    g.act, g.precision = act, precision
    return g


torch.manual_seed(0)
M, N, K = 500, 200, 264  # ragged tile edges
a, b, bias = torch.randn(1, M, K, device=dev), torch.randn(1, N, K, device=dev), torch.randn(N, device=dev)
want = torch.relu(a[0].double() @ b[0].double().T + bias.double())
scale = (a[0].double().abs() @ b[0].double().abs().T).max()
for prec, name in ((0, "fp32 MFMA"), (1, "bf16 x3"), (2, "bf16 x6")):
    c = torch.zeros(1, M, N, device=dev)
    g = make(M, N, K, 1, prec, a, b, c, bias, act=1)
    _lib.check(L.as_gemm_f32(C.byref(g), st))
    err = (c[0].double() - want).abs().max()
    print(f"{name:10s} max |err| = {float(err):.3e}  relative to max sum|a||b| = {float(err / scale):.3e}", flush=True)

for (M, N, K, batch) in [(70400, 256, 256, 1), (6400, 256, 256, 11), (8192, 8192, 256, 1), (4096, 4096, 4096, 1), (6400, 2816, 128, 1)]:
    a, b = torch.randn(batch, M, K, device=dev), torch.randn(batch, N, K, device=dev)
    c = torch.empty(batch, M, N, device=dev)
    for prec, name in ((0, "fp32 MFMA"), (1, "bf16 x3"), (2, "bf16 x6")):
        g = make(M, N, K, batch, prec, a, b, c)
        for _ in range(2):
            _lib.check(L.as_gemm_f32(C.byref(g), st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.as_gemm_f32(C.byref(g), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"M={M:6d} N={N:5d} K={K:5d} batch={batch:3d} {name:10s}: {us:9.1f} us {2 * M * N * K * batch / us / 1e6:7.1f} TFLOP/s (fp32-equivalent)", flush=True)
