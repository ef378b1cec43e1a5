"""NT / NN / TN layouts of as_gemm_f32 at one large and one K = 256 shape (which operand image costs what).
usage: python tools/bench_gemm_layouts.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
st = _lib.stream_ptr()
for (M, N, K) in [(4096, 4096, 4096), (70400, 256, 256), (256, 256, 70400)]:
    a = torch.randn(M * K, device=dev)
    b = torch.randn(N * K, device=dev)
    c = torch.empty(M * N, device=dev)
    for name, (a_i, a_k, b_j, b_k) in (("nt", (K, 1, K, 1)), ("nn", (K, 1, 1, N)), ("tn", (1, M, 1, N)), ("tt", (1, M, K, 1))):
        g = _lib.Gemm()
        g.A, g.B, g.C = a.data_ptr(), b.data_ptr(), c.data_ptr()
        g.M, g.N, g.K = M, N, K
        g.a_i, g.a_k, g.b_j, g.b_k, g.ldc = a_i, a_k, b_j, b_k, N
        g.batch = 1
        for _ in range(2):
            _lib.check(L.as_gemm_f32(C.byref(g), st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.as_gemm_f32(C.byref(g), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        print(f"{name} M={M:6d} N={N:5d} K={K:6d}: {us:9.1f} us {2 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
