"""Asymptotic and shape sweep of as_gemm_f32 (NT layout): where does the fp32-MFMA kernel lose time?
usage: python tools/bench_gemm_sweep.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
st = _lib.stream_ptr()


def run(M, N, K, batch=1, iters=10):
    a = torch.randn(batch, M, K, device=dev)
    b = torch.randn(batch, N, K, device=dev)
    c = torch.empty(batch, M, N, device=dev)
    g = _lib.Gemm()
    g.A, g.B, g.C = a.data_ptr(), b.data_ptr(), c.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.a_i, g.a_k, g.b_j, g.b_k, g.ldc = K, 1, K, 1, N
    g.batch, g.a_batch, g.b_batch, g.c_batch = batch, M * K, N * K, M * N
    for _ in range(2):
        _lib.check(L.as_gemm_f32(C.byref(g), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.as_gemm_f32(C.byref(g), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    tiles = batch * ((M + 127) // 128) * ((N + 127) // 128)
    print(f"M={M:6d} N={N:5d} K={K:5d} batch={batch:3d} tiles128={tiles:6d}: {us:9.1f} us {2 * M * N * K * batch / us / 1e6:7.1f} TFLOP/s", flush=True)


for shape in [(8192, 8192, 8192, 1), (4096, 4096, 4096, 1), (8192, 8192, 256, 1), (6400, 256, 256, 11), (6400, 256, 256, 12), (6144, 256, 256, 16),
              (6400, 256, 4096, 11), (6400, 2816, 128, 1), (6400, 2816, 256, 1), (70400, 256, 256, 1), (6400, 768, 256, 1), (6400, 128, 256, 1)]:
    run(*shape)
