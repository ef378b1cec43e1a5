"""Weight-gradient (TN) GEMM shapes of the transformer's grouped linears: C[g] (256x256) = A[g]^T (6400x256) . B[g] (6400x256), g = 110 / 11,
under the tile configurations of as_gemm_f32 (AS_GEMM_TILE) -- tuning aid.  usage: AS_GEMM_TILE=64x128 python tools/bench_gemm_tn.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
st = _lib.stream_ptr()
slab = torch.empty(int(os.environ.get("SLAB_MFLOATS", "64")) << 20, device=dev)  # split-K workspace
SHAPES = [(110, 256, 256, 6400), (11, 256, 256, 6400), (110, 256, 256, 3200), (11, 256, 2560, 6400), (192, 256, 256, 6400),
                     (384, 256, 256, 3200)]
if os.environ.get("TN_SHAPE"):  # e.g. TN_SHAPE=0 for a single (profiled) shape
    SHAPES = [SHAPES[int(os.environ["TN_SHAPE"])]]
for (G, M, N, K) in SHAPES:
    a = torch.randn(G, K, M, device=dev)
    b = torch.randn(G, K, N, device=dev)
    c = torch.empty(G, M, N, device=dev)
    db = torch.empty(G, M, device=dev)
    g = _lib.Gemm()
    g.A, g.B, g.C = a.data_ptr(), b.data_ptr(), c.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.a_i, g.a_k, g.b_j, g.b_k, g.ldc = 1, M, 1, N, N
    g.batch, g.a_batch, g.b_batch, g.c_batch = G, K * M, K * N, M * N
    g.colsum, g.colsum_batch = db.data_ptr(), M
    g.splitk_ws, g.splitk_ws_floats = slab.data_ptr(), slab.numel()
    for _ in range(2):
        _lib.check(L.as_gemm_f32(C.byref(g), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.as_gemm_f32(C.byref(g), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print(f"{os.environ.get('AS_GEMM_TILE', 'auto'):8s} G={G:4d} M={M} N={N} K={K}: {us:9.1f} us {2 * M * N * K * G / us / 1e6:7.1f} TFLOP/s", flush=True)
