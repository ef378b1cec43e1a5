"""gemm_s6.hip's weight-gradient orientation (as_gemm.precision = 3, both operands reduction-strided) at 1, 2, 3 ... workgroups
per CU: 256 x 256 x K per batch member = 4 tiles of 128 x 128, batch = 64 n -> n workgroups per CU.
usage: python tools/bench_tn_occupancy.py [K] [iters]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L = _lib.lib()
dev = torch.device("cuda", 0)
for batch in (8, 64, 128, 192, 256, 110):
    A = torch.randn(batch, K, 256, device=dev)
    B = torch.randn(batch, K, 256, device=dev)
    Cm = torch.empty(batch, 256, 256, device=dev)
    g = _lib.Gemm()
    g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), Cm.data_ptr()
    g.M, g.N, g.K, g.batch = 256, 256, K, batch
    g.a_i, g.a_k, g.b_j, g.b_k, g.ldc = 1, 256, 1, 256, 256
    g.a_batch, g.b_batch, g.c_batch = K * 256, K * 256, 256 * 256
    g.precision = 3
    st = _lib.stream_ptr()
    for _ in range(2):
        _lib.check(L.as_gemm_f32(C.byref(g), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.as_gemm_f32(C.byref(g), st)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    fl = 2.0 * K * 256 * 256 * batch
    print(f"batch {batch:4d} ({batch * 4 / 256:.2f} workgroups per CU): {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s-equivalent", flush=True)
    del A, B, Cm
