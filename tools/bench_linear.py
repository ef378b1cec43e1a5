"""nn.Linear forward shapes through as_linear_fwd in the three forms the library has: the fp32 matrix instruction (mode 0), the
split arithmetic with pre-split weight planes (mode 1, planes_ws) and with both operands split in the kernel (mode 1, no scratch):
microseconds per launch and the error against an fp64 product.   usage: python tools/bench_linear.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SHAPES = [("GRU input projection", 6400, 768, 256), ("trunk Linear", 6400, 128, 256), ("GRU input gradient", 6400, 256, 768),
          ("transformer block Linear x 8", 51200, 256, 256), ("transformer block group x 110 (as one matrix)", 704000, 256, 256),
          ("big", 51200, 1024, 1024)]
for name, M, N, K in SHAPES:
    torch.manual_seed(0)
    a = torch.randn(M, K, device=dev)
    w = (torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5
    b = torch.zeros(N, device=dev)
    ref = a[:51200].double() @ w.double().T      # (error measured on the first 51200 rows)
    out = torch.empty(M, N, device=dev)
    pw = torch.empty(max(64, L.as_linear_planes_floats(N, K)), device=dev)
    print(f"{name}: M={M} N={N} K={K}  {2e-9 * M * N * K:.2f} GFLOP")
    for label, mode, scratch in (("fp32 MFMA", 0, None), ("split, weight planes", 1, pw), ("split, in-kernel", 1, None)):
        L.as_set_matrix_arith(mode)

        def run():
            _lib.check(L.as_linear_fwd(_lib.ptr(a), K, _lib.ptr(w), K, _lib.ptr(b), _lib.ptr(out), N, M, N, K, 0, _lib.ptr(scratch), _lib.stream_ptr()))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        err = (out[:51200].double() - ref)
        print(f"  {label:24s} {us:8.1f} us  {2e-6 * M * N * K / us:7.1f} TF/s   rms err / rms C {(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item():.2e}")
L.as_set_matrix_arith(1)
