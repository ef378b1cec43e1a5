"""Micro-benchmark of the fp32-MFMA GEMM shapes of the head (C ABI as_gemm_f32), for kernel tuning.
usage: python tools/bench_gemm.py [iters]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
R, A, D, H, O = 6400, 11, 256, 128, 100
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else None


def mk(**kw):
    g = _lib.Gemm()
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    return g


x = torch.randn(R, H, device=dev)
w1 = torch.randn(A * D, H, device=dev)
b1 = torch.randn(A * D, device=dev)
r1 = torch.randn(R, A * D, device=dev)
r2 = torch.randn(R, A * D, device=dev)
w2 = torch.randn(A, D, D, device=dev)
b2 = torch.randn(A * D, device=dev)
w3 = torch.randn(A, O, D, device=dev)
o3 = torch.randn(R, A * O, device=dev)
dw2 = torch.empty(A, D, D, device=dev)
dw1 = torch.empty(A * D, H, device=dev)
dxh = torch.empty(R, H, device=dev)
db = torch.empty(A * D, device=dev)
slab = torch.empty(8 << 20, device=dev)

cases = {
    "gemm1 nt 6400x2816x128": (mk(A=x, B=w1, C=r1, bias=b1, M=R, N=A * D, K=H, a_i=H, a_k=1, b_j=H, b_k=1, ldc=A * D, batch=1, act=1), 2 * R * A * D * H),
    "gemm2 nt b11 6400x256x256": (mk(A=r1, B=w2, C=r2, bias=b2, M=R, N=D, K=D, a_i=A * D, a_k=1, b_j=D, b_k=1, ldc=A * D, batch=A, a_batch=D, b_batch=D * D, c_batch=D, bias_batch=D, act=1), 2 * R * A * D * D),
    "gemm3 nt b11 6400x100x256": (mk(A=r1, B=w3, C=o3, bias=b2, M=R, N=O, K=D, a_i=A * D, a_k=1, b_j=D, b_k=1, ldc=A * O, batch=A, a_batch=D, b_batch=O * D, c_batch=O, bias_batch=O, act=2), 2 * R * A * O * D),
    "dx2 nn b11 6400x256x256": (mk(A=r1, B=w2, C=r2, M=R, N=D, K=D, a_i=A * D, a_k=1, b_j=1, b_k=D, ldc=A * D, batch=A, a_batch=D, b_batch=D * D, c_batch=D), 2 * R * A * D * D),
    "dx1 nn 6400x128x2816": (mk(A=r1, B=w1, C=dxh, M=R, N=H, K=A * D, a_i=A * D, a_k=1, b_j=1, b_k=H, ldc=H, batch=1), 2 * R * A * D * H),
    "dw2 tn b11 256x256x6400": (mk(A=r1, B=r2, C=dw2, M=D, N=D, K=R, a_i=1, a_k=A * D, b_j=1, b_k=A * D, ldc=D, batch=A, a_batch=D, b_batch=D, c_batch=D * D, splitk_ws=slab, splitk_ws_floats=slab.numel(), colsum=db, colsum_batch=D), 2 * R * A * D * D),
    "dw1 tn 2816x128x6400": (mk(A=r1, B=x, C=dw1, M=A * D, N=H, K=R, a_i=1, a_k=A * D, b_j=1, b_k=H, ldc=H, batch=1, splitk_ws=slab, splitk_ws_floats=slab.numel(), colsum=db, colsum_batch=0), 2 * R * A * D * H),
}
st = _lib.stream_ptr()
for name, (g, flops) in cases.items():
    if only and only not in name:
        continue
    for _ in range(3):
        _lib.check(L.as_gemm_f32(C.byref(g), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.as_gemm_f32(C.byref(g), st)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    print(f"{name:28s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
