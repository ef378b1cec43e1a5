"""Weight-gradient GEMM shapes of the BiGRU training step (and the transformer's grouped 256 x 256 x 6400 one) through
as_gemm_f32, alone on the chip: microseconds, TFLOP/s, fraction of the fp32 MFMA peak, and a check against fp64.
AS_NO_WGRAD=1 in the environment routes the same calls to the general kernel (the round-1 path) for comparison.
usage: python tools/bench_wgrad.py [iters] [cu_budget]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cu_budget = int(sys.argv[2]) if len(sys.argv) > 2 else 0
R, T = 6400, 200
ws = torch.empty(16 << 20, device=dev)
torch.manual_seed(0)

# name: (M, N, batch, lda, a_batch, ldb, b_batch, kshift, kT, kshift_batch)
SHAPES = {
    "headb.dw3  100x256 x11": (100, 256, 11, 1100, 100, 2816, 256, 0, 0, 0),
    "headb.dw2  256x256 x11": (256, 256, 11, 2816, 256, 2816, 256, 0, 0, 0),
    "headb.dw1 2816x128": (2816, 128, 1, 2816, 0, 128, 0, 0, 0, 0),
    "trunkb.dw  128x256": (128, 256, 1, 128, 0, 256, 0, 0, 0, 0),
    "grub.dw_ih1 768x256": (768, 256, 1, 768, 0, 256, 0, 0, 0, 0),
    "grub.dw_hh 384x128 x2 (shift)": (384, 128, 2, 768, 384, 256, 128, -1, T, 2),
    "transformer dW 256x256 x110": (256, 256, 110, 110 * 256, 256, 110 * 256, 256, 0, 0, 0),
    # the layout the transformer's modules really use: block-major operands [G][R][256]
    "transformer dW x110 block-major": (256, 256, 110, 256, R * 256, 256, R * 256, 0, 0, 0),
    "transformer dW x11 block-major": (256, 256, 11, 256, R * 256, 256, R * 256, 0, 0, 0),
    "transformer dW x1": (256, 256, 1, 256, 0, 256, 0, 0, 0, 0),
}
PREC = int(os.environ.get("AS_BENCH_PRECISION", "0"))   # 3: as_gemm.precision = 3 (gemm_s6.hip's kernel where it takes the shape)

label = "general kernel (AS_NO_WGRAD)" if os.environ.get("AS_NO_WGRAD") else "wgrad_f32_kernel"
print(f"--- {label}, K = {R}, cu_budget = {cu_budget or 'chip'}, {iters} launches each")
total_us = 0.0
for name, (M, N, batch, lda, ab, ldb, bb, ksh, kT, kshb) in SHAPES.items():
    block_major = ab >= R * M
    A = torch.randn(batch if block_major else 1, R, lda, device=dev)
    Bm = torch.randn(batch if block_major else 1, R, ldb, device=dev)
    Cm = torch.empty(batch, M, N, device=dev)
    cs = torch.empty(batch, M, device=dev)
    g = _lib.Gemm()
    g.A, g.B, g.C = A.data_ptr(), Bm.data_ptr(), Cm.data_ptr()
    g.M, g.N, g.K = M, N, R
    g.a_i, g.a_k, g.b_j, g.b_k, g.ldc = 1, lda, 1, ldb, N
    g.batch, g.a_batch, g.b_batch, g.c_batch = batch, ab, bb, M * N
    g.b_kshift, g.b_kT, g.b_kshift_batch = ksh, kT, kshb
    g.splitk_ws, g.splitk_ws_floats = ws.data_ptr(), ws.numel()
    g.colsum, g.colsum_batch = cs.data_ptr(), M
    g.cu_budget = cu_budget
    g.precision = PREC
    st = _lib.stream_ptr()
    for _ in range(3):
        _lib.check(L.as_gemm_f32(C.byref(g), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.as_gemm_f32(C.byref(g), st)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    flops = 2.0 * R * M * N * batch
    # check against fp64 on a few batch members
    err = 0.0
    for b in sorted({0, batch // 2, batch - 1}):
        a64 = (A[b] if block_major else A[0][:, b * ab:b * ab + M]).double()
        b64 = (Bm[b] if block_major else Bm[0][:, b * bb:b * bb + N]).double()
        if kT:
            s = ksh + b * kshb
            v = b64.reshape(R // kT, kT, N)
            sh = torch.zeros_like(v)
            if s < 0:
                sh[:, -s:] = v[:, :s]
            else:
                sh[:, :kT - s] = v[:, s:]
            b64 = sh.reshape(R, N)
        ref = a64.T @ b64
        err = max(err, float((Cm[b].double() - ref).abs().max() / ref.abs().max()))
        err = max(err, float((cs[b].double() - a64.sum(0)).abs().max() / a64.sum(0).abs().max()))
    if "transformer" not in name:
        total_us += us * (2 if "dw_hh" in name else 1)
    print(f"{name:32s} {us:8.1f} us  {flops / us / 1e6:6.1f} TFLOP/s  {flops / us / 1e6 / 157.3:5.2f} of peak   rel err {err:.1e}", flush=True)
    del A, Bm, Cm
print(f"sum over one training step's weight gradients: {total_us:.1f} us (22.9 GFLOP -> {22.9e3 / total_us:.1f} TFLOP/s)")
