"""What slows the GRU backward recurrence down beside other work?  The kernel (64 workgroups, latency-bound) is timed
alone, beside a stream of fp32-MFMA GEMMs (matrix load on the other CUs), and beside a stream of memory-bound row kernels
(HBM load, no matrix work).  usage: python tools/check_corun.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
B, T, H = 32, 200, 128
w_hh = torch.randn(2, 3 * H, H, device=dev) * 0.05
lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
y = torch.randn(B, T, 2 * H, device=dev)
gates = torch.rand(B, T, 2, 4, H, device=dev)
dy = torch.randn(B, T, 2 * H, device=dev)
dgi = torch.empty(B * T, 2, 3 * H, device=dev)
dgh = torch.empty(B * T, 2, 3 * H, device=dev)
side = torch.cuda.Stream(device=dev)

# background work 1: the weight-gradient GEMM of head layer 2 (11 x 256 x 256 under K = 6400), fp32 MFMA on every CU
A_ = torch.randn(6400, 11 * 256, device=dev)
B_ = torch.randn(6400, 11 * 256, device=dev)
C_ = torch.empty(11, 256, 256, device=dev)
slab = torch.empty(8 << 20, device=dev)
g = _lib.Gemm()
g.A, g.B, g.C = A_.data_ptr(), B_.data_ptr(), C_.data_ptr()
g.M, g.N, g.K = 256, 256, 6400
g.a_i, g.a_k, g.b_j, g.b_k, g.ldc = 1, 11 * 256, 1, 11 * 256, 256
g.batch, g.a_batch, g.b_batch, g.c_batch = 11, 256, 256, 256 * 256
g.splitk_ws, g.splitk_ws_floats = slab.data_ptr(), slab.numel()
# background work 2: LayerNorm-style normalisation of 70 400 rows of 256 (memory-bound, no matrix instructions)
x = torch.randn(70400, 256, device=dev)
xh = torch.empty_like(x)
rstd = torch.empty(70400, device=dev)


def bwd(stream):
    return L.as_gru_bidir_bwd(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(gates), _lib.ptr(w_hh), _lib.ptr(lengths), B, T, H, _lib.ptr(dgi),
                              _lib.ptr(dgh), stream)


def background(kind, n):
    for _ in range(n):
        if kind == "gemm":
            _lib.check(L.as_gemm_f32(C.byref(g), side.cuda_stream))
        elif kind == "rows":
            _lib.check(L.as_layernorm_fwd(_lib.ptr(x), None, None, None, _lib.ptr(xh), None, _lib.ptr(rstd), 70400, 256, 0, side.cuda_stream))


stamps = torch.zeros(2 * 2 * B * 4, dtype=torch.int64, device=dev)   # in-kernel stamps: cycles and wall ticks per workgroup (two halves)
L.as_gru_debug_stamps(_lib.ptr(stamps))
main = torch.cuda.current_stream().cuda_stream
for kind in ("alone", "gemm", "rows"):
    for _ in range(2):
        _lib.check(bwd(main))
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        background(kind, 12 if kind == "gemm" else 60)  # keeps the side stream busy for ~1.5 ms: the whole recurrence
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        bwd(main)
        e1.record()
        torch.cuda.synchronize()
        times.append(1e3 * e0.elapsed_time(e1))
    st = stamps.cpu().numpy().reshape(2, -1, 4)
    st = st[0] if st[0][:, 3].max() >= st[1][:, 3].max() else st[1]   # the half the LAST launch wrote (latest wall start)
    cyc, ticks = st[:, 0].astype(float), st[:, 1].astype(float)
    clock = float((cyc / ticks * 100).mean())   # MHz: shader cycles per 100 MHz wall tick, averaged over the 64 workgroups
    print(f"gru backward {kind:6s}: {min(times):7.1f} us/launch (min of 5), {sorted(times)[2]:7.1f} median | last launch, per workgroup: "
          f"{ticks.mean() / 100:6.1f} us inside the kernel, {cyc.mean() / T:6.0f} shader cycles per recurrent step, in-kernel clock {clock:5.0f} MHz",
          flush=True)
