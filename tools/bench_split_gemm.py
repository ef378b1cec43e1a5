"""Probe: Linear 2 of the heads (11 heads x 6400 frames x 256 -> 256) as an fp32 GEMM on the bf16 matrix instruction with
operands split into three bfloat16 planes (tools/split_gemm_probe.hip), against the library's fp32-MFMA kernel: HIP-event time
per launch and the error of both against an fp64 product of the same fp32 operands.
usage: python tools/bench_split_gemm.py [iters]     (the probe library is built by hipcc on first use)"""
import ctypes as C
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from artspeech_amd import _lib  # noqa: E402

SO = os.path.join(HERE, "_build", "libsplit_probe.so")
SRC = os.path.join(HERE, "split_gemm_probe.hip")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", SRC, "-o", SO])
P = C.CDLL(SO)
P.split_gemm_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
P.split_gemm_probe.restype = C.c_int

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
L = _lib.lib()


def split(x):
    """x = hi + mid + lo exactly (three bfloat16 planes, round to nearest even at every level)"""
    hi = x.bfloat16()
    r = x - hi.float()
    mid = r.bfloat16()
    lo = (r - mid.float()).bfloat16()
    return torch.stack([hi, mid, lo])


def tiled(p):
    """[3][Z][rows][K] -> [3][Z][K / 16][rows][16]"""
    three, Z, R, K = p.shape
    return p.view(three, Z, R, K // 16, 16).permute(0, 1, 3, 2, 4).contiguous()


def run(M, K, Z, label):
    torch.manual_seed(0)
    # operands shaped like the layer's: normalised activations, weights ~ U(+-1/sqrt(K))
    x = torch.randn(Z, M, K, device=dev)
    w = (torch.rand(Z, 256, K, device=dev) * 2 - 1) / K ** 0.5
    ref = torch.bmm(x.double(), w.double().transpose(1, 2))
    scale = ref.abs().max().item()
    xs, ws = split(x), split(w)
    assert torch.equal(xs.float().sum(0), x) and torch.equal(ws.float().sum(0), w), "the split is exact"
    xp, wp = tiled(xs), tiled(ws)
    out = torch.empty(Z, M, 256, device=dev)

    def fp32():
        g = _lib.Gemm()
        for k, v in dict(A=x, B=w, C=out, M=M, N=256, K=K, a_i=K, a_k=1, b_j=K, b_k=1, batch=Z, a_batch=M * K, b_batch=256 * K, ldc=256,
                         c_batch=M * 256).items():
            setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
        _lib.check(L.as_gemm_f32(C.byref(g), _lib.stream_ptr()), "as_gemm_f32")

    def probe(n):
        rc = P.split_gemm_probe(xp.data_ptr(), wp.data_ptr(), out.data_ptr(), M, K, Z, n, _lib.stream_ptr())
        assert rc == 0, rc

    flop = 2.0 * M * 256 * K * Z
    print(f"{label}: Z={Z} M={M} K={K}  {flop / 1e9:.2f} GFLOP")
    for name, fn in [("fp32 MFMA (as_gemm_f32)", fp32), ("bf16 planes, 9 products", lambda: probe(9)), ("bf16 planes, 6 products", lambda: probe(6)),
                     ("bf16 planes, 3 products", lambda: probe(3))]:
        out.zero_()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        err = (out.double() - ref)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        print(f"  {name:28s} {us:8.1f} us  {flop / us / 1e6:7.1f} TF/s   max|err|/max|C| {err.abs().max().item() / scale:.2e}   "
              f"rms err / rms C {(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item():.2e}", flush=True)


run(6400, 256, 11, "Linear 2 of the heads")
run(6528, 256, 10, "whole rounds (510 tiles)")
run(6400, 256, 110, "one interaction group of the transformer")
