"""The grouped GEMM shapes of one interaction group of the transformer decoder (G = 110 blocks, R = 6400 rows, d = 256) alone,
plain and with the extended operands (as_gemm.res / .mask / .k_seg): HIP-event time per launch and TFLOP/s.
usage: python tools/bench_gemm_ext.py [iters]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
L = _lib.lib()
G, R, d, A_ = 110, 6400, 256, 11
per = G // A_
torch.manual_seed(0)
x = torch.randn(G, R, d, device=dev)
y = torch.relu(torch.randn(G, R, d, device=dev))
w = torch.randn(G, d, d, device=dev) / 16
b = torch.randn(G, d, device=dev)
out = torch.empty(G, R, d, device=dev)
cat = torch.randn(A_, R, per * d, device=dev)
dx = torch.empty(A_, R, d, device=dev)
bits = torch.randint(-2**31, 2**31 - 1, (G, R, d // 32), dtype=torch.int32, device=dev)
tbl = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)  # noqa: E731
coff = tbl([c * R * per * d + j * d for c in range(A_) for j in range(per)])
aseg = tbl([g * R * d for g in range(G)])
bseg = tbl([g * d * d for g in range(G)])


def gemm(**kw):
    g = _lib.Gemm()
    g.batch = 1
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if torch.is_tensor(v) else v)
    _lib.check(L.as_gemm_f32(C.byref(g), _lib.stream_ptr()), "as_gemm_f32")


nt = dict(M=R, N=d, K=d, a_i=d, a_k=1, b_j=d, b_k=1, batch=G, a_batch=R * d, b_batch=d * d)
nn = dict(M=R, N=d, K=d, a_i=d, a_k=1, b_j=1, b_k=d, batch=G, a_batch=R * d, b_batch=d * d)
cases = {
    "forward, bias + relu (plain)": (lambda: gemm(A=x, B=w, C=out, bias=b, bias_batch=d, act=1, ldc=d, c_batch=R * d, **nt), G),
    "forward, bias -> concatenated (plain)": (lambda: gemm(A=x, B=w, C=cat, bias=b, bias_batch=d, ldc=per * d, c_off=coff, **nt), G),
    "forward, bias (plain) + torch add": (lambda: (gemm(A=x, B=w, C=out, bias=b, bias_batch=d, ldc=d, c_batch=R * d, **nt), out.add_(y)), G),
    "forward, bias + residual as the accumulators' start (ext)": (lambda: gemm(A=x, B=w, C=out, bias=b, bias_batch=d, ldc=d, c_batch=R * d, res=y,
                                                                                res_ld=d, res_batch=R * d, **nt), G),
    "forward, bias + residual start -> concatenated (ext)": (lambda: gemm(A=x, B=w, C=cat, bias=b, bias_batch=d, ldc=per * d, c_off=coff, res=y, res_ld=d, res_batch=R * d, **nt), G),
    "input gradient (plain)": (lambda: gemm(A=x, B=w, C=out, ldc=d, c_batch=R * d, **nn), G),
    "forward, bias + relu + bit image (ext)": (lambda: gemm(A=x, B=w, C=out, bias=b, bias_batch=d, act=1, ldc=d, c_batch=R * d, relu_bits=bits,
                                                            relu_bits_batch=R * (d // 32), **nt), G),
    "input gradient + relu mask (ext)": (lambda: gemm(A=x, B=w, C=out, ldc=d, c_batch=R * d, mask_bits=bits, mask_batch=R * (d // 32), **nn), G),
    "input gradient + residual + relu mask (ext)": (lambda: gemm(A=x, B=w, C=out, ldc=d, c_batch=R * d, mask_bits=bits, mask_batch=R * (d // 32),
                                                                    res=cat, res_ld=per * d, res_off=coff, **nn), G),
    "channel sums, segmented K = 10 d (ext)": (lambda: gemm(A=x, B=w, C=dx, M=R, N=d, K=per * d, a_i=d, a_k=1, b_j=1, b_k=d, ldc=d, batch=A_,
                                                              c_batch=R * d, k_seg=d, a_seg_off=aseg, b_seg_off=bseg), G),
}
# attention backward products of one interaction group: Z = G * B * heads slices of P^T [Tk][T] against [T][dh] operands
Tq, dh, Z = 200, 64, 110 * 32 * 4
pt = torch.triu(torch.rand(Tq, Tq, device=dev)).repeat(Z, 1, 1).contiguous()   # zero for q < key (causal)
xa = torch.randn(Z, Tq, dh, device=dev)
oa = torch.empty(Z, Tq, dh, device=dev)
att = dict(M=Tq, N=dh, K=Tq, b_j=1, b_k=dh, ldc=dh, batch=Z, a_batch=Tq * Tq, b_batch=Tq * dh, c_batch=Tq * dh)
cases.update({
    "attention bwd, P^T x (dV / dK), full": (lambda: gemm(A=pt, B=xa, C=oa, a_i=Tq, a_k=1, **att), None),
    "attention bwd, P^T x (dV / dK), k_tri = 1": (lambda: gemm(A=pt, B=xa, C=oa, a_i=Tq, a_k=1, k_tri=1, **att), None),
    "attention bwd, P x (dQ), full": (lambda: gemm(A=pt, B=xa, C=oa, a_i=1, a_k=Tq, **att), None),
    "attention bwd, P x (dQ), k_tri = 2": (lambda: gemm(A=pt, B=xa, C=oa, a_i=1, a_k=Tq, k_tri=2, **att), None),
})
print(f"G={G} R={R} d={d}: {2 * R * d * d * G / 1e9:.1f} GFLOP per launch; {iters} launches each")
for _ in range(30):   # clocks and caches settle before the first case is timed
    gemm(A=x, B=w, C=out, ldc=d, c_batch=R * d, **nn)
for name, (fn, blocks) in cases.items():
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    flops = 2 * R * d * d * blocks if blocks else 2.0 * Tq * Tq * dh * Z   # (attention rows: dense-equivalent FLOPs)
    print(f"{name:52s} {us:8.1f} us  {flops / us / 1e6:6.1f} TF/s", flush=True)
