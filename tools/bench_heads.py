"""The ArticulatorPredictor heads alone (as_head_fwd + as_head_bwd, A = 11, 6400 frames): per-phase microseconds from the
library's own HIP-event table.  AS_NO_LIN=1 = general GEMMs + row kernels (round-1 path); AS_LIN_ABL=1/2 ablations.
usage: python tools/bench_heads.py [iters]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rows, A, H, N = 6400, 11, 128, 50
dims = _lib.Dims(45, A, 64, H, N, 0)
lay = _lib.layout(dims)
torch.manual_seed(0)
P = torch.randn(lay.total, device=dev) * 0.05
for off, n in ((lay.ln1_g, A * H), (lay.ln2_g, A * 256), (lay.ln3_g, A * 256)):
    P[off:off + n] = 1.0 + 0.1 * torch.randn(n, device=dev)
x = torch.relu(torch.randn(rows, H, device=dev))
out = torch.empty(rows, A, 2, N, device=dev)
dout = torch.randn_like(out) * 1e-3
dx = torch.empty(rows, H, device=dev)
G = torch.zeros_like(P)
ws = torch.empty(L.as_head_workspace_floats(C.byref(dims), rows), device=dev)
st = _lib.stream_ptr()


def step():
    _lib.check(L.as_head_fwd(C.byref(dims), C.byref(lay), _lib.ptr(P), _lib.ptr(x), rows, _lib.ptr(out), _lib.ptr(ws), 1, st))
    _lib.check(L.as_head_bwd(C.byref(dims), C.byref(lay), _lib.ptr(P), _lib.ptr(out), _lib.ptr(dout), rows, _lib.ptr(dx), _lib.ptr(G),
                             _lib.ptr(ws), st))


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    step()
e1.record()
torch.cuda.synchronize()
print(f"heads fwd + bwd (no phase events): {1e3 * e0.elapsed_time(e1) / iters:.1f} us per pass")
L.as_profile_reset()
L.as_profile_enable(1)
for _ in range(iters):
    step()
torch.cuda.synchronize()
L.as_profile_enable(0)
buf = C.create_string_buffer(1 << 16)
L.as_profile_report(buf, len(buf))
tot = 0.0
for line in buf.value.decode().splitlines():
    name, cnt, ms = line.split()
    us = 1e3 * float(ms) / iters
    tot += us
    print(f"  {name:16s} {us:8.1f} us")
print(f"  {'sum':16s} {tot:8.1f} us   finite: {bool(torch.isfinite(out).all())}")
