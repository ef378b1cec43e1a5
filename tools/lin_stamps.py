"""In-kernel cycle stamps of one fused head-layer launch (head GEMM 2: 1100 workgroups of 64 frames x 256 features, K = 256):
when each workgroup started, how long its prologue, main loop and epilogue took, and which CU it ran on.
usage: python tools/lin_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
rows, A, H, N = 6400, 11, 128, 50
dims = _lib.Dims(45, A, 64, H, N, 0)
lay = _lib.layout(dims)
torch.manual_seed(0)
P = torch.randn(lay.total, device=dev) * 0.05
x = torch.relu(torch.randn(rows, H, device=dev))
out = torch.empty(rows, A, 2, N, device=dev)
ws = torch.empty(L.as_head_workspace_floats(C.byref(dims), rows), device=dev)
st = _lib.stream_ptr()
nwg = 1400
for _ in range(3):
    _lib.check(L.as_head_fwd(C.byref(dims), C.byref(lay), _lib.ptr(P), _lib.ptr(x), rows, _lib.ptr(out), _lib.ptr(ws), 1, st))
stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
L.as_lin_debug_stamps(_lib.ptr(stamps), nwg)
_lib.check(L.as_head_fwd(C.byref(dims), C.byref(lay), _lib.ptr(P), _lib.ptr(x), rows, _lib.ptr(out), _lib.ptr(ws), 1, st))
torch.cuda.synchronize()
L.as_lin_debug_stamps(None, 0)
# the buffer holds the LAST stamped launch that wrote each slot: gemm1 then gemm2 (same grid) -> gemm2's stamps
s = stamps.cpu().numpy().reshape(nwg, 8).astype(np.int64)
pro, loop, epi, tot = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 3] - s[:, 0]
r0 = s[:, 4].min()
r0 = s[s[:, 3] > 0, 4].min()
start_us, end_us = (s[:, 4] - r0) / 100.0, (s[:, 6] - r0) / 100.0     # s_memrealtime: 100 MHz, one clock for the chip
print("per workgroup, shader cycles (s_memtime differences) and wall microseconds (s_memrealtime), launch order")
live = s[:, 3] > 0
nwg = int(live.sum())
print('workgroups stamped:', nwg)
for lo in range(0, nwg, 256):
    hi = min(lo + 256, nwg)
    sl = slice(lo, hi)
    print(f"wg {lo:4d}-{hi - 1:4d}: start {np.median(start_us[sl]):6.1f} us  end {np.median(end_us[sl]):6.1f} us | prologue {np.median(pro[sl]):7.0f}  "
          f"loop {np.median(loop[sl]):8.0f}  epilogue {np.median(epi[sl]):7.0f}  total {np.median(tot[sl]):8.0f} cycles")
print(f"launch: {end_us.max():.1f} us from first start to last end; clock = {np.median(tot / np.maximum(end_us - start_us, 1e-3)):.0f} cycles/us")
# how many workgroups are in flight over time
start_us, end_us = start_us[live], end_us[live]
ts = np.linspace(0, end_us.max(), 25)
print("in flight:", [int(((start_us <= t) & (end_us > t)).sum()) for t in ts])
