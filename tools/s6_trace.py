"""Per-tile cycle stamps inside lin_s6_kernel's main loop (diagnostic build with -DAS_S6_TRACE): for the waves of a few
workgroups of head Linear 2, when each k-tile's loads were issued (0), its first fragments were in registers (1), the first and
second k-step's matrix instructions were issued (2, 3), the split + LDS stores were done (4) and the barrier was passed (5).
usage: ARTSPEECH_DIAG_LIB=artspeech_amd/libartspeech_hip_diag_trace.so python tools/s6_trace.py"""
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
nwg = 4096
for _ in range(3):
    _lib.check(L.as_head_fwd(C.byref(dims), C.byref(lay), _lib.ptr(P), _lib.ptr(x), rows, _lib.ptr(out), _lib.ptr(ws), 1, st))
stamps = torch.zeros(nwg * 8 + nwg * 8 * 64, dtype=torch.int64, device=dev)
L.as_lin_debug_stamps(_lib.ptr(stamps), 1200)
_lib.check(L.as_head_fwd(C.byref(dims), C.byref(lay), _lib.ptr(P), _lib.ptr(x), rows, _lib.ptr(out), _lib.ptr(ws), 1, st))
torch.cuda.synchronize()
L.as_lin_debug_stamps(None, 0)
s = stamps.cpu().numpy()
head = s[:nwg * 8].reshape(nwg, 8)
tr = s[nwg * 8:].reshape(nwg, 8, 64)[:, :, :60].reshape(nwg, 8, 10, 6)
names = ["loads issued", "frags in regs", "k-step 0 issued", "k-step 1 issued", "split+stores", "barrier passed"]
for wg in (0, 1, 256, 257, 600):
    t0 = head[wg, 0]
    print(f"workgroup {wg}: start {t0}, prologue {head[wg, 1] - t0}, loop end {head[wg, 2] - t0}, end {head[wg, 3] - t0}")
    for w in (0, 4):
        print(f"  wave {w}: per tile, cycles since the tile's first stamp: " + ", ".join(names[1:]))
        for kt in range(8):
            r = tr[wg, w, kt]
            print(f"    tile {kt}: at {r[0] - t0:7d}  " + "  ".join(f"{int(v - r[0]):6d}" for v in r[1:]))
# averages over all stamped workgroups and waves
d = (tr[:1100, :, 1:8, 1:] - tr[:1100, :, 1:8, :1]).reshape(-1, 5)
d = d[(d > 0).all(1) & (d < 100000).all(1)]
print("mean over tiles 1-7 of all waves:", dict(zip(names[1:], d.mean(0).round().astype(int).tolist())))
per = (tr[:1100, :, 2:8, 0] - tr[:1100, :, 1:7, 0]).reshape(-1)
per = per[(per > 0) & (per < 100000)]
print("mean cycles per k-tile:", per.mean().round())
