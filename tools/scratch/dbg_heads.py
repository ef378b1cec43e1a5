import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from artspeech_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
rows, A, H, N = 1280, 2, 128, 50
dims = _lib.Dims(45, A, 64, H, N, 0)
lay = _lib.layout(dims)
torch.manual_seed(0)
P = torch.randn(lay.total, device=dev) * 0.05
x = torch.relu(torch.randn(rows, H, device=dev))
dout = torch.randn(rows, A, 2, N, device=dev) * 1e-3
nws = L.as_head_workspace_floats(C.byref(dims), rows)
st = _lib.stream_ptr()
def run(mode, off, fill):
    L.as_set_matrix_arith(mode)
    big = torch.full((nws + 4096,), fill, device=dev)
    ws = big[off:off + nws]
    out = torch.empty(rows, A, 2, N, device=dev)
    dx = torch.empty(rows, H, device=dev)
    G = torch.zeros_like(P)
    _lib.check(L.as_head_fwd(C.byref(dims), C.byref(lay), _lib.ptr(P), _lib.ptr(x), rows, _lib.ptr(out), _lib.ptr(ws), 1, st))
    _lib.check(L.as_head_bwd(C.byref(dims), C.byref(lay), _lib.ptr(P), _lib.ptr(out), _lib.ptr(dout), rows, _lib.ptr(dx), _lib.ptr(G), _lib.ptr(ws), st))
    torch.cuda.synchronize()
    return out.cpu(), dx.cpu(), G.cpu()
ref = run(0, 0, 0.0)
for off, fill in ((0, 0.0), (0, float('nan')), (64, 0.0), (64, 1e30), (1024, float('nan'))):
    o, dx, G = run(1, off, fill)
    print(off, fill, "out %.2e dx %.2e G %.2e" % ((o - ref[0]).abs().max(), (dx - ref[1]).abs().max() / ref[1].abs().max(), (G - ref[2]).abs().max() / ref[2].abs().max()),
          "nan:", bool(torch.isnan(dx).any()), bool(torch.isnan(G).any()))
