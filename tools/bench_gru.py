"""Micro-benchmark of the GRU recurrence kernels alone (C ABI), optionally against an alternative
library build (ARTSPEECH_LIB=path) for ablation studies.  usage: python tools/bench_gru.py [iters]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402

if os.environ.get("ARTSPEECH_LIB"):
    _lib.LIB_PATH = os.environ["ARTSPEECH_LIB"]
L = _lib.lib()
dev = torch.device("cuda:0")
B, T, H = 32, 200, 128
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
gi = torch.randn(B * T, 2, 3 * H, device=dev)
w_hh = torch.randn(2, 3 * H, H, device=dev) * 0.05
b_hh = torch.randn(2, 3 * H, device=dev) * 0.05
lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
y = torch.empty(B, T, 2 * H, device=dev)
gates = torch.empty(B, T, 2, 4, H, device=dev)
dy = torch.randn(B, T, 2 * H, device=dev)
dgi = torch.empty(B * T, 2, 3 * H, device=dev)
dgh = torch.empty(B * T, 2, 3 * H, device=dev)
st = _lib.stream_ptr()


def fwd():
    return L.as_gru_bidir_fwd(_lib.ptr(gi), None, 0, _lib.ptr(w_hh), _lib.ptr(b_hh), _lib.ptr(lengths), B, T, H, _lib.ptr(y),
                              _lib.ptr(gates), st)


def bwd():
    return L.as_gru_bidir_bwd(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(gates), _lib.ptr(w_hh), _lib.ptr(lengths), B, T, H,
                              _lib.ptr(dgi), _lib.ptr(dgh), st)


for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(3):
        _lib.check(fn())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    print(f"{os.environ.get('ARTSPEECH_LIB', 'default')[-24:]:24s} gru {name}: {us:8.1f} us/launch  {us / T * 1e3:7.1f} ns/step", flush=True)


# ---- LSTM cell (RNNType.LSTM): same shapes, 4 gate planes + cell state
gi4 = torch.randn(B * T, 2, 4 * H, device=dev)
w4 = torch.randn(2, 4 * H, H, device=dev) * 0.05
b4 = torch.randn(2, 4 * H, device=dev) * 0.05
gates5 = torch.empty(B, T, 2, 5, H, device=dev)
dg4 = torch.empty(B * T, 2, 4 * H, device=dev)


def lfwd():
    return L.as_lstm_bidir_fwd(_lib.ptr(gi4), None, 0, _lib.ptr(w4), _lib.ptr(b4), _lib.ptr(lengths), B, T, H, _lib.ptr(y), _lib.ptr(gates5), st)


def lbwd():
    return L.as_lstm_bidir_bwd(_lib.ptr(dy), _lib.ptr(gates5), _lib.ptr(w4), _lib.ptr(lengths), B, T, H, _lib.ptr(dg4), st)


for name, fn in (("fwd", lfwd), ("bwd", lbwd)):
    for _ in range(3):
        _lib.check(fn())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    print(f"{'default':24s} lstm {name}: {us:8.1f} us/launch  {us / T * 1e3:7.1f} ns/step", flush=True)
