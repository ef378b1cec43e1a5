"""Training-step engine: the whole forward + loss + backward (+ gradient all-reduce + Adam) of the
model-free path as a fixed sequence of C-ABI calls on persistent device buffers.

This is the host side of ``run_epoch``'s per-batch body (reference train_phoneme_to_articulation.py:
80-96) without autograd bookkeeping: every buffer (workspace, outputs, gradients, optimizer moments)
is allocated once for a (B, T) shape, so a step is five library calls and no allocation -- suitable for
HIP-graph capture.  The drop-in ``nn.Module`` / autograd path (models.py, metrics.py) runs the very
same kernels and is what the parity tests use; this engine is what the benchmark and the DP trainer use.
"""
import ctypes as C

import torch

from . import _lib


class TrainStep:
    def __init__(self, model, B, T, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, process_group=None,
                 optimizer=True):
        self.model = model
        self.dims = model.dims
        self.B, self.T = B, T
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.pg = process_group
        self.use_dist = process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                                      and torch.distributed.get_world_size() > 1)
        self.optimizer = optimizer
        L = _lib.lib()
        dev = model.flat.device
        _lib.require_gpu(model.flat, "model parameters")
        d = self.dims
        self.ws = torch.empty(L.as_artspeech_workspace_floats(C.byref(d), B, T), dtype=torch.float32, device=dev)
        self.out = torch.empty((B, T, d.n_art, 2, d.n_samp), dtype=torch.float32, device=dev)
        self.dout = torch.empty_like(self.out)
        self.grads = torch.zeros_like(model.flat.data)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.partial = torch.empty(L.as_euclid_masked_partials(), dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.grads)
        self.exp_avg_sq = torch.zeros_like(self.grads)
        self.steps = 0

    def forward_backward(self, tokens, lengths_dev, targets, loss_scale):
        """tokens (B, >=T) int64, lengths_dev (B,) int32 on device, targets (B, >=T, A, 2, N).
        loss_scale = 1 / (N_valid_global * A * N).  Leaves loss in self.loss, gradients in self.grads."""
        L, d, st = _lib.lib(), self.dims, _lib.stream_ptr()
        P = self.model.flat.data
        B, T = self.B, self.T
        _lib.check(L.as_artspeech_fwd(C.byref(d), _lib.ptr(P), _lib.ptr(tokens), tokens.stride(0), _lib.ptr(lengths_dev),
                                      B, T, _lib.ptr(self.out), _lib.ptr(self.ws), 1, None, st), "as_artspeech_fwd")
        _lib.check(L.as_euclid_masked_fwd_bwd(_lib.ptr(self.out), _lib.ptr(targets), targets.shape[1], _lib.ptr(lengths_dev),
                                              B, T, d.n_art, d.n_samp, float(loss_scale), _lib.ptr(self.loss),
                                              _lib.ptr(self.dout), _lib.ptr(self.partial), st), "as_euclid_masked_fwd_bwd")
        _lib.check(L.as_artspeech_bwd(C.byref(d), _lib.ptr(P), _lib.ptr(tokens), tokens.stride(0), _lib.ptr(lengths_dev),
                                      B, T, _lib.ptr(self.out), _lib.ptr(self.dout), _lib.ptr(self.grads), _lib.ptr(self.ws),
                                      None, st), "as_artspeech_bwd")

    def all_reduce(self):
        """One RCCL all-reduce (SUM) of the flat gradient buffer: shard losses are scaled by the GLOBAL
        valid-frame count, so the sum over ranks is the reference's full-batch gradient."""
        if self.use_dist:
            torch.distributed.all_reduce(self.grads, op=torch.distributed.ReduceOp.SUM, group=self.pg)

    def adam(self):
        L = _lib.lib()
        self.steps += 1
        _lib.check(L.as_adam_step(_lib.ptr(self.model.flat.data), _lib.ptr(self.grads), _lib.ptr(self.exp_avg),
                                  _lib.ptr(self.exp_avg_sq), self.grads.numel(), self.lr, self.betas[0], self.betas[1],
                                  self.eps, self.weight_decay, self.steps, 1.0, _lib.stream_ptr()), "as_adam_step")

    def step(self, tokens, lengths_dev, targets, loss_scale):
        self.forward_backward(tokens, lengths_dev, targets, loss_scale)
        self.all_reduce()
        if self.optimizer:
            self.adam()
        return self.loss
