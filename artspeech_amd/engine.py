"""Training-step engine: the whole forward + loss + backward (+ gradient all-reduce + Adam) of the
model-free path as a fixed sequence of C-ABI calls on persistent device buffers.

This is the host side of ``run_epoch``'s per-batch body (reference train_phoneme_to_articulation.py:
80-96) without autograd bookkeeping: every buffer (workspace, outputs, gradients, optimizer moments)
is allocated once for a (B, T) shape, so a step is five library calls and no allocation -- suitable for
HIP-graph capture.  The drop-in ``nn.Module`` / autograd path (models.py, metrics.py) runs the very
same kernels and is what the parity tests use; this engine is what the benchmark and the DP trainer use.
"""
import ctypes as C
import os

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
        # gradient all-reduce in two pieces: trunk + heads (74 % of the parameters, final before the GRU backward
        # recurrences start) on a communication stream beside those recurrences, the rest after them
        # (RCCL only: gloo's asynchronous path on device tensors is pathologically slow -- rehearsals use one all-reduce)
        self.ar_overlap = (self.use_dist and os.environ.get("ARTSPEECH_NO_AR_OVERLAP") is None
                           and torch.distributed.get_backend(process_group) == "nccl")
        self.head_off = int(_lib.layout(self.dims).lin_w)
        self.comm_stream = torch.cuda.Stream(device=model.flat.device) if self.ar_overlap else None
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
        self.bwd_opts = _lib.Opts(0.0, 0, 1)

    def forward_backward(self, tokens, lengths_dev, targets, loss_scale):
        """tokens (B, >=T) int64, lengths_dev (B,) int32 on device, targets (B, >=T, A, 2, N).
        loss_scale = 1 / (N_valid_global * A * N).  Leaves loss in self.loss, gradients in self.grads."""
        L, d, st = _lib.lib(), self.dims, _lib.stream_ptr()
        P = self.model.flat.data
        B, T = self.B, self.T
        _lib.check(L.as_artspeech_fwd(C.byref(d), _lib.ptr(P), _lib.ptr(tokens), tokens.stride(0), _lib.ptr(lengths_dev),
                                      B, T, _lib.ptr(self.out), _lib.ptr(self.ws), 1, None, st), "as_artspeech_fwd")
        # criterion backward and the model's final sigmoid backward in one pass: dout holds d(loss)/d(pre-sigmoid)
        _lib.check(L.as_euclid_masked_fwd_bwd_presigmoid(_lib.ptr(self.out), _lib.ptr(targets), targets.shape[1],
                                                         _lib.ptr(lengths_dev), B, T, d.n_art, d.n_samp, float(loss_scale),
                                                         _lib.ptr(self.loss), _lib.ptr(self.dout), _lib.ptr(self.partial), st),
                   "as_euclid_masked_fwd_bwd_presigmoid")
        _lib.check(L.as_artspeech_bwd(C.byref(d), _lib.ptr(P), _lib.ptr(tokens), tokens.stride(0), _lib.ptr(lengths_dev),
                                      B, T, _lib.ptr(self.out), _lib.ptr(self.dout), _lib.ptr(self.grads), _lib.ptr(self.ws),
                                      C.byref(self.bwd_opts), st), "as_artspeech_bwd")

    def bad_tokens(self):
        """Number of token ids outside [0, V) in the last batch (device word written by as_artspeech_fwd; the kernels clamp
        such ids).  Reading it synchronises: call it where the loss is read."""
        return int(self.ws[:1].view(torch.int32).item())

    def loss_value(self):
        """float(loss) of the last step; raises like nn.Embedding if that batch held out-of-range token ids."""
        v = float(self.loss)
        n = self.bad_tokens()
        if n:
            raise IndexError(f"index out of range in self ({n} token ids outside [0, {self.dims.vocab}))")
        return v

    def all_reduce(self):
        """RCCL all-reduce (SUM) of the flat gradient buffer: shard losses are scaled by the GLOBAL valid-frame count, so the
        sum over ranks is the reference's full-batch gradient.  With ``ar_overlap`` the tail of the buffer (trunk Linear +
        heads) is reduced on ``comm_stream`` as soon as ``as_artspeech_bwd`` has produced it -- while the GRU backward is
        still running on the compute streams -- and the head of the buffer after the whole backward; the caller's stream
        continues (Adam) only after both."""
        if not self.use_dist:
            return
        dist = torch.distributed
        if not self.ar_overlap:
            dist.all_reduce(self.grads, op=dist.ReduceOp.SUM, group=self.pg)
            return
        _lib.check(_lib.lib().as_artspeech_wait_head_grads(_lib.stream_ptr(), self.comm_stream.cuda_stream), "as_artspeech_wait_head_grads")
        with torch.cuda.stream(self.comm_stream):
            tail = dist.all_reduce(self.grads[self.head_off:], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        head = dist.all_reduce(self.grads[:self.head_off], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        tail.wait()
        head.wait()
        torch.cuda.current_stream().wait_stream(self.comm_stream)

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
