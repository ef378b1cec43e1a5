"""Training-step engine: the whole forward + loss + backward (+ gradient all-reduce + Adam) of the
model-free path as a fixed sequence of C-ABI calls on persistent device buffers.

This is the host side of ``run_epoch``'s per-batch body (reference train_phoneme_to_articulation.py:
80-96) without autograd bookkeeping: every buffer (workspace, outputs, gradients, optimizer moments)
is allocated once for a (B, T) shape, so a step is five library calls and no allocation -- suitable for
HIP-graph capture.  The drop-in ``nn.Module`` / autograd path (models.py, metrics.py) runs the very
same kernels and is what the parity tests use; this engine is what the benchmark and the DP trainer use.

Software pipelining (``pipeline=True``).  The forward recurrences keep 64 of the 256 CUs busy for ~200 us with nothing
beside them, while the backward has more weight-gradient GEMM work than fits beside ITS recurrences.  So the weight
gradient of the heads' second Linear (9.2 GFLOP, its slab reduce and LayerNorm unfold), that slice's all-reduce and its
Adam update are issued one step late, on a side stream beside the NEXT step's forward recurrences; the forward folds the
head weights only after that update has landed (``as_opts.fold_wait_event``).  Every parameter still receives exactly the
update of the unpipelined loop -- same gradients, same step count, same order of operations per element -- so the
parameters after ``flush()`` are bit-identical; only the schedule differs.  ``flush()`` applies the pending slice at once
(call it before reading parameters, checkpointing, or evaluating).
"""
import ctypes as C
import os

import torch

from . import _lib


class TrainStep:
    def __init__(self, model, B, T, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, process_group=None,
                 optimizer=True, pipeline=False):
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
        lay = _lib.layout(self.dims)
        self.head_off = int(lay.lin_w)
        # pipelined mode: [late_off, total) = {ln2_g, ln2_b, w2, b2}, the last group of the flat layout, is produced and
        # applied one step late (module docstring); needs the optimizer to be this engine's Adam and the GRU model
        self.pipeline = bool(pipeline) and optimizer and not self.dims.simple
        self.late_off = int(lay.ln2_g)
        self.pending = None      # Adam step number of the slice that is still to be applied
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
        self.bwd_opts = _lib.Opts(0.0, 0, 1, 1 if self.pipeline else 0, None)
        self.fwd_opts = _lib.Opts(0.0, 0, 0, 0, None)
        # criterion handed to the forward (as_opts.loss_*): fused into the output layer of the heads where that kernel takes
        # the shape (2 N <= 128 outputs per head, ...), else run by the library as its own kernel -- either way the forward
        # leaves the loss and d loss / d(pre-sigmoid)
        self.fuse_loss = True
        if self.pipeline:
            self.late_stream = torch.cuda.Stream(device=dev)
            self.late_event = torch.cuda.Event()

    def forward_backward(self, tokens, lengths_dev, targets, loss_scale):
        """tokens (B, >=T) int64, lengths_dev (B,) int32 on device, targets (B, >=T, A, 2, N).
        loss_scale = 1 / (N_valid_global * A * N).  Leaves loss in self.loss, gradients in self.grads."""
        L, d, st = _lib.lib(), self.dims, _lib.stream_ptr()
        P = self.model.flat.data
        B, T = self.B, self.T
        fo = self.fwd_opts
        fo.fold_wait_event = None
        if self.fuse_loss:
            fo.loss_targets, fo.loss_tgt_T, fo.loss_scale = targets.data_ptr(), targets.shape[1], float(loss_scale)
            fo.loss_out, fo.loss_dout = self.loss.data_ptr(), self.dout.data_ptr()
        fwd_opts = C.byref(fo) if self.fuse_loss else None
        if self.pipeline and self.pending is not None:
            # the previous step's late slice: weight gradient -> all-reduce -> Adam on the side stream, beside this step's
            # forward recurrences; the forward's fold of the head weights waits for late_event, the recurrences do not
            self._late_update(self.late_stream)
            self.late_event.record(self.late_stream)
            fo.fold_wait_event = self.late_event.cuda_event
            fwd_opts = C.byref(fo)
        _lib.check(L.as_artspeech_fwd(C.byref(d), _lib.ptr(P), _lib.ptr(tokens), tokens.stride(0), _lib.ptr(lengths_dev),
                                      B, T, _lib.ptr(self.out), _lib.ptr(self.ws), 1, fwd_opts, st), "as_artspeech_fwd")
        # criterion backward and the model's final sigmoid backward in one pass: dout holds d(loss)/d(pre-sigmoid)
        # (fused into the forward's output layer when the head is narrow enough: nothing to do here then)
        if not self.fuse_loss:
            _lib.check(L.as_euclid_masked_fwd_bwd_presigmoid(_lib.ptr(self.out), _lib.ptr(targets), targets.shape[1],
                                                         _lib.ptr(lengths_dev), B, T, d.n_art, d.n_samp, float(loss_scale),
                                                         _lib.ptr(self.loss), _lib.ptr(self.dout), _lib.ptr(self.partial), st),
                   "as_euclid_masked_fwd_bwd_presigmoid")
        _lib.check(L.as_artspeech_bwd(C.byref(d), _lib.ptr(P), _lib.ptr(tokens), tokens.stride(0), _lib.ptr(lengths_dev),
                                      B, T, _lib.ptr(self.out), _lib.ptr(self.dout), _lib.ptr(self.grads), _lib.ptr(self.ws),
                                      C.byref(self.bwd_opts), st), "as_artspeech_bwd")

    def _late_update(self, stream):
        """Weight gradient of the heads' second Linear of the step recorded in ``pending`` + its all-reduce + its Adam update,
        enqueued on `stream` after everything the current stream holds (the backward that produced its operands)."""
        L, d = _lib.lib(), self.dims
        P, sp = self.model.flat.data, C.c_void_p(stream.cuda_stream)
        stream.wait_stream(torch.cuda.current_stream())
        _lib.check(L.as_artspeech_dw2(C.byref(d), _lib.ptr(P), self.B, self.T, _lib.ptr(self.grads), _lib.ptr(self.ws), sp),
                   "as_artspeech_dw2")
        lo, n = self.late_off, self.grads.numel() - self.late_off
        if self.use_dist:
            with torch.cuda.stream(stream):
                ev = self._ar_mark("late", stream)
                torch.distributed.all_reduce(self.grads[lo:], op=torch.distributed.ReduceOp.SUM, group=self.pg)
                self._ar_mark_end(ev, stream)
        _lib.check(L.as_adam_step(_lib.ptr(P[lo:]), _lib.ptr(self.grads[lo:]), _lib.ptr(self.exp_avg[lo:]),
                                  _lib.ptr(self.exp_avg_sq[lo:]), n, self.lr, self.betas[0], self.betas[1], self.eps,
                                  self.weight_decay, self.pending, 1.0, sp), "as_adam_step")
        self.pending = None

    def flush(self):
        """Apply the pending late slice now, on the current stream (no-op when nothing is pending)."""
        if self.pipeline and self.pending is not None:
            self._late_update(torch.cuda.current_stream())

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
        end = self.late_off if self.pipeline else self.grads.numel()   # pipelined: [late_off, total) follows one step late
        if not self.ar_overlap:
            cur = torch.cuda.current_stream()
            ev = self._ar_mark("exposed", cur)     # one piece, all of it between the backward and Adam
            dist.all_reduce(self.grads[:end], op=dist.ReduceOp.SUM, group=self.pg)
            self._ar_mark_end(ev, cur)
            return
        _lib.check(_lib.lib().as_artspeech_wait_head_grads(_lib.stream_ptr(), self.comm_stream.cuda_stream), "as_artspeech_wait_head_grads")
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(self.comm_stream):
            ev_t = self._ar_mark("tail", self.comm_stream)
            tail = dist.all_reduce(self.grads[self.head_off:end], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            if ev_t is not None:   # timing pass: the piece's end on its own stream
                tail.wait()
                self._ar_mark_end(ev_t, self.comm_stream)
        ev_x = self._ar_mark("exposed", cur)    # from here on the compute stream does nothing but communicate / wait
        ev_h = self._ar_mark("head", cur)
        head = dist.all_reduce(self.grads[:self.head_off], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        head.wait()
        self._ar_mark_end(ev_h, cur)
        tail.wait()
        cur.wait_stream(self.comm_stream)
        self._ar_mark_end(ev_x, cur)

    # ---- optional timing of the gradient exchange (bench.py --gpus N): stream events around each of its pieces -- "tail"
    # (trunk + heads, on the communication stream beside the GRU backward), "head" (GRU + embedding, on the compute stream
    # after the backward), "late" (the deferred slice, on the late stream in the next step's forward) -- and "exposed": the
    # span of the compute stream between the end of the backward and the start of Adam, i.e. what the exchange costs the step.
    ar_timing = False

    def _ar_mark(self, name, stream):
        if not self.ar_timing:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        return (name, e0, e1)

    def _ar_mark_end(self, ev, stream):
        if ev is not None:
            ev[2].record(stream)
            self.__dict__.setdefault("_ar_events", []).append(ev)

    def ar_report(self):
        """{piece: mean milliseconds} over the steps run with ``ar_timing`` on (synchronises); clears the record."""
        torch.cuda.synchronize()
        acc = {}
        for name, e0, e1 in self.__dict__.pop("_ar_events", []):
            acc.setdefault(name, []).append(e0.elapsed_time(e1))
        return {k: round(sum(v) / len(v), 4) for k, v in acc.items()}

    def adam(self):
        L = _lib.lib()
        self.steps += 1
        n = self.late_off if self.pipeline else self.grads.numel()   # pipelined: the late slice is updated by _late_update
        _lib.check(L.as_adam_step(_lib.ptr(self.model.flat.data), _lib.ptr(self.grads), _lib.ptr(self.exp_avg),
                                  _lib.ptr(self.exp_avg_sq), n, self.lr, self.betas[0], self.betas[1],
                                  self.eps, self.weight_decay, self.steps, 1.0, _lib.stream_ptr()), "as_adam_step")
        if self.pipeline:
            self.pending = self.steps

    def step(self, tokens, lengths_dev, targets, loss_scale):
        self.forward_backward(tokens, lengths_dev, targets, loss_scale)
        self.all_reduce()
        if self.optimizer:
            self.adam()
        return self.loss
