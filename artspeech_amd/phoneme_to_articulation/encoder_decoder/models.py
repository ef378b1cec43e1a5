"""Drop-in models of the model-free phoneme-to-articulation path on MI355X.

Mirrors reference ``phoneme_to_articulation/encoder_decoder/models.py``: ``ArtSpeech`` (:99-145) and
``SimpleArtSpeech`` (:53-96) keep their constructor signatures, ``forward`` contracts,
``total_parameters`` and -- the checkpoint contract -- their ``state_dict`` keys and shapes
(``predictors.{a}.*`` are the reference's ``ArticulatorPredictor`` (:7-33) parameters, stacked over heads).  Internally every parameter lives in ONE flat fp32 buffer (``self.flat``): one gradient
buffer, one RCCL all-reduce, one optimizer launch; all device math runs in libartspeech_hip.so.
There is no CPU path: tensors must be on an MI355X.
"""
import ctypes as C

import torch
import torch.nn as nn

from ... import _lib

HEAD_HIDDEN = 256  # fixed width of ArticulatorPredictor (reference models.py:12-17)


def _reference_init(vocab_size, n_articulators, embed_dim, hidden_size, n_samples, simple, in_features=None):
    """Default PyTorch initialisation, drawn in the reference's construction order so that the same
    torch seed yields the same initial weights (Embedding N(0,1); GRU / Linear U(+-1/sqrt(fan));
    LayerNorm 1/0).  Returns {state_dict key: tensor}."""
    sd = {}
    if vocab_size is not None:
        sd["embedding.weight"] = nn.Embedding(vocab_size, embed_dim).weight.detach()
        if simple:
            nn.Dropout(0.0)
            lin = nn.Linear(embed_dim, hidden_size)
        else:
            rnn = nn.GRU(embed_dim, hidden_size, num_layers=2, bidirectional=True, batch_first=True)
            for k, v in rnn.state_dict().items():
                sd[f"rnn.{k}"] = v.detach()
            lin = nn.Linear(2 * hidden_size, hidden_size)
        sd["linear.0.weight"], sd["linear.0.bias"] = lin.weight.detach(), lin.bias.detach()
        in_features = hidden_size
    for a in range(n_articulators):
        pre = f"predictors.{a}." if vocab_size is not None else ""
        dims = [(in_features, None), (in_features, HEAD_HIDDEN), (HEAD_HIDDEN, None), (HEAD_HIDDEN, HEAD_HIDDEN),
                (HEAD_HIDDEN, None)]
        for idx, (i, o) in zip((0, 1, 3, 4, 6), dims):
            m = nn.LayerNorm(i) if o is None else nn.Linear(i, o)
            sd[f"{pre}linear.{idx}.weight"], sd[f"{pre}linear.{idx}.bias"] = m.weight.detach(), m.bias.detach()
        for name in ("x_coords", "y_coords"):
            m = nn.Linear(HEAD_HIDDEN, n_samples)
            sd[f"{pre}{name}.weight"], sd[f"{pre}{name}.bias"] = m.weight.detach(), m.bias.detach()
    return sd


def _build_views(dims, lay):
    """state_dict key -> (offset in floats, shape) inside the flat buffer (include/artspeech_hip.h)."""
    V, A, E, H, N = dims.vocab, dims.n_art, dims.embed, dims.hidden, dims.n_samp
    D = HEAD_HIDDEN
    v = {"embedding.weight": (lay.embedding, (V, E))}
    if not dims.simple:
        for l, inp in ((0, E), (1, 2 * H)):
            for d, sfx in enumerate(("", "_reverse")):
                v[f"rnn.weight_ih_l{l}{sfx}"] = (lay.w_ih[l] + d * 3 * H * inp, (3 * H, inp))
                v[f"rnn.weight_hh_l{l}{sfx}"] = (lay.w_hh[l] + d * 3 * H * H, (3 * H, H))
                v[f"rnn.bias_ih_l{l}{sfx}"] = (lay.b_ih[l] + d * 3 * H, (3 * H,))
                v[f"rnn.bias_hh_l{l}{sfx}"] = (lay.b_hh[l] + d * 3 * H, (3 * H,))
        v["linear.0.weight"] = (lay.lin_w, (H, 2 * H))
    else:
        v["linear.0.weight"] = (lay.lin_w, (H, E))
    v["linear.0.bias"] = (lay.lin_b, (H,))
    for a in range(A):
        p = f"predictors.{a}."
        v[p + "linear.0.weight"] = (lay.ln1_g + a * H, (H,))
        v[p + "linear.0.bias"] = (lay.ln1_b + a * H, (H,))
        v[p + "linear.1.weight"] = (lay.w1 + a * D * H, (D, H))
        v[p + "linear.1.bias"] = (lay.b1 + a * D, (D,))
        v[p + "linear.3.weight"] = (lay.ln2_g + a * D, (D,))
        v[p + "linear.3.bias"] = (lay.ln2_b + a * D, (D,))
        v[p + "linear.4.weight"] = (lay.w2 + a * D * D, (D, D))
        v[p + "linear.4.bias"] = (lay.b2 + a * D, (D,))
        v[p + "linear.6.weight"] = (lay.ln3_g + a * D, (D,))
        v[p + "linear.6.bias"] = (lay.ln3_b + a * D, (D,))
        v[p + "x_coords.weight"] = (lay.w3 + a * 2 * N * D, (N, D))
        v[p + "y_coords.weight"] = (lay.w3 + a * 2 * N * D + N * D, (N, D))
        v[p + "x_coords.bias"] = (lay.b3 + a * 2 * N, (N,))
        v[p + "y_coords.bias"] = (lay.b3 + a * 2 * N + N, (N,))
    return v


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


class _ArtSpeechFn(torch.autograd.Function):
    """outputs = model(tokens, lengths) with every kernel of forward and backward enqueued by
    as_artspeech_fwd / as_artspeech_bwd (C ABI)."""

    @staticmethod
    def forward(ctx, flat, tokens, lengths_dev, dims, B, T, opts=None, defer=None):
        L = _lib.lib()
        # `train` = keep what the backward needs.  Dropout (opts) is a property of the MODULE's mode, not of grad mode: the
        # reference's nn.Dropout / nn.GRU(dropout=p) also drop under torch.no_grad() while model.training is set, so the C
        # side is told to run in training mode whenever opts carry a dropout probability
        train = bool(ctx.needs_input_grad[0])  # (grad mode is always off inside Function.forward)
        out = torch.empty((B, T, dims.n_art, 2, dims.n_samp), dtype=torch.float32, device=flat.device)
        n_ws = L.as_artspeech_workspace_floats(C.byref(dims), B, T)
        if n_ws <= 0:
            _lib.check(int(n_ws) or -1, "as_artspeech_workspace_floats")
        ws = torch.empty(n_ws, dtype=torch.float32, device=flat.device)
        _lib.check(L.as_artspeech_fwd(C.byref(dims), _lib.ptr(flat), _lib.ptr(tokens), tokens.stride(0),
                                      _lib.ptr(lengths_dev), B, T, _lib.ptr(out), _lib.ptr(ws), int(train or opts is not None),
                                      C.byref(opts) if opts is not None else None, _lib.stream_ptr()), "as_artspeech_fwd")
        # nn.Embedding raises for ids outside [0, V) (reference models.py:135); the kernels clamp them (memory safety) and
        # count them in the first word of the workspace -- read here, where the drop-in path may synchronise
        if defer is None:
            _raise_if_bad_tokens(ws, dims.vocab)
        else:
            # a loop that synchronises anyway (loss.item()) calls model.check_tokens() there.  Only the flag word is kept (a
            # 4-byte copy): holding the workspace itself would pin ~1 GB per batch until the check
            defer.append(ws[:1].clone())
        if train:
            ctx.save_for_backward(flat, tokens, lengths_dev, out, ws)
            ctx.meta = (dims, B, T, opts)
        return out

    @staticmethod
    def backward(ctx, dout):
        flat, tokens, lengths_dev, out, ws = ctx.saved_tensors
        dims, B, T, opts = ctx.meta
        L = _lib.lib()
        dout = dout.contiguous()
        grads = torch.zeros_like(flat)  # padding words between parameter groups stay zero
        _lib.check(L.as_artspeech_bwd(C.byref(dims), _lib.ptr(flat), _lib.ptr(tokens), tokens.stride(0),
                                      _lib.ptr(lengths_dev), B, T, _lib.ptr(out), _lib.ptr(dout), _lib.ptr(grads),
                                      _lib.ptr(ws), C.byref(opts) if opts is not None else None, _lib.stream_ptr()),
                   "as_artspeech_bwd")
        return grads, None, None, None, None, None, None, None


def _raise_if_bad_tokens(ws, vocab):
    n_bad = int(ws[:1].view(torch.int32).item())
    if n_bad:
        raise IndexError(f"index out of range in self ({n_bad} token ids outside [0, {vocab}))")


class _FlatModule(nn.Module):
    """nn.Module whose parameters are views of one flat buffer but whose state_dict speaks the
    reference's key names (checkpoints are loaded with strict load_state_dict:
    train_phoneme_to_articulation.py:165-167)."""

    def _setup(self, dims, init_sd):
        self.dims = dims
        self._lay = _lib.layout(dims)
        self._views = _build_views(dims, self._lay)
        flat = torch.zeros(self._lay.total, dtype=torch.float32)
        for k, (off, shape) in self._views.items():
            flat[off:off + _numel(shape)] = init_sd[k].reshape(-1).to(torch.float32)
        self.flat = nn.Parameter(flat)

    # Token-id check (nn.Embedding raises IndexError, reference models.py:135).  Default: forward() reads the device-side
    # count right after its launch (one synchronisation, the reference's behaviour).  A training loop that synchronises
    # once per step anyway sets ``defer_token_check = True`` and calls ``check_tokens()`` next to its ``loss.item()``.
    defer_token_check = False

    def _defer_list(self):
        if not self.defer_token_check:
            return None
        if not hasattr(self, "_pending_ws"):
            self._pending_ws = []
        return self._pending_ws

    def check_tokens(self):
        pending, self._pending_ws = getattr(self, "_pending_ws", []), []
        for ws in pending:
            _raise_if_bad_tokens(ws, self.dims.vocab)

    def named_views(self):
        """state_dict key -> view of the flat parameter (shares storage)."""
        return {k: self.flat.detach()[off:off + _numel(shape)].view(shape) for k, (off, shape) in self._views.items()}

    def named_grad_views(self):
        if self.flat.grad is None:
            return {}
        return {k: self.flat.grad[off:off + _numel(shape)].view(shape) for k, (off, shape) in self._views.items()}

    @property
    def total_parameters(self):
        return sum(_numel(shape) for _, shape in self._views.values())

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for k, v in self.named_views().items():
            destination[prefix + k] = v if keep_vars else v.clone()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        views = self.named_views()
        for k, dst in views.items():
            key = prefix + k
            if key not in state_dict:
                missing_keys.append(key)
                continue
            src = state_dict[key]
            if tuple(src.shape) != tuple(dst.shape):
                error_msgs.append(f"size mismatch for {key}: copying a param with shape {tuple(src.shape)} from "
                                  f"checkpoint, the shape in current model is {tuple(dst.shape)}.")
                continue
            with torch.no_grad():
                dst.copy_(src)
        if strict:
            for key in state_dict:
                if key.startswith(prefix) and key[len(prefix):] not in views:
                    unexpected_keys.append(key)


class ArtSpeech(_FlatModule):
    """Embedding -> 2-layer bidirectional GRU (packed) -> Linear+ReLU -> A ArticulatorPredictor heads ->
    sigmoid (reference models.py:99-145)."""

    def __init__(self, vocab_size, n_articulators, embed_dim=64, hidden_size=128, n_samples=50, dropout=0.):
        super().__init__()
        self.dropout = float(dropout)
        dims = _lib.Dims(vocab_size, n_articulators, embed_dim, hidden_size, n_samples, 0)
        self._setup(dims, _reference_init(vocab_size, n_articulators, embed_dim, hidden_size, n_samples, simple=False))

    def forward(self, x, lengths):
        """
        Args:
            x (torch.tensor): (bs, seq_len) int64 phoneme indices on the GPU.
            lengths: lengths of the sequences, sorted in decreasing order (CPU tensor or list, as the
                reference's pack_padded_sequence requires).
        Return:
            (bs, max(lengths), n_articulators, 2, n_samples)
        """
        _lib.require_gpu(x, "x")
        _lib.require_gpu(self.flat, "model parameters")
        lengths_cpu = torch.as_tensor(lengths, dtype=torch.int32, device="cpu")
        if lengths_cpu.numel() != x.shape[0]:
            raise RuntimeError(f"Expected `len(lengths)` to be equal to batch_size, but got {lengths_cpu.numel()} "
                               f"(batch_size={x.shape[0]})")
        if lengths_cpu.numel() > 1 and bool((lengths_cpu[1:] > lengths_cpu[:-1]).any()):
            raise RuntimeError("`lengths` array must be sorted in decreasing order when `enforce_sorted` is True.")
        if int(lengths_cpu.min()) <= 0:
            raise RuntimeError("Length of all samples has to be greater than 0, but found an element in 'lengths' "
                               "that is <= 0")
        T = int(lengths_cpu.max())
        if T > x.shape[1]:
            raise RuntimeError(f"lengths.max()={T} exceeds the padded sequence length {x.shape[1]}")
        if x.dtype != torch.int64:
            x = x.long()
        if x.stride(1) != 1:
            x = x.contiguous()
        lengths_dev = lengths_cpu.to(x.device, non_blocking=True)
        opts = None
        if self.training and self.dropout > 0.0:
            # nn.GRU(dropout=p): inter-layer dropout in training mode.  The seed is drawn from torch's CPU
            # generator, so torch.manual_seed() controls it (statistical parity with the reference's mask).
            opts = _lib.Opts(self.dropout, int(torch.randint(0, 2 ** 62, (1,)).item()))
        return _ArtSpeechFn.apply(self.flat, x, lengths_dev, self.dims, x.shape[0], T, opts, self._defer_list())


class SimpleArtSpeech(_FlatModule):
    """ArtSpeech without the recurrent encoder (reference models.py:53-96); ``lengths`` is ignored."""

    def __init__(self, vocab_size, n_articulators, embed_dim=64, hidden_size=128, num_samples=50, dropout=0.):
        super().__init__()
        self.dropout = float(dropout)
        dims = _lib.Dims(vocab_size, n_articulators, embed_dim, hidden_size, num_samples, 1)
        self._setup(dims, _reference_init(vocab_size, n_articulators, embed_dim, hidden_size, num_samples, simple=True))

    def forward(self, x, lengths=None):
        _lib.require_gpu(x, "x")
        _lib.require_gpu(self.flat, "model parameters")
        if x.dtype != torch.int64:
            x = x.long()
        if x.stride(1) != 1:
            x = x.contiguous()
        opts = None
        if self.training and self.dropout > 0.0:
            # nn.Dropout on the embedded frames (reference models.py:64,85): per-position counter mask, seed drawn from
            # torch's CPU generator like the GRU model's inter-layer dropout
            opts = _lib.Opts(self.dropout, int(torch.randint(0, 2 ** 62, (1,)).item()))
        return _ArtSpeechFn.apply(self.flat, x, None, self.dims, x.shape[0], x.shape[1], opts, self._defer_list())
