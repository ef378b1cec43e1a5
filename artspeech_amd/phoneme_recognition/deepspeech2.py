"""DeepSpeech2-style articulatory scorer, inference on the C ABI (reference ``phoneme_recognition/deepspeech2.py``).

Same constructor, ``state_dict`` keys and seed-for-seed initialisation as the reference ``DeepSpeech2`` (:90-157): the
sub-modules below are PARAMETER CONTAINERS created in the reference's order; none of their ``forward`` methods is ever
called.  ``DeepSpeech2.forward`` runs (deepspeech2.py:159-195)

    [Adapter: LN -> Linear -> LN -> Linear over the feature axis]                 as_layernorm_fwd + as_gemm_f32
    Conv2d(Cin, 32, 3x3) (+ voicing)                                              as_conv3x3_stem
    ResidualCNN x N: (LN over features -> GELU -> Conv2d(32, 32, 3x3)) x 2 + skip as_ln_feat_gelu + as_conv3x3_c32 (MFMA)
    Linear(32*D -> H)                                                             as_gemm_f32
    RecurrentBlock x M: LN -> GELU -> uni-GRU                                     as_layernorm_fwd + as_gelu + as_gemm_f32
                                                                                  + as_gru_unidir_fwd
    Linear -> GELU (features), Linear (logits)                                    as_gemm_f32 (GELU epilogue)

on channels-last feature maps ``[B][T][D][32]``: a frame's 32*D features are one contiguous row, so the reference's
``view(B, C*D, T).permute(2, 0, 1)`` (:183-185) costs nothing -- the Linear weight's columns are permuted once instead.
Inference only (``eval()`` mode, the way ``phoneme_recognition/__init__.py:213-236`` scores); there is no CPU path.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib

OUT_CHANNELS = 32  # deepspeech2.py:104


class _Params(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the scorer runs in DeepSpeech2.forward on the HIP library")


class ResidualCNN(_Params):
    """Parameters of deepspeech2.py:15-27 (kernel 3, stride 1)."""

    def __init__(self, channels, num_features):
        super().__init__()
        self.cnn1 = nn.Conv2d(channels, channels, 3, 1, padding=1)
        self.layer_norm1 = nn.LayerNorm(num_features)
        self.cnn2 = nn.Conv2d(channels, channels, 3, 1, padding=1)
        self.layer_norm2 = nn.LayerNorm(num_features)


class RecurrentBlock(_Params):
    """Parameters of deepspeech2.py:50-62."""

    def __init__(self, size):
        super().__init__()
        self.rnn = nn.GRU(input_size=size, hidden_size=size, num_layers=1, bidirectional=False, batch_first=False)
        self.layer_norm = nn.LayerNorm(size)


class Adapter(_Params):
    """Parameters of deepspeech2.py:73-81."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.adapter = nn.Sequential(nn.LayerNorm(in_features), nn.Linear(in_features, out_features),
                                     nn.LayerNorm(out_features), nn.Linear(out_features, out_features))


_SLAB = {}


def _slab(dev):
    if dev not in _SLAB:
        _SLAB[dev] = torch.empty(4 << 20, dtype=torch.float32, device=dev)  # split-K partial tiles (16 MB)
    return _SLAB[dev]


def _gemm(A, W, bias, out, act=0, split_k=False):
    """out[M][N] = act(A[M][K] . W[N][K]^T + bias).  split_k (act == 0 only): few output tiles under a long reduction --
    the bias is laid down first and the GEMM accumulates onto it, its K range split over workgroups (deterministic slabs)."""
    assert A.is_contiguous() and W.is_contiguous() and out.is_contiguous(), "bare pointers below: dense row-major operands"
    g = _lib.Gemm()
    g.A, g.B, g.C = A.data_ptr(), W.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = A.shape[0], W.shape[0], W.shape[1]
    g.a_i, g.a_k, g.b_j, g.b_k, g.ldc = g.K, 1, g.K, 1, g.N
    g.batch, g.act = 1, act
    if split_k and act == 0:
        out.copy_(bias.expand_as(out))
        slab = _slab(out.device)
        g.accumulate, g.splitk_ws, g.splitk_ws_floats = 1, slab.data_ptr(), slab.numel()
    else:
        g.bias = bias.data_ptr()
    _lib.check(_lib.lib().as_gemm_f32(C.byref(g), _lib.stream_ptr()), "as_gemm_f32")
    return out


def _ln(x, ln, out):
    rows, D = x.shape
    assert x.is_contiguous() and out.is_contiguous(), "bare pointers below: dense row-major operands"
    _lib.check(_lib.lib().as_layernorm_fwd(_lib.ptr(x), None, _lib.ptr(ln.weight), _lib.ptr(ln.bias), _lib.ptr(out), None, None,
                                           rows, D, 0, _lib.stream_ptr()), "as_layernorm_fwd")
    return out


def top1_phonemes(logits):
    """``torch.topk(outputs, k=1, dim=-1).indices`` (phoneme_recognition/__init__.py:236, decoders.py:36)."""
    return torch.topk(logits, k=1, dim=-1).indices


class DeepSpeech2(nn.Module):
    def __init__(self, in_channels, num_residual_layers, num_rnn_layers, rnn_hidden_size, num_classes=31, num_features=80,
                 dropout=0.1, adapter_out_features=None):
        super().__init__()
        if adapter_out_features is not None:
            self.adapter = Adapter(num_features, adapter_out_features)
            num_features = adapter_out_features
        else:
            self.adapter = None
        self.cnn = nn.Conv2d(in_channels, OUT_CHANNELS, 3, stride=1, padding=1)
        self.residual_layers = nn.ModuleList([ResidualCNN(OUT_CHANNELS, num_features) for _ in range(num_residual_layers)])
        self.linear = nn.Linear(num_features * OUT_CHANNELS, rnn_hidden_size)
        self.recurrent_layers = nn.ModuleList([RecurrentBlock(rnn_hidden_size) for _ in range(num_rnn_layers)])
        self.feature_extractor = nn.Sequential(nn.Linear(rnn_hidden_size, rnn_hidden_size), nn.GELU())
        self.classifier = nn.Linear(rnn_hidden_size, num_classes)
        self.dropout_p = dropout  # nn.Dropout holds no state; eval-mode forward never applies it
        self.num_features, self.hidden, self.num_classes, self.in_channels = num_features, rnn_hidden_size, num_classes, in_channels
        self._prepared = None

    @property
    def total_parameters(self):
        return sum(p.numel() for p in self.parameters())

    @staticmethod
    def get_noise_logits(x, factor):
        return x + factor * torch.randn_like(x)

    @staticmethod
    def get_normalized_outputs(x, use_log_prob=False):
        return (torch.log_softmax if use_log_prob else torch.softmax)(x, dim=-1)

    # ------------------------------------------------------------------ weights in kernel order (cached per version)
    def _prepare(self):
        # (address, version) of every parameter; the entry also holds the parameters' storages, so none of those
        # addresses can be freed and handed to a different tensor while the entry is alive
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._prepared is not None and self._prepared[0] == key:
            return self._prepared[1]
        keep = [p.untyped_storage() for p in self.parameters()]
        with torch.no_grad():
            taps = lambda conv: conv.weight.permute(2, 3, 0, 1).contiguous()  # [kd][kt][co][ci]
            D = self.num_features
            w = dict(stem=taps(self.cnn),
                     res=[(taps(r.cnn1), taps(r.cnn2)) for r in self.residual_layers],
                     # column c*D + d of the reference's (B, C*D, T) view -> column d*32 + c of a channels-last frame row
                     linear=self.linear.weight.view(self.hidden, OUT_CHANNELS, D).permute(0, 2, 1).reshape(self.hidden, -1).contiguous())
        self._prepared = (key, w, keep)
        return w

    def forward(self, x, voicing=None, return_features=False):
        """x (B, C, D, T) float32 on the GPU, voicing (B, T) or None -> logits (B, T, classes) [, features (B, T, H)]."""
        if self.training:
            raise RuntimeError("DeepSpeech2 (HIP): inference only -- call .eval() (the reference scores in eval mode)")
        _lib.require_gpu(x, "x")
        L, st = _lib.lib(), _lib.stream_ptr()
        B, Cin, Din, T = x.shape
        assert Cin == self.in_channels
        x = x.float()
        dev, f32 = x.device, torch.float32
        w = self._prepare()
        D, H = self.num_features, self.hidden
        with torch.no_grad():
            if self.adapter is not None:
                ad = self.adapter.adapter
                # the reference's own transpose (:84); glue copy of the input.  (.contiguous(): with B = C = 1 the reshape of the
                # transposed view is itself a VIEW with strides (1, T) -- the kernels below take bare pointers)
                rows = x.transpose(2, 3).reshape(B * Cin * T, Din).contiguous()
                a = _ln(rows, ad[0], torch.empty_like(rows))
                a = _gemm(a, ad[1].weight, ad[1].bias, torch.empty(rows.shape[0], D, device=dev, dtype=f32))
                a = _ln(a, ad[2], torch.empty_like(a))
                planes = _gemm(a, ad[3].weight, ad[3].bias, torch.empty_like(a))  # (B, C, T, D)
                strides = (Cin * T * D, T * D, 1, D)
            else:
                assert Din == D
                planes = x.contiguous()  # (B, C, D, T)
                strides = (Cin * D * T, D * T, T, 1)
            if voicing is not None:
                voicing = voicing.to(device=dev, dtype=f32).contiguous()
            fmap = torch.empty(B, T, D, OUT_CHANNELS, device=dev, dtype=f32)
            _lib.check(L.as_conv3x3_stem(_lib.ptr(planes), *strides, _lib.ptr(w["stem"]), _lib.ptr(self.cnn.bias),
                                         _lib.ptr(voicing) if voicing is not None else None, _lib.ptr(fmap), B, T, D, Cin, st),
                       "as_conv3x3_stem")
            act, mid = torch.empty_like(fmap), torch.empty_like(fmap)
            for r, (w1, w2) in zip(self.residual_layers, w["res"]):
                _lib.check(L.as_ln_feat_gelu(_lib.ptr(fmap), _lib.ptr(r.layer_norm1.weight), _lib.ptr(r.layer_norm1.bias), _lib.ptr(act),
                                             B * T, D, OUT_CHANNELS, st), "as_ln_feat_gelu")
                _lib.check(L.as_conv3x3_c32(_lib.ptr(act), _lib.ptr(w1), _lib.ptr(r.cnn1.bias), None, _lib.ptr(mid), B, T, D, st),
                           "as_conv3x3_c32")
                _lib.check(L.as_ln_feat_gelu(_lib.ptr(mid), _lib.ptr(r.layer_norm2.weight), _lib.ptr(r.layer_norm2.bias), _lib.ptr(act),
                                             B * T, D, OUT_CHANNELS, st), "as_ln_feat_gelu")
                nxt = torch.empty_like(fmap)
                _lib.check(L.as_conv3x3_c32(_lib.ptr(act), _lib.ptr(w2), _lib.ptr(r.cnn2.bias), _lib.ptr(fmap), _lib.ptr(nxt), B, T, D, st),
                           "as_conv3x3_c32")
                fmap = nxt
            h = _gemm(fmap.view(B * T, D * OUT_CHANNELS), w["linear"], self.linear.bias, torch.empty(B * T, H, device=dev, dtype=f32),
                      split_k=True)  # K = 32 * D (2560) against N = H (64) columns
            lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
            gi = torch.empty(B * T, 3 * H, device=dev, dtype=f32)
            for blk in self.recurrent_layers:
                a = _ln(h, blk.layer_norm, torch.empty_like(h))
                _lib.check(L.as_gelu(_lib.ptr(a), _lib.ptr(a), a.numel(), st), "as_gelu")
                _gemm(a, blk.rnn.weight_ih_l0, blk.rnn.bias_ih_l0, gi)
                h = torch.empty_like(h)
                _lib.check(L.as_gru_unidir_fwd(_lib.ptr(gi), _lib.ptr(blk.rnn.weight_hh_l0), _lib.ptr(blk.rnn.bias_hh_l0),
                                               _lib.ptr(lengths), B, T, H, _lib.ptr(h), st), "as_gru_unidir_fwd")
            fe = self.feature_extractor[0]
            features = _gemm(h, fe.weight, fe.bias, torch.empty_like(h), act=3)
            logits = _gemm(features, self.classifier.weight, self.classifier.bias,
                           torch.empty(B * T, self.num_classes, device=dev, dtype=f32))
        logits, features = logits.view(B, T, -1), features.view(B, T, H)
        return (logits, features) if return_features else logits
