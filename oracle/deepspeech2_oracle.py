"""CPU ORACLE for the DeepSpeech2-style articulatory scorer -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy (float64) restatement of reference ``phoneme_recognition/deepspeech2.py`` in eval mode (dropout = identity),
driven by a state_dict with the reference's key names.  Pinned by ``tests/test_oracle_golden.py`` against
``tests/golden/deepspeech2_{small,plain}.npz`` (outputs of the reference itself, tests/golden/make_golden.py).
"""
import numpy as np
from scipy.special import erf

from .artspeech_oracle import _sigmoid, layernorm_fwd


def gelu(x):
    """F.gelu default (exact erf form) -- deepspeech2.py:36, 44, 66, 141."""
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def _ln(x, p, prefix):
    return layernorm_fwd(x, p[prefix + "weight"], p[prefix + "bias"])[0]


def _lin(x, p, prefix):
    return x @ p[prefix + "weight"].T + p[prefix + "bias"]


def conv3x3(x, w, b):
    """nn.Conv2d(kernel 3, stride 1, padding 1) on x (B, Ci, D, T) with w (Co, Ci, 3, 3) -- deepspeech2.py:22, 25, 104."""
    B, Ci, D, T = x.shape
    xp = np.zeros((B, Ci, D + 2, T + 2), x.dtype)
    xp[:, :, 1:-1, 1:-1] = x
    out = np.zeros((B, w.shape[0], D, T), x.dtype)
    for kd in range(3):
        for kt in range(3):
            out += np.einsum("oc,bcdt->bodt", w[:, :, kd, kt], xp[:, :, kd:kd + D, kt:kt + T])
    return out + b[None, :, None, None]


def ln_features(x, p, prefix):
    """transpose(2, 3) -> LayerNorm(num_features) -> transpose(2, 3) (deepspeech2.py:30-32, 38-40)."""
    return _ln(x.transpose(0, 1, 3, 2), p, prefix).transpose(0, 1, 3, 2)


def residual_cnn(x, p, prefix):
    """ResidualCNN.forward (:29-47)."""
    out = conv3x3(gelu(ln_features(x, p, prefix + "layer_norm1.")), p[prefix + "cnn1.weight"], p[prefix + "cnn1.bias"])
    out = conv3x3(gelu(ln_features(out, p, prefix + "layer_norm2.")), p[prefix + "cnn2.weight"], p[prefix + "cnn2.bias"])
    return out + x


def gru(x, p, prefix):
    """nn.GRU(num_layers=1, unidirectional, batch_first=False) with h0 = 0 (:54-60); x (T, B, I) -> (T, B, H)."""
    w_ih, w_hh = p[prefix + "weight_ih_l0"], p[prefix + "weight_hh_l0"]
    b_ih, b_hh = p[prefix + "bias_ih_l0"], p[prefix + "bias_hh_l0"]
    H = w_hh.shape[1]
    T, B, _ = x.shape
    h = np.zeros((B, H), x.dtype)
    ys = []
    for t in range(T):
        gi = x[t] @ w_ih.T + b_ih
        gh = h @ w_hh.T + b_hh
        r = _sigmoid(gi[:, :H] + gh[:, :H])
        z = _sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        ys.append(h)
    return np.stack(ys)


def forward(params, x, voicing=None):
    """DeepSpeech2.forward(x, voicing, return_features=True) (:159-195): x (B, C, D, T) -> logits (B, T, classes),
    features (B, T, H).  Layer counts and the adapter are read off the state_dict keys."""
    p = {k: np.asarray(v, np.float64) for k, v in params.items()}
    x = np.asarray(x, np.float64)
    if "adapter.adapter.0.weight" in p:  # Adapter.forward (:83-87)
        h = x.transpose(0, 1, 3, 2)
        h = _lin(_ln(h, p, "adapter.adapter.0."), p, "adapter.adapter.1.")
        h = _lin(_ln(h, p, "adapter.adapter.2."), p, "adapter.adapter.3.")
        x = h.transpose(0, 1, 3, 2)
    out = conv3x3(x, p["cnn.weight"], p["cnn.bias"])
    if voicing is not None:
        out = out + np.asarray(voicing, np.float64)[:, None, None, :]
    l = 0
    while f"residual_layers.{l}.cnn1.weight" in p:
        out = residual_cnn(out, p, f"residual_layers.{l}.")
        l += 1
    B, Cc, D, T = out.shape
    out = out.reshape(B, Cc * D, T).transpose(2, 0, 1)  # (T, B, C*D)
    out = _lin(out, p, "linear.")
    l = 0
    while f"recurrent_layers.{l}.rnn.weight_ih_l0" in p:
        pre = f"recurrent_layers.{l}."
        out = gru(gelu(_ln(out, p, pre + "layer_norm.")), p, pre + "rnn.")  # RecurrentBlock.forward (:64-70)
        l += 1
    out = out.transpose(1, 0, 2)
    features = gelu(_lin(out, p, "feature_extractor.0."))
    return _lin(features, p, "classifier."), features
