"""CPU ORACLE for the principal-components recurrent model -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy (float64) restatement of reference ``phoneme_to_articulation/principal_components/models/rnn.py`` (forward) for
both cells of the ``RNNType`` switch (``phoneme_to_articulation/__init__.py:47-49``), driven by a state_dict with the
reference's key names.  Pinned by ``tests/test_oracle_golden.py`` against ``tests/golden/pc_{lstm,gru}_small.npz``
(outputs of the reference itself).  Gradients are pinned by the same fixtures (the reference's autograd results).
"""
import numpy as np

from .artspeech_oracle import _sigmoid, gru_dir_fwd, layernorm_fwd


def lstm_dir_fwd(x, lengths, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one nn.LSTM layer on a packed batch (rnn.py:58-68, 97-104): gate rows [i; f; g; o], h0 = c0 = 0,
    sequence b participates for t < lengths[b], the reverse direction walks t = len_b-1 .. 0, padded outputs are zeros."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    y = np.zeros((B, T, H), x.dtype)
    h, c = np.zeros((B, H), x.dtype), np.zeros((B, H), x.dtype)
    gi_all = x @ w_ih.T + b_ih
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        act = lengths > t
        if not act.any():
            continue
        g = gi_all[:, t] + h @ w_hh.T + b_hh
        i, f, gg, o = _sigmoid(g[:, :H]), _sigmoid(g[:, H:2 * H]), np.tanh(g[:, 2 * H:3 * H]), _sigmoid(g[:, 3 * H:])
        c_new = f * c + i * gg
        h_new = o * np.tanh(c_new)
        h = np.where(act[:, None], h_new, h)
        c = np.where(act[:, None], c_new, c)
        y[act, t] = h_new[act]
    return y


def birnn_fwd(p, prefix, x, lengths, lstm, layers=2):
    """nn.GRU / nn.LSTM(num_layers=2, bidirectional=True, batch_first=True) on a packed batch, zero padded output."""
    for l in range(layers):
        outs = []
        for sfx, rev in (("", False), ("_reverse", True)):
            args = (x, lengths, p[f"{prefix}weight_ih_l{l}{sfx}"], p[f"{prefix}weight_hh_l{l}{sfx}"], p[f"{prefix}bias_ih_l{l}{sfx}"],
                    p[f"{prefix}bias_hh_l{l}{sfx}"], rev)
            outs.append(lstm_dir_fwd(*args) if lstm else gru_dir_fwd(*args)[0])
        x = np.concatenate(outs, -1)
    return x


def forward(params, tokens, lengths, lstm):
    """PrincipalComponentsArtSpeech.forward (rnn.py:86-109): tokens (B, T) -> components (B, max(lengths), latent)."""
    p = {k: np.asarray(v, np.float64) for k, v in params.items()}
    lengths = np.asarray(lengths)
    T = int(lengths.max())
    x = p["embedding.weight"][np.asarray(tokens)[:, :T]]
    x = birnn_fwd(p, "rnn.", x, lengths, lstm)
    x = np.maximum(x @ p["linear.0.weight"].T + p["linear.0.bias"], 0)
    for ln, lin, relu in ((0, 1, True), (3, 4, True), (6, 7, False)):  # PrincipalComponentsPredictor (rnn.py:19-33)
        x = layernorm_fwd(x, p[f"predictor.linear.{ln}.weight"], p[f"predictor.linear.{ln}.bias"])[0]
        x = x @ p[f"predictor.linear.{lin}.weight"].T + p[f"predictor.linear.{lin}.bias"]
        if relu:
            x = np.maximum(x, 0)
    return np.tanh(x)
