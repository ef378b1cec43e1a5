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


# ---- autoencoders and the critical-distance loss of the same method ------------------------------------------------------
def _mlp(x, p, prefix):
    """nn.Sequential(Linear, ReLU, Linear, ReLU, Linear) -- Encoder / Decoder (models/autoencoder.py:83-112)."""
    for idx, relu in ((0, True), (2, True), (4, False)):
        x = x @ p[f"{prefix}{idx}.weight"].T + p[f"{prefix}{idx}.bias"]
        if relu:
            x = np.maximum(x, 0)
    return x


def _indices(comps):
    """make_indices_dict (helpers.py:94-114) on {articulator: n_components} in insertion order."""
    out, start = {}, 0
    for name, n in comps.items():
        out[name] = list(range(start, start + n))
        start += n
    return out


def autoencoder_forward(params, x, comps):
    """MultiArticulatorAutoencoder.forward (models/autoencoder.py:253-260): x (bs, A, F), channels in sorted-name order ->
    (outputs (bs, A, F), latent (bs, latent_size)); a latent index takes the max over the encoders that own it (:155-173)."""
    p = {k: np.asarray(v, np.float64) for k, v in params.items()}
    x = np.asarray(x, np.float64)
    idx = _indices(comps)
    names = sorted(idx)
    latent_size = 1 + max(i for v in idx.values() for i in v)
    spaces = np.full((x.shape[0], len(names), latent_size), -np.inf)
    for i, name in enumerate(names):
        spaces[:, i, idx[name]] = _mlp(x[:, i], p, f"encoders.encoders.{name}.encoder.")
    latent = np.tanh(spaces.max(1))
    outs = np.stack([_mlp(latent[:, idx[name]], p, f"decoders.decoders.{name}.decoder.") for name in names], 1)
    return outs, latent


def critical_loss(shapes, reference_arrays, mask, TVs, articulators):
    """CriticalLoss.forward (losses.py:52-99) without denormalisation: per tract variable the minimum pairwise distance
    between its two articulators' contours per frame, mean over mask == 1.  shapes (bs, T, A, 2, N)."""
    pairs = {"LA": ("lower-lip", "upper-lip"), "TTCD": ("tongue", "upper-incisor"), "TBCD": ("tongue", "upper-incisor"),
             "VEL": ("soft-palate", "pharynx")}
    shapes = np.asarray(shapes, np.float64)
    arts = list(articulators)
    if "upper-incisor" not in arts:
        arts = sorted(arts + ["upper-incisor"])
        r = arts.index("upper-incisor")
        shapes = np.concatenate([shapes[:, :, :r], np.asarray(reference_arrays, np.float64), shapes[:, :, r:]], 2)
    vals = []
    for tv in sorted(TVs):
        a, b = (shapes[:, :, arts.index(n)] for n in pairs[tv])          # (bs, T, 2, N)
        d = np.sqrt(((a[..., :, None] - b[..., None, :]) ** 2).sum(-3))  # (bs, T, N, N)
        vals.append(d.reshape(*d.shape[:2], -1).min(-1))
    crit = np.stack(vals, 1)                                              # (bs, n_TVs, T)
    return crit[np.asarray(mask) == 1].mean()
