"""CPU ORACLE for the phoneme_to_articulation hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product path (``artspeech_amd``) never imports it and fails loudly without its HIP library.

It is a from-the-math numpy restatement (explicit forward AND hand-derived backward, no autograd) of
the reference's algorithm.  Every function cites the reference lines it follows (paths relative to the
reference repository root).  Parity status: PINNED -- ``tests/test_oracle_golden.py`` checks every
function against fixtures in ``tests/golden/`` that were produced by running the reference itself
(``tests/golden/make_golden.py``), except ``evenly_spaced_fx`` / ``intersect_semipolar_grid`` (shapely absent => parity unpinned) and the
``vt_tools.metrics.euclidean`` semantic assumed by the area-function fixture.

All functions take/return numpy arrays.  ``dtype`` selects the arithmetic type: float64 (default,
the tight comparator for fp32 kernels) or float32 (what the reference computes in; used for timing).
"""
import numpy as np

# --------------------------------------------------------------------------------------------- utils


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def make_padding_mask(lengths):
    """helpers.py:79-91 -- bool (B, max(lengths)), True on valid frames."""
    lengths = np.asarray(lengths)
    return np.arange(1, int(lengths.max()) + 1)[None, :] <= lengths[:, None]


# --------------------------------------------------------------------------------------------- GRU


def gru_dir_fwd(x, lengths, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one nn.GRU layer on a packed batch (encoder_decoder/models.py:111,136-138).

    PyTorch gate order in the stacked weights is [r; z; n]; h0 = 0; sequence b participates for
    t < lengths[b]; the reverse direction walks t = len_b-1 .. 0; padded outputs are exact zeros.
    x: (B, T, I) -> y: (B, T, H) plus the per-step cache the backward needs.
    """
    B, T, _ = x.shape
    H = w_hh.shape[1]
    dt = x.dtype
    y = np.zeros((B, T, H), dt)
    r_s, z_s, n_s, hn_s = (np.zeros((B, T, H), dt) for _ in range(4))
    h = np.zeros((B, H), dt)
    gi_all = x @ w_ih.T + b_ih  # (B, T, 3H)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        act = lengths > t  # sequences alive at frame t
        if not act.any():
            continue
        gi = gi_all[:, t]
        gh = h @ w_hh.T + b_hh
        r = _sigmoid(gi[:, :H] + gh[:, :H])
        z = _sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        hn = gh[:, 2 * H:]
        n = np.tanh(gi[:, 2 * H:] + r * hn)
        h_new = (1.0 - z) * n + z * h
        h = np.where(act[:, None], h_new, h)  # reverse dir: h stays 0 until t == len_b-1
        y[act, t] = h_new[act]
        r_s[act, t], z_s[act, t], n_s[act, t], hn_s[act, t] = r[act], z[act], n[act], hn[act]
    return y, (r_s, z_s, n_s, hn_s)


def gru_dir_bwd(dy, x, y, cache, lengths, w_ih, w_hh, reverse):
    """BPTT of gru_dir_fwd (SURVEY Appendix A.1 equations).  Returns dx, dw_ih, dw_hh, db_ih, db_hh."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    dt = x.dtype
    r_s, z_s, n_s, hn_s = cache
    dgi = np.zeros((B, T, 3 * H), dt)
    dgh = np.zeros((B, T, 3 * H), dt)
    hprev_all = np.zeros((B, T, H), dt)
    dh = np.zeros((B, H), dt)
    steps = range(T) if reverse else range(T - 1, -1, -1)  # opposite of the forward walk
    for t in steps:
        act = lengths > t
        if not act.any():
            continue
        tp = t + 1 if reverse else t - 1  # frame whose output was this step's h_{prev}
        if 0 <= tp < T:
            hprev = np.where((lengths > tp)[:, None], y[:, tp], 0.0)
        else:
            hprev = np.zeros((B, H), dt)
        r, z, n, hn = r_s[:, t], z_s[:, t], n_s[:, t], hn_s[:, t]
        dht = np.where(act[:, None], dh + dy[:, t], 0.0)
        dn = dht * (1.0 - z)
        dz = dht * (hprev - n)
        dnt = dn * (1.0 - n * n)
        dr = dnt * hn
        g_r = dr * r * (1.0 - r)
        g_z = dz * z * (1.0 - z)
        dgi_t = np.concatenate([g_r, g_z, dnt], axis=1)
        dgh_t = np.concatenate([g_r, g_z, dnt * r], axis=1)
        dh_new = dht * z + dgh_t @ w_hh
        dh = np.where(act[:, None], dh_new, dh)
        dgi[:, t], dgh[:, t], hprev_all[:, t] = dgi_t, dgh_t, hprev
    dx = dgi @ w_ih
    dw_ih = np.einsum("btg,bti->gi", dgi, x)
    dw_hh = np.einsum("btg,bth->gh", dgh, hprev_all)
    return dx, dw_ih, dw_hh, dgi.sum((0, 1)), dgh.sum((0, 1))


# --------------------------------------------------------------------------------------------- head


def layernorm_fwd(x, g, b, eps=1e-5):
    """nn.LayerNorm over the last dim, biased variance, eps 1e-5 (models.py:11,14,17)."""
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mu) * rstd
    return xhat * g + b, (xhat, rstd)


def layernorm_bwd(dout, cache, g):
    xhat, rstd = cache
    dxhat = dout * g
    dx = rstd * (dxhat - dxhat.mean(-1, keepdims=True) - xhat * (dxhat * xhat).mean(-1, keepdims=True))
    red = tuple(range(dout.ndim - 1))
    return dx, (dout * xhat).sum(red), dout.sum(red)


def predictor_fwd(x, p):
    """ArticulatorPredictor.forward (models.py:23-33), WITHOUT the final sigmoid (that is applied
    after stacking, models.py:145).  x (..., in) -> (..., 2, N) pre-activation.  p: dict with the
    state_dict keys of one predictor (``linear.0.weight`` ... ``y_coords.bias``)."""
    a1, c1 = layernorm_fwd(x, p["linear.0.weight"], p["linear.0.bias"])
    z1 = a1 @ p["linear.1.weight"].T + p["linear.1.bias"]
    r1 = np.maximum(z1, 0.0)
    a2, c2 = layernorm_fwd(r1, p["linear.3.weight"], p["linear.3.bias"])
    z2 = a2 @ p["linear.4.weight"].T + p["linear.4.bias"]
    r2 = np.maximum(z2, 0.0)
    a3, c3 = layernorm_fwd(r2, p["linear.6.weight"], p["linear.6.bias"])
    ox = a3 @ p["x_coords.weight"].T + p["x_coords.bias"]
    oy = a3 @ p["y_coords.weight"].T + p["y_coords.bias"]
    out = np.stack([ox, oy], axis=-2)
    return out, (x, a1, c1, z1, a2, c2, z2, a3, c3)


def predictor_bwd(dout, cache, p):
    x, a1, c1, z1, a2, c2, z2, a3, c3 = cache
    F = lambda t: t.reshape(-1, t.shape[-1])  # noqa: E731
    g = {}
    dox, doy = dout[..., 0, :], dout[..., 1, :]
    g["x_coords.weight"] = F(dox).T @ F(a3)
    g["x_coords.bias"] = F(dox).sum(0)
    g["y_coords.weight"] = F(doy).T @ F(a3)
    g["y_coords.bias"] = F(doy).sum(0)
    da3 = dox @ p["x_coords.weight"] + doy @ p["y_coords.weight"]
    dr2, g["linear.6.weight"], g["linear.6.bias"] = layernorm_bwd(da3, c3, p["linear.6.weight"])
    dz2 = dr2 * (z2 > 0)
    g["linear.4.weight"] = F(dz2).T @ F(a2)
    g["linear.4.bias"] = F(dz2).sum(0)
    da2 = dz2 @ p["linear.4.weight"]
    dr1, g["linear.3.weight"], g["linear.3.bias"] = layernorm_bwd(da2, c2, p["linear.3.weight"])
    dz1 = dr1 * (z1 > 0)
    g["linear.1.weight"] = F(dz1).T @ F(a1)
    g["linear.1.bias"] = F(dz1).sum(0)
    da1 = dz1 @ p["linear.1.weight"]
    dx, g["linear.0.weight"], g["linear.0.bias"] = layernorm_bwd(da1, c1, p["linear.0.weight"])
    return dx, g


def _sub(params, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in params.items() if k.startswith(prefix)}


# --------------------------------------------------------------------------------------------- models


def _cast(params, dtype):
    return {k: np.asarray(v).astype(dtype) for k, v in params.items()}


def heads_fwd(lin, params, n_art):
    outs, caches = [], []
    for a in range(n_art):
        o, c = predictor_fwd(lin, _sub(params, f"predictors.{a}."))
        outs.append(o)
        caches.append(c)
    pre = np.stack(outs, axis=2)  # (B, T, A, 2, N)  models.py:141-143
    return _sigmoid(pre), caches  # models.py:145


def heads_bwd(dout, out, caches, params, n_art):
    dpre = dout * out * (1.0 - out)
    grads, dlin = {}, 0.0
    for a in range(n_art):
        dx, g = predictor_bwd(dpre[:, :, a], caches[a], _sub(params, f"predictors.{a}."))
        dlin = dlin + dx
        grads.update({f"predictors.{a}.{k}": v for k, v in g.items()})
    return dlin, grads


def artspeech_fwd(params, x, lengths, n_art, dtype=np.float64, interlayer_scale=None):
    """ArtSpeech.forward (encoder_decoder/models.py:126-145).  interlayer_scale: optional (B, T, 2H) array
    mask / (1 - p) of nn.GRU's inter-layer dropout (training mode); None = dropout 0.
    params: state_dict as numpy; x (B, T) int; lengths (B,) sorted descending.
    Returns out (B, max(len), A, 2, N) and a cache for artspeech_bwd."""
    p = _cast(params, dtype)
    lengths = np.asarray(lengths)
    T = int(lengths.max())
    x = np.asarray(x)[:, :T]
    emb = p["embedding.weight"][x]  # models.py:135
    layer_in, gru_cache = emb, []
    for l in range(2):
        ys, cs = [], []
        for sfx, rev in (("", False), ("_reverse", True)):
            y, c = gru_dir_fwd(layer_in, lengths, p[f"rnn.weight_ih_l{l}{sfx}"], p[f"rnn.weight_hh_l{l}{sfx}"],
                               p[f"rnn.bias_ih_l{l}{sfx}"], p[f"rnn.bias_hh_l{l}{sfx}"], rev)
            ys.append(y)
            cs.append(c)
        out_l = np.concatenate(ys, axis=-1)
        gru_cache.append((layer_in, ys, cs))
        if l == 0 and interlayer_scale is not None:
            out_l = out_l * np.asarray(interlayer_scale, dtype)[:, :T]
        layer_in = out_l
    zlin = layer_in @ p["linear.0.weight"].T + p["linear.0.bias"]  # models.py:113-116,140
    lin = np.maximum(zlin, 0.0)
    out, head_caches = heads_fwd(lin, p, n_art)
    return out, (p, x, lengths, gru_cache, layer_in, zlin, head_caches, out, interlayer_scale)


def artspeech_bwd(dout, cache, n_art):
    p, x, lengths, gru_cache, rnn_out, zlin, head_caches, out, interlayer_scale = cache
    H = p["rnn.weight_hh_l0"].shape[1]
    dlin, grads = heads_bwd(dout, out, head_caches, p, n_art)
    dz = dlin * (zlin > 0)
    F = lambda t: t.reshape(-1, t.shape[-1])  # noqa: E731
    grads["linear.0.weight"] = F(dz).T @ F(rnn_out)
    grads["linear.0.bias"] = F(dz).sum(0)
    dlayer = dz @ p["linear.0.weight"]
    for l in (1, 0):
        layer_in, ys, cs = gru_cache[l]
        dxs = 0.0
        for d, (sfx, rev) in enumerate((("", False), ("_reverse", True))):
            dx, dwi, dwh, dbi, dbh = gru_dir_bwd(dlayer[..., d * H:(d + 1) * H], layer_in, ys[d], cs[d], lengths,
                                                 p[f"rnn.weight_ih_l{l}{sfx}"], p[f"rnn.weight_hh_l{l}{sfx}"], rev)
            dxs = dxs + dx
            grads[f"rnn.weight_ih_l{l}{sfx}"], grads[f"rnn.weight_hh_l{l}{sfx}"] = dwi, dwh
            grads[f"rnn.bias_ih_l{l}{sfx}"], grads[f"rnn.bias_hh_l{l}{sfx}"] = dbi, dbh
        dlayer = dxs
        if l == 1 and interlayer_scale is not None:
            dlayer = dlayer * np.asarray(interlayer_scale, dlayer.dtype)[:, :dlayer.shape[1]]
    demb = np.zeros_like(p["embedding.weight"])
    np.add.at(demb, x.reshape(-1), F(dlayer))
    grads["embedding.weight"] = demb
    return grads


def simple_artspeech_fwd(params, x, n_art, dtype=np.float64, embed_scale=None):
    """SimpleArtSpeech.forward (models.py:75-96); ignores lengths.  embed_scale: optional (B, T, E) array that multiplies the
    embedded frames -- the nn.Dropout of models.py:64,85 in training mode, given its mask (0 or 1 / (1 - p))."""
    p = _cast(params, dtype)
    emb = p["embedding.weight"][np.asarray(x)]
    if embed_scale is not None:
        emb = emb * np.asarray(embed_scale, dtype)
    zlin = emb @ p["linear.0.weight"].T + p["linear.0.bias"]
    lin = np.maximum(zlin, 0.0)
    out, head_caches = heads_fwd(lin, p, n_art)
    return out, (p, np.asarray(x), emb, zlin, head_caches, out, embed_scale)


def simple_artspeech_bwd(dout, cache, n_art):
    p, x, emb, zlin, head_caches, out, embed_scale = cache
    dlin, grads = heads_bwd(dout, out, head_caches, p, n_art)
    dz = dlin * (zlin > 0)
    F = lambda t: t.reshape(-1, t.shape[-1])  # noqa: E731
    grads["linear.0.weight"] = F(dz).T @ F(emb)
    grads["linear.0.bias"] = F(dz).sum(0)
    demb_frames = dz @ p["linear.0.weight"]
    if embed_scale is not None:
        demb_frames = demb_frames * np.asarray(embed_scale, demb_frames.dtype)
    demb = np.zeros_like(p["embedding.weight"])
    np.add.at(demb, x.reshape(-1), F(demb_frames))
    grads["embedding.weight"] = demb
    return grads


# --------------------------------------------------------------------------------------------- losses / metrics


def euclidean_distance(outputs, targets):
    """EuclideanDistance("none") (phoneme_to_articulation/metrics.py:17-24): (..., 2, N) -> (..., N)."""
    dx = outputs[..., 0, :] - targets[..., 0, :]
    dy = outputs[..., 1, :] - targets[..., 1, :]
    return np.sqrt(dx * dx + dy * dy)


def masked_euclid_loss(outputs, targets, lengths, dtype=np.float64):
    """Masked mean of the per-point Euclidean distance over valid frames
    (train_phoneme_to_articulation.py:86-90).  Returns (loss, d loss / d outputs)."""
    o, t = np.asarray(outputs, dtype), np.asarray(targets, dtype)
    mask = make_padding_mask(lengths)[:, :o.shape[1]]
    dist = euclidean_distance(o, t)  # (B, T, A, N)
    count = mask.sum() * dist.shape[2] * dist.shape[3]
    loss = (dist * mask[:, :, None, None]).sum() / count
    with np.errstate(invalid="ignore", divide="ignore"):
        inv = np.where(mask[:, :, None, None], 1.0 / (dist * count), 0.0)  # NaN/inf at dist==0 like the reference
    grad = np.stack([(o[..., 0, :] - t[..., 0, :]) * inv, (o[..., 1, :] - t[..., 1, :]) * inv], axis=-2)
    return loss, grad


def mean_p2cp(u, v, dtype=np.float64):
    """MeanP2CPDistance("none") (phoneme_to_articulation/metrics.py:33-46) with the distance matrix
    computed by direct differences.  u (*, N, 2), v (*, M, 2) -> (*)."""
    u, v = np.asarray(u, dtype), np.asarray(v, dtype)
    d = np.sqrt(((u[..., :, None, :] - v[..., None, :, :]) ** 2).sum(-1))
    return (d.min(-1).sum(-1) / u.shape[-2] + d.min(-2).sum(-1) / v.shape[-2]) / 2


def mean_p2cp_mm(u, v):
    """Same metric through the matmul expansion torch.cdist switches to when N or M > 25
    (|u|^2 + |v|^2 - 2 u.v, clamp at 0, sqrt) in float32: explains the reference's own ~1e-3
    relative deviation from the direct formula."""
    u, v = np.asarray(u, np.float32), np.asarray(v, np.float32)
    un = (u * u).sum(-1, keepdims=True)
    vn = (v * v).sum(-1, keepdims=True)
    u_ = np.concatenate([-2 * u, un, np.ones_like(un)], -1)
    v_ = np.concatenate([v, np.ones_like(vn), vn], -1)
    d = np.sqrt(np.maximum(u_ @ np.swapaxes(v_, -1, -2), 0))
    return (d.min(-1).sum(-1) / u.shape[-2] + d.min(-2).sum(-1) / v.shape[-2]) / 2


def p2cp_distance(outputs, targets, dtype=np.float64):
    """metrics.py:38-52 (root): (B, T, A, 2, N) x2 -> (B, T, A)."""
    return mean_p2cp(np.swapaxes(outputs, -1, -2), np.swapaxes(targets, -1, -2), dtype)


def p2cp_distance_mm(outputs, targets, lengths, to_mm, dtype=np.float64):
    """P2CPDistance.forward (encoder_decoder/metrics.py:18-26): per-utterance mean over valid frames
    and articulators of P2CP * to_mm, then the mean over the batch."""
    p = p2cp_distance(outputs, targets, dtype) * to_mm
    return np.mean([p[i, :l].mean() for i, l in enumerate(lengths)])


def euclidean_distance_metric(outputs, targets):
    """metrics.py:54-68 (root): mean over points -> (B, T, A)."""
    return euclidean_distance(outputs, targets).mean(-1)


def pearsons_correlation(outputs, targets, eps=1e-5):
    """metrics.py:9-35 (root).  NOTE the reference centres the x TARGETS with the x OUTPUTS' mean
    (metrics.py:22) -- reproduced as is; y is centred correctly (metrics.py:30)."""
    xo, yo = outputs[:, :, :, 0, :], outputs[:, :, :, 1, :]
    xt, yt = targets[:, :, :, 0, :], targets[:, :, :, 1, :]
    vxo = xo - xo.mean(1, keepdims=True)
    vxt = xt - xo.mean(1, keepdims=True)
    vyo = yo - yo.mean(1, keepdims=True)
    vyt = yt - yt.mean(1, keepdims=True)
    xc = (vxo * vxt).sum(1) / (np.sqrt((vxo ** 2).sum(1)) * np.sqrt((vxt ** 2).sum(1)) + eps)
    yc = (vyo * vyt).sum(1) / (np.sqrt((vyo ** 2).sum(1)) * np.sqrt((vyt ** 2).sum(1)) + eps)
    return xc, yc


# --------------------------------------------------------------------------------------------- tract variables

ART_SLICES = {  # tract_variables.py:13-20
    "tongue-tip": (30, 45), "tongue-body": (10, 30), "upper-incisor": (25, 50),
    "hard-palate": (0, 25), "soft-palate": (35, 50), "velum": (0, 15),
}
TV_NAMES = ("LA", "TTCD", "TBCD", "VEL")


def _tv(arr1, arr2):
    """_calculate_TV (tract_variables.py:23-35): min over dim 0 first, then over dim 0 of the result;
    argmin = first minimum in each pass."""
    d = np.sqrt(((arr1[:, None, :] - arr2[None, :, :]) ** 2).sum(-1))
    arg0 = d.argmin(0)
    min0 = d.min(0)
    j = int(min0.argmin())
    i = int(arg0[j])
    return min0[j], i, j


def tract_variables(frame, articulators, dtype=np.float64):
    """calculate_vocal_tract_variables (tract_variables.py:73-125) for ONE frame in model-output
    layout: frame (A, 2, N), articulators = list of names in channel order.
    Returns values (4,), poc1 (4, 2), poc2 (4, 2), idx (4, 2) [indices into arr1/arr2]."""
    pts = {a: np.asarray(frame[i], dtype).T for i, a in enumerate(articulators)}  # (N, 2) each
    s = lambda k: slice(*ART_SLICES[k])  # noqa: E731
    pairs = (
        (pts["lower-lip"], pts["upper-lip"]),
        (pts["tongue"][s("tongue-tip")], pts["upper-incisor"][s("upper-incisor")]),
        (pts["tongue"][s("tongue-body")],
         np.concatenate([pts["upper-incisor"][s("hard-palate")], pts["soft-palate-midline"][s("soft-palate")]], 0)),
        (pts["soft-palate-midline"][s("velum")], pts["pharynx"]),
    )
    vals, p1, p2, idx = [], [], [], []
    for a1, a2 in pairs:
        v, i, j = _tv(a1, a2)
        vals.append(v), p1.append(a1[i]), p2.append(a2[j]), idx.append((i, j))
    return np.array(vals), np.array(p1), np.array(p2), np.array(idx)


# --------------------------------------------------------------------------------------------- area function


def area_function(internal_wall, external_wall, alpha=np.pi, beta=2.0):
    """area_function (area_function.py:124-142) + mid_point (:113-121), float64.
    walls (Nw, 2) -> dists (Nw,), fx (Nw,)."""
    a, b = np.asarray(internal_wall, np.float64), np.asarray(external_wall, np.float64)
    mid = np.minimum(a, b) + np.abs(a - b) / 2
    radius = np.sqrt(((a - b) ** 2).sum(-1)) / 2
    fx = alpha * radius ** beta
    seg = np.sqrt(((mid[1:] - mid[:-1]) ** 2).sum(-1))
    dists = np.concatenate([[0.0], np.cumsum(seg)])  # sequential running sum, as the reference's loop
    return dists, fx


def evenly_spaced_fx(x, fx, n_samples=200):
    """evenly_spaced_fx (area_function.py:145-159): vertical line at each of n_samples even abscissae
    intersected with the polyline (x, fx) == piecewise-linear interpolation for increasing x.
    PARITY UNPINNED (shapely is absent in the build container, the reference function cannot run)."""
    xs = np.linspace(x[0], x[-1], n_samples)
    return np.stack([xs, np.interp(xs, x, fx)])


def build_semipolar_grid(center, theta_rad, omega_rad, linear_step, polar_step_rad, grid_res=50):
    """build_semipolar_grid (area_function.py:31-110): Maeda's semipolar grid lines, each sampled at
    grid_res points from the inner to the outer end.  Order: larynx lines reversed, polar lines
    reversed, mouth lines."""
    center = np.asarray(center, np.float64)

    def rot(p, ang):  # area_function.py:12-28
        m = np.array([[np.cos(ang), np.sin(ang)], [-np.sin(ang), np.cos(ang)]])
        return m @ p

    xs = np.arange(0.0, -0.5, -linear_step)
    mouth_int = [rot(np.array([x, 0.0]), theta_rad) + center for x in xs]
    mouth_ext = [rot(np.array([x, -0.4]), theta_rad) + center for x in xs]
    ys = np.arange(0.0, 0.5, linear_step)
    lar_int = [rot(np.array([0.0, y]), omega_rad) + center for y in ys]
    lar_ext = [rot(np.array([0.4, y]), omega_rad) + center for y in ys]
    angles = np.arange(theta_rad - polar_step_rad, -(np.pi / 2) + omega_rad, -polar_step_rad)
    pol_ext = [rot(np.array([0.0, -0.4]), a) + center for a in angles]
    pol_int = [center.copy() for _ in angles]
    lines = []
    for seq_int, seq_ext in ((lar_int[::-1], lar_ext[::-1]), (pol_int[::-1], pol_ext[::-1]), (mouth_int, mouth_ext)):
        for pi, pe in zip(seq_int, seq_ext):
            lines.append(np.stack([np.linspace(pi[0], pe[0], grid_res), np.linspace(pi[1], pe[1], grid_res)], 1))
    return np.array(lines)


def _polyline_intersections(line, wall):
    """Intersection points of two polylines ((G, 2) and (W, 2) float64), ordered along `line`.  Segment pair test in
    float64: p + t r = q + u s with r x s != 0; a hit at a junction of two consecutive segments is attributed to the
    segment where the parameter is 0 (half-open [0, 1) except on the last segment), so no point is reported twice."""
    pts = []
    G, W = len(line), len(wall)
    for gs in range(G - 1):
        p, r = line[gs], line[gs + 1] - line[gs]
        for ws in range(W - 1):
            q, s = wall[ws], wall[ws + 1] - wall[ws]
            den = r[0] * s[1] - r[1] * s[0]
            if den == 0.0:
                continue  # parallel (collinear overlaps are not points; the reference would fail on them)
            dq = q - p
            t = (dq[0] * s[1] - dq[1] * s[0]) / den
            u = (dq[0] * r[1] - dq[1] * r[0]) / den
            t_ok = 0.0 <= t and (t < 1.0 or (gs == G - 2 and t <= 1.0))
            u_ok = 0.0 <= u and (u < 1.0 or (ws == W - 2 and u <= 1.0))
            if t_ok and u_ok:
                pts.append((gs + t, ws, p + t * r))
    pts.sort(key=lambda e: (e[0], e[1]))
    return np.array([e[2] for e in pts]).reshape(-1, 2)


def _argmin_matrix(a, b):
    """argmatrix(distance_matrix(a, b), "min") (area_function.py:160-172): first minimum in row-major order."""
    d = np.sqrt(((a[:, None, :] - b[None, :, :]) ** 2).sum(-1))
    k = int(np.argmin(d))
    return k // d.shape[1], k % d.shape[1]


def intersect_semipolar_grid(internal_wall, external_wall, semipolar_grid):
    """intersect_semipolar_grid (area_function.py:175-223): for every grid line that touches a wall, the wall's intersection
    point closest to the other wall's intersections (or, when the other wall is not crossed, closest to the EXTERNAL wall's
    end points -- also in the external branch, :211 -- whose nearer end then stands in for the missing point).
    PARITY UNPINNED: the reference computes the intersections with shapely/GEOS (absent here); this restates them with the
    float64 segment test above, points ordered along the grid line (the order only decides exact distance ties).
    Returns (flags (L,), internal (L, 2), external (L, 2)); flags bit 0 / 1 = internal / external wall crossed; lines with
    flags 0 are the ones the reference skips."""
    internal_wall, external_wall = np.asarray(internal_wall, np.float64), np.asarray(external_wall, np.float64)
    grid = np.asarray(semipolar_grid, np.float64)
    L = len(grid)
    flags, pi, pe = np.zeros(L, np.int32), np.zeros((L, 2)), np.zeros((L, 2))
    default = np.array([external_wall[0], external_wall[-1]])
    for l in range(L):
        li, le = _polyline_intersections(grid[l], internal_wall), _polyline_intersections(grid[l], external_wall)
        ic, ec = len(li) > 0, len(le) > 0
        flags[l] = int(ic) | (int(ec) << 1)
        if ic:
            i_min, j_min = _argmin_matrix(li, le if ec else default)
            pi[l] = li[i_min]
            if not ec:
                pe[l] = external_wall[-j_min]
        if ec:
            i_min, j_min = _argmin_matrix(le, li if ic else default)
            pe[l] = le[i_min]
            if not ic:
                pi[l] = internal_wall[-j_min]
    return flags, pi, pe
