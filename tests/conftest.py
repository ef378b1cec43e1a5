import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


# worst observed gradient error per check (max |a - b| / max |b|), written to gpurun_out/ at the end of a -m gpu session so
# that regressions show as numbers, not only as pass / fail
WORST = {}


def assert_grad_close(got, ref, what, rtol=1e-4, atol_frac=1e-5, atol_abs=0.0):
    """Element-wise gradient check: |got - ref| <= rtol * |ref| + atol_frac * max|ref| for every element (the absolute floor
    scales with the tensor because fp32 accumulation error does not shrink with the element it lands on).  Measured worst
    case over all fixtures is 2.5e-6 of max|ref| (profiles/r02_parity_worst_errors.json); the message carries the observed figure."""
    a, b = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(np.abs(b).max()), 1e-30)
    err = np.abs(a - b)
    ratio = float((err / (rtol * np.abs(b) + atol_frac * scale + atol_abs)).max())
    rel_of_max = float(err.max()) / scale
    WORST[what] = max(WORST.get(what, 0.0), rel_of_max)
    assert ratio <= 1.0, (f"{what}: |got - ref| exceeds {rtol:g} * |ref| + {atol_frac:g} * max|ref| by x{ratio:.2f}; "
                          f"max|got - ref| / max|ref| = {rel_of_max:.2e}")
    return rel_of_max


FLIPS = {}   # test label -> [(where, index, z)]: every ReLU decision a parity test took from the device (written at session end)


def oracle_gradients_with_the_devices_relu_decisions(got, o_dout, cache, n_art, tau=5e-6, max_flips=8, label=None):
    """The fp64 oracle and an fp32 implementation can disagree about `z > 0` where the oracle's pre-activation z lies within
    rounding of zero; one such element changes the gradients of its head by about one frame's term and everything upstream
    with it (stock PyTorch fp32 against fp64 shows it in 9 of 40 seeded draws of a 450-frame batch: DESIGN section 2).  This
    returns the oracle's gradients for the decisions the device took: candidates are the ReLU inputs of the heads and of the
    trunk with |z| < tau on valid frames; candidate (frame, feature j) feeds row j of the weight gradient of its Linear, and
    its decision is flipped only if that row is off by more than 2e-5 of the tensor's maximum and the flip at least halves
    the row's error.  Returns (gradients, flips) with flips = [(where, index, z)].  cache is O.artspeech_fwd's."""
    from oracle import artspeech_oracle as O
    p, x, lengths, gru_cache, rnn_out, zlin, head_caches, out, _ = cache
    valid = np.arange(out.shape[1])[None, :] < np.asarray(lengths)[:, None]
    dpre = o_dout * out * (1.0 - out)
    flips = []

    def row_err(key, g, j):
        return float(np.abs(np.asarray(got[key][j], np.float64) - g[j]).max()) / max(float(np.abs(g).max()), 1e-30)

    def greedy(zs, keys, grads, where):
        cand = [(abs(float(z[i])), k, i) for k, z in enumerate(zs) for i in zip(*np.nonzero((np.abs(z) < tau) & valid[..., None]))]
        g = None
        for _, k, i in sorted(cand)[:64]:
            if len(flips) >= max_flips:
                break
            g = grads() if g is None else g
            e = row_err(keys[k], g[k], i[-1])
            if e < 2e-5:
                continue
            keep = float(zs[k][i])
            zs[k][i] = -abs(keep) if keep > 0 else max(abs(keep), 1e-300)
            g2 = grads()
            if row_err(keys[k], g2[k], i[-1]) < 0.5 * e:
                g = g2
                flips.append((f"{where}, ReLU {k + 1}", tuple(int(j) for j in i), keep))
            else:
                zs[k][i] = keep

    for a in range(n_art):
        def grads(a=a):
            g = O.predictor_bwd(dpre[:, :, a], head_caches[a], O._sub(p, f"predictors.{a}."))[1]
            return g["linear.1.weight"], g["linear.4.weight"]
        greedy([head_caches[a][3], head_caches[a][6]], [f"predictors.{a}.linear.1.weight", f"predictors.{a}.linear.4.weight"], grads, f"head {a}")
    greedy([zlin], ["linear.0.weight"], lambda: (O.artspeech_bwd(o_dout, cache, n_art)["linear.0.weight"],), "trunk")
    if label is not None:
        FLIPS[label] = [(w, list(i), float(z)) for w, i, z in flips]
    return O.artspeech_bwd(o_dout, cache, n_art), flips


def pytest_sessionfinish(session, exitstatus):
    if not WORST:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_worst_errors.json"), "w") as f:
            json.dump({k: float(f"{v:.3e}") for k, v in sorted(WORST.items())}, f, indent=1)
        # the ReLU decisions the gradient checks took from the device instead of the fp64 oracle: where, element, oracle input z
        with open(os.path.join(out, "parity_relu_flips.json"), "w") as f:
            json.dump({"tests": FLIPS, "count": sum(len(v) for v in FLIPS.values()),
                       "max_abs_z": max([abs(z) for v in FLIPS.values() for _, _, z in v], default=0.0)}, f, indent=1)
    except OSError:
        pass


def split_wg(g):
    """fixture dict -> (weights dict, grads dict) keyed by state_dict names."""
    w = {k[2:]: v for k, v in g.items() if k.startswith("w.")}
    gr = {k[2:]: v for k, v in g.items() if k.startswith("g.")}
    return w, gr
