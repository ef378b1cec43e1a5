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


def pytest_sessionfinish(session, exitstatus):
    if not WORST:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_worst_errors.json"), "w") as f:
            json.dump({k: float(f"{v:.3e}") for k, v in sorted(WORST.items())}, f, indent=1)
    except OSError:
        pass


def split_wg(g):
    """fixture dict -> (weights dict, grads dict) keyed by state_dict names."""
    w = {k[2:]: v for k, v in g.items() if k.startswith("w.")}
    gr = {k[2:]: v for k, v in g.items() if k.startswith("g.")}
    return w, gr
