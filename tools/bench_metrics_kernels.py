"""The three small kernels north_star names besides the recurrence -- point-to-curve distance (P2CP), tract variables,
vocal-tract area function (+ the evenly spaced resampling) -- alone, in a driver-style loop at B=32, T=200 through the C ABI
(no per-call allocation, no host work between launches): HIP-event time per launch, algorithmic bytes (inputs read
once + outputs written once) and GB/s against the 8 TB/s HBM figure.  The same command runs under rocprofv3 (program directly
after `--`) for profiles/r03_metrics_kernels.*:

    python3 tools/bench_metrics_kernels.py [iters] [--json path]

reference: phoneme_to_articulation/metrics.py:38-46 (P2CP), tract_variables.py:23-35 (TVs), area_function.py:124-142, 145-159."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artspeech_amd import _lib  # noqa: E402
from artspeech_amd.tract_variables import _spec  # noqa: E402

HBM_PEAK_GBS = 8000.0
ARTS = sorted(["arytenoid-cartilage", "epiglottis", "lower-incisor", "lower-lip", "pharynx", "soft-palate-midline", "thyroid-cartilage",
               "tongue", "upper-incisor", "upper-lip", "vocal-folds"])


def main(iters=None, json_path=None, log=print):
    if iters is None:
        iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 50
        json_path = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    L = _lib.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, T, A, N, NW, NS = 32, 200, 11, 50, 100, 200
    frames = B * T
    out = torch.rand(B, T, A, 2, N, device=dev)
    tgt = torch.rand(B, T, A, 2, N, device=dev)
    st = _lib.stream_ptr()
    cases = {}

    # ---- P2CP over every (b, t, a) tile of 50 x 50 points: the (*, 2, N) storage read in place (point stride 1, xy stride N)
    p2cp = torch.empty(B, T, A, device=dev)
    def run_p2cp():
        _lib.check(L.as_p2cp_fwd(_lib.ptr(out), 2 * N, 1, N, N, _lib.ptr(tgt), 2 * N, 1, N, N, frames * A, _lib.ptr(p2cp), st), "as_p2cp_fwd")
    cases["as_p2cp_fwd (p2cp_kernel)"] = (run_p2cp, 2 * frames * A * 2 * N * 4 + frames * A * 4)

    # ---- tract variables of every frame (LA, TTCD, TBCD, VEL): 6 of the 11 articulators are read
    spec = _spec(ARTS, N).to(dev)
    tv_v = torch.empty(frames, 4, device=dev)
    tv_p1, tv_p2 = torch.empty(frames, 4, 2, device=dev), torch.empty(frames, 4, 2, device=dev)
    tv_i = torch.empty(frames, 4, 2, dtype=torch.int32, device=dev)
    contours = out.view(frames, A, 2, N)
    def run_tv():
        _lib.check(L.as_tract_variables_fwd(_lib.ptr(contours), frames, A, N, _lib.ptr(spec), 4, _lib.ptr(tv_v), _lib.ptr(tv_p1),
                                            _lib.ptr(tv_p2), _lib.ptr(tv_i), st), "as_tract_variables_fwd")
    tv_in = frames * (50 + 50 + 15 + 25 + 20 + 25 + 15 + 15 + 50) * 2 * 4      # the slices of ART_SLICES, x and y
    cases["as_tract_variables_fwd (tv_kernel)"] = (run_tv, tv_in + frames * 4 * (4 + 8 + 8 + 8))

    # ---- area function of every frame: two fp64 walls of 100 points -> dists, fx (fp64); then 200 evenly spaced samples
    air = torch.rand(frames, 2, 2, NW, device=dev, dtype=torch.float64)     # air_column file layout (2 walls, 2, 100)
    air[:, :, 0] = torch.cumsum(air[:, :, 0], dim=-1)                        # increasing x: a tube, not a scribble
    internal, external = air[:, 0], air[:, 1]
    dists = torch.empty(frames, NW, device=dev, dtype=torch.float64)
    fx = torch.empty(frames, NW, device=dev, dtype=torch.float64)
    def run_area():
        _lib.check(L.as_area_function_fwd(_lib.ptr(internal), _lib.ptr(external), air.stride(0), air.stride(3), air.stride(2), frames, NW,
                                          3.141592653589793, 2.0, _lib.ptr(dists), _lib.ptr(fx), st), "as_area_function_fwd")
    cases["as_area_function_fwd (area_kernel)"] = (run_area, frames * (2 * NW * 2 * 8 + 2 * NW * 8))
    res = torch.empty(frames, 2, NS, device=dev)
    def run_resample():
        _lib.check(L.as_evenly_spaced_fx(_lib.ptr(dists), _lib.ptr(fx), frames, NW, NS, _lib.ptr(res), st), "as_evenly_spaced_fx")
    cases["as_evenly_spaced_fx (resample_kernel)"] = (run_resample, frames * (2 * NW * 8 + 2 * NS * 4))

    report = {}
    log(f"--- B={B} T={T} ({frames} frames), {iters} launches each, HIP events on the launch stream")
    for name, (fn, nbytes) in cases.items():
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / iters
        gbs = nbytes / (us * 1e-6) / 1e9
        report[name] = {"us_per_launch": round(us, 2), "algorithmic_bytes": nbytes, "achieved_GBs": round(gbs, 1),
                        "frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 4), "frames_per_s": round(frames / (us * 1e-6), 0)}
        log(f"{name:42s} {us:8.2f} us   {nbytes / 1e6:7.2f} MB   {gbs:8.1f} GB/s   {gbs / HBM_PEAK_GBS:6.3f} of 8 TB/s")
    assert torch.isfinite(p2cp).all() and torch.isfinite(tv_v).all() and torch.isfinite(dists).all() and torch.isfinite(res).all()
    if json_path:
        with open(json_path, "w") as f:
            json.dump(report, f, indent=1)
    return report


if __name__ == "__main__":
    main()
