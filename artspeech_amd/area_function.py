"""Vocal-tract area function on MI355X (reference: area_function.py:113-142), float64 like the reference.

``area_function(internal_wall, external_wall, alpha, beta)`` keeps the reference's numpy-in /
numpy-out contract for one wall pair; ``area_function_batched`` takes the air-column layout
(frames, 2 walls, 2, Nw) (phoneme_recognition/datasets.py:152, scripts/shape_to_air_column.py:81)
and runs one wave per frame."""
import numpy as np
import torch

from . import _lib


def area_function_batched(air_column, alpha=np.pi, beta=2.0):
    """air_column (frames, 2, 2, Nw) float64 on the GPU: [:, 0] internal wall, [:, 1] external wall,
    coordinates (x, y) on dim 2.  Returns dists (frames, Nw), fx (frames, Nw)."""
    _lib.require_gpu(air_column, "air_column")
    L = _lib.lib()
    ac = air_column.to(torch.float64).contiguous()
    frames, walls, two, nw = ac.shape
    assert walls == 2 and two == 2
    dists = torch.empty((frames, nw), dtype=torch.float64, device=ac.device)
    fx = torch.empty_like(dists)
    internal, external = ac[:, 0], ac[:, 1]
    _lib.check(L.as_area_function_fwd(_lib.ptr(internal), _lib.ptr(external), ac.stride(0), ac.stride(3), ac.stride(2),
                                      frames, nw, float(alpha), float(beta), _lib.ptr(dists), _lib.ptr(fx),
                                      _lib.stream_ptr()), "as_area_function_fwd")
    return dists, fx


def area_function(internal_wall, external_wall, alpha=np.pi, beta=2.):
    """internal_wall, external_wall: (Nw, 2) arrays -> (dists (Nw,), fx (Nw,)) numpy float64."""
    internal_wall, external_wall = np.asarray(internal_wall), np.asarray(external_wall)
    assert internal_wall.shape == external_wall.shape
    if not torch.cuda.is_available():
        raise RuntimeError("artspeech_amd.area_function needs an MI355X device; there is no CPU path")
    ac = torch.from_numpy(np.stack([internal_wall.T, external_wall.T])[None].astype(np.float64)).cuda()
    dists, fx = area_function_batched(ac, alpha, beta)
    return dists[0].cpu().numpy(), fx[0].cpu().numpy()
