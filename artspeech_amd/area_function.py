"""Vocal-tract area function on MI355X (reference: area_function.py:113-142), float64 like the reference.

``area_function(internal_wall, external_wall, alpha, beta)`` keeps the reference's numpy-in /
numpy-out contract for one wall pair; ``area_function_batched`` takes the air-column layout
(frames, 2 walls, 2, Nw) (phoneme_recognition/datasets.py:152, scripts/shape_to_air_column.py:81)
and runs one wave per frame."""
import os

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


def evenly_spaced_fx_batched(x, fx, n_samples=200):
    """x, fx: (frames, Nw) float64 on the GPU with x increasing along dim 1 -> (frames, 2, n_samples) float32
    (abscissae, resampled values)."""
    _lib.require_gpu(x, "x")
    x, fx = x.to(torch.float64).contiguous(), fx.to(torch.float64).contiguous()
    frames, nw = x.shape
    out = torch.empty((frames, 2, n_samples), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().as_evenly_spaced_fx(_lib.ptr(x), _lib.ptr(fx), frames, nw, n_samples, _lib.ptr(out), _lib.stream_ptr()),
               "as_evenly_spaced_fx")
    return out


def evenly_spaced_fx(x, fx, n_samples=200):
    """Resample the area function at n_samples evenly spaced abscissae (reference area_function.py:145-159: vertical
    line x = x_s intersected with the polyline (x, fx)).  Returns a (2, n_samples) float32 tensor like the reference."""
    if not torch.cuda.is_available():
        raise RuntimeError("artspeech_amd.area_function needs an MI355X device; there is no CPU path")
    xt = torch.as_tensor(np.asarray(x, dtype=np.float64)).cuda()[None]
    ft = torch.as_tensor(np.asarray(fx, dtype=np.float64)).cuda()[None]
    return evenly_spaced_fx_batched(xt, ft, n_samples)[0].cpu()


def _rotate(point, ang_rad):
    rot = np.array([[np.cos(ang_rad), np.sin(ang_rad)], [-np.sin(ang_rad), np.cos(ang_rad)]])
    return rot @ point


def build_semipolar_grid(center, theta_rad, omega_rad, linear_step, polar_step_rad, grid_res=50):
    """Maeda's semipolar grid (reference area_function.py:31-110): larynx lines (reversed), polar lines (reversed), mouth
    lines; every line sampled at grid_res points from its inner to its outer end.  Host-side set-up geometry (a few
    thousand points, built once): plain numpy, float64, returns (n_lines, grid_res, 2)."""
    center = np.asarray(center, dtype=np.float64)
    xs = np.arange(0., -0.5, -linear_step)
    ys = np.arange(0., 0.5, linear_step)
    angles = np.arange(theta_rad - polar_step_rad, -(np.pi / 2) + omega_rad, -polar_step_rad)
    mouth = [(_rotate(np.array([x, 0.]), theta_rad) + center, _rotate(np.array([x, -0.4]), theta_rad) + center) for x in xs]
    larynx = [(_rotate(np.array([0., y]), omega_rad) + center, _rotate(np.array([0.4, y]), omega_rad) + center) for y in ys]
    polar = [(center.copy(), _rotate(np.array([0., -0.4]), a) + center) for a in angles]
    lines = []
    for p_int, p_ext in larynx[::-1] + polar[::-1] + mouth:
        lines.append(np.stack([np.linspace(p_int[0], p_ext[0], grid_res), np.linspace(p_int[1], p_ext[1], grid_res)], axis=1))
    return np.array(lines)


def save_air_column(filepath, internal_wall, external_wall):
    """Write one frame's air column in the reference's on-disk layout: ``<frame>.npy`` holding (2 walls, 2 coordinates,
    Nw points) float64, internal wall first (generate_vocal_tract_shape_v2.py:425-431, scripts/shape_to_air_column.py:77-83).
    Walls are (Nw, 2) arrays as ``area_function`` takes them."""
    air_column = np.array([np.asarray(internal_wall).T, np.asarray(external_wall).T])
    np.save(filepath, air_column)
    return air_column


def load_air_columns(directory, frame_ids, device=None):
    """Read ``<directory>/<frame>.npy`` air columns (phoneme_recognition/datasets.py:151-156) into one (frames, 2, 2, Nw)
    float64 tensor -- the input layout of ``area_function_batched``."""
    stack = np.stack([np.load(os.path.join(directory, f"{frame_id}.npy")) for frame_id in frame_ids]).astype(np.float64)
    t = torch.from_numpy(stack)
    return t.to(device) if device is not None else t


def intersect_semipolar_grid_batched(air_column, semipolar_grid):
    """air_column (frames, 2, 2, Nw) float64 on the GPU (internal wall first), semipolar_grid (n_lines, grid_res, 2) ->
    flags int32 (frames, n_lines) (bit 0 / 1: internal / external wall crossed; 0 = line skipped by the reference; bit 2:
    more than 16 crossings of one wall), internal / external points float64 (frames, n_lines, 2).  One wave per
    (frame, grid line); see ``intersect_semipolar_grid`` for the selection rule."""
    _lib.require_gpu(air_column, "air_column")
    air = air_column.to(torch.float64).contiguous()
    grid = torch.as_tensor(np.asarray(semipolar_grid, dtype=np.float64)).to(air.device).contiguous()
    frames, _, _, n_pts = air.shape
    n_lines, grid_res, _ = grid.shape
    flags = torch.empty((frames, n_lines), dtype=torch.int32, device=air.device)
    p_int = torch.empty((frames, n_lines, 2), dtype=torch.float64, device=air.device)
    p_ext = torch.empty_like(p_int)
    _lib.check(_lib.lib().as_intersect_semipolar_grid(_lib.ptr(air), _lib.ptr(grid), frames, n_pts, n_lines, grid_res, _lib.ptr(flags),
                                                      _lib.ptr(p_int), _lib.ptr(p_ext), _lib.stream_ptr()), "as_intersect_semipolar_grid")
    return flags, p_int, p_ext


def intersect_semipolar_grid(internal_wall, external_wall, semipolar_grid):
    """Section points of one vocal tract: for every grid line that crosses a wall, the crossing closest to the other wall's
    crossings; a wall that the line does not cross contributes its end point nearer to ... the external wall's ends
    (reference area_function.py:175-223, quirk of :211 included).  Walls (Nw, 2) -> (internal (K, 2), external (K, 2))
    numpy float64 over the K lines with contact, in grid order -- the inputs of ``area_function``."""
    if not torch.cuda.is_available():
        raise RuntimeError("artspeech_amd.area_function needs an MI355X device; there is no CPU path")
    air = torch.from_numpy(np.array([np.asarray(internal_wall, dtype=np.float64).T, np.asarray(external_wall, dtype=np.float64).T]))
    flags, p_int, p_ext = intersect_semipolar_grid_batched(air[None].cuda(), semipolar_grid)
    keep = (flags[0] & 3).cpu().numpy() != 0
    return p_int[0].cpu().numpy()[keep], p_ext[0].cpu().numpy()[keep]
