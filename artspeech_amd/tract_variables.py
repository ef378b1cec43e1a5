"""Vocal-tract variables on MI355X (reference: tract_variables.py).

LA / TTCD / TBCD / VEL = minimum pairwise distance between two articulator slices plus the two
closest points; LP / TTCL / TBCL / GLO are ``None`` as in the reference (:104-123).  The batched entry
point runs one wave per (frame, variable) for a whole (frames, A, 2, N) contour tensor; the per-frame
dict API of the reference is kept on top of it."""
import torch

from . import _lib

LOWER_LIP = "lower-lip"
PHARYNX = "pharynx"
SOFT_PALATE_MIDLINE = "soft-palate-midline"
TONGUE = "tongue"
UPPER_LIP = "upper-lip"
UPPER_INCISOR = "upper-incisor"

ART_SLICES = {  # reference tract_variables.py:13-20
    "tongue-tip": (30, 45),
    "tongue-body": (10, 30),
    "upper-incisor": (25, 50),
    "hard-palate": (0, 25),
    "soft-palate": (35, 50),
    "velum": (0, 15),
}
TV_NAMES = ("LA", "TTCD", "TBCD", "VEL")
REQUIRED_ARTICULATORS = (LOWER_LIP, PHARYNX, SOFT_PALATE_MIDLINE, TONGUE, UPPER_LIP, UPPER_INCISOR)


def _spec(articulators, n_samples):
    """int32 [4][3][3] {channel, start, end}: arr1, arr2 part 1, arr2 part 2 (reference :38-70)."""
    ch = {a: i for i, a in enumerate(articulators)}
    missing = [a for a in REQUIRED_ARTICULATORS if a not in ch]
    if missing:
        raise KeyError(missing[0])
    full = (0, n_samples)
    none = (-1, 0, 0)
    return torch.tensor([
        [(ch[LOWER_LIP], *full), (ch[UPPER_LIP], *full), none],
        [(ch[TONGUE], *ART_SLICES["tongue-tip"]), (ch[UPPER_INCISOR], *ART_SLICES["upper-incisor"]), none],
        [(ch[TONGUE], *ART_SLICES["tongue-body"]), (ch[UPPER_INCISOR], *ART_SLICES["hard-palate"]),
         (ch[SOFT_PALATE_MIDLINE], *ART_SLICES["soft-palate"])],
        [(ch[SOFT_PALATE_MIDLINE], *ART_SLICES["velum"]), (ch[PHARYNX], *full), none],
    ], dtype=torch.int32)


def tract_variables_batched(contours, articulators):
    """contours (frames, A, 2, N) on the GPU in model-output layout, articulators = channel names.
    Returns values (frames, 4), poc1 (frames, 4, 2), poc2 (frames, 4, 2), idx int32 (frames, 4, 2) with
    the variables ordered LA, TTCD, TBCD, VEL."""
    _lib.require_gpu(contours, "contours")
    L = _lib.lib()
    frames, A, two, N = contours.shape
    if N < 50:
        raise IndexError(f"tract variables slice contours up to point 50, got n_samples={N}")
    c = contours.contiguous().float()
    spec = _spec(list(articulators), N).to(c.device)
    values = torch.empty((frames, 4), dtype=torch.float32, device=c.device)
    poc1 = torch.empty((frames, 4, 2), dtype=torch.float32, device=c.device)
    poc2 = torch.empty_like(poc1)
    idx = torch.empty((frames, 4, 2), dtype=torch.int32, device=c.device)
    _lib.check(L.as_tract_variables_fwd(_lib.ptr(c), frames, A, N, _lib.ptr(spec), 4, _lib.ptr(values), _lib.ptr(poc1),
                                        _lib.ptr(poc2), _lib.ptr(idx), _lib.stream_ptr()), "as_tract_variables_fwd")
    return values, poc1, poc2, idx


def calculate_vocal_tract_variables(inputs_dict):
    """
    Args:
        inputs_dict (dict): articulator name -> (N, 2) tensor of contour points.
    Return:
        TVs (dict): TV name -> {"value", "poc_1", "poc_2"} or None (reference :73-125).
    """
    arts = sorted(inputs_dict)
    frame = torch.stack([inputs_dict[a].T for a in arts]).unsqueeze(0)  # (1, A, 2, N)
    if not frame.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("artspeech_amd.tract_variables needs an MI355X device; there is no CPU path")
        frame = frame.cuda()
    values, poc1, poc2, _ = tract_variables_batched(frame, arts)
    values = values[0].tolist()
    tvs = {"LA": None, "LP": None, "TTCD": None, "TTCL": None, "TBCD": None, "TBCL": None, "VEL": None, "GLO": None}
    for j, name in enumerate(TV_NAMES):
        tvs[name] = {"value": values[j], "poc_1": poc1[0, j], "poc_2": poc2[0, j]}
    return tvs
