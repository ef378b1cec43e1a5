"""Critical-distance loss of the principal-components method (reference principal_components/losses.py:23-99): for every
tract variable the minimum pairwise distance between two predicted articulators per frame, averaged over the frames where
the constriction is critical.  The reference materialises ``cdist`` (bs, T, N, N) per variable and takes ``min``; here one
launch of the tract-variable tile kernel (``as_tract_variables_fwd``) yields the minima and their arg-min points for all
variables and frames, and the backward sends the gradient to those two points (what ``torch.min`` / ``cdist`` do).
"""
import torch
import torch.nn as nn

from ... import _lib

LOWER_LIP, PHARYNX, SOFT_PALATE, TONGUE = "lower-lip", "pharynx", "soft-palate", "tongue"
UPPER_INCISOR, UPPER_LIP = "upper-incisor", "upper-lip"


class _MinPairDistance(torch.autograd.Function):
    """pairs [frames, 2 * n_tv, 2, N] (channel 2k = first point set of variable k, 2k + 1 = second) -> [frames, n_tv]."""

    @staticmethod
    def forward(ctx, pairs):
        pairs = pairs.contiguous().float()
        frames, channels, _, N = pairs.shape
        n_tv = channels // 2
        dev = pairs.device
        spec = torch.tensor([[[2 * k, 0, N], [2 * k + 1, 0, N], [-1, 0, 0]] for k in range(n_tv)], dtype=torch.int32, device=dev)
        values = torch.empty((frames, n_tv), dtype=torch.float32, device=dev)
        poc1, poc2 = torch.empty((frames, n_tv, 2), dtype=torch.float32, device=dev), torch.empty((frames, n_tv, 2), dtype=torch.float32, device=dev)
        idx = torch.empty((frames, n_tv, 2), dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().as_tract_variables_fwd(_lib.ptr(pairs), frames, channels, N, _lib.ptr(spec), n_tv, _lib.ptr(values),
                                                     _lib.ptr(poc1), _lib.ptr(poc2), _lib.ptr(idx), _lib.stream_ptr()),
                   "as_tract_variables_fwd")
        ctx.save_for_backward(values, poc1, poc2, idx)
        ctx.shape = pairs.shape
        return values

    @staticmethod
    def backward(ctx, dvalues):
        values, poc1, poc2, idx = ctx.saved_tensors
        frames, channels, _, N = ctx.shape
        n_tv = channels // 2
        unit = (poc1 - poc2) / values.unsqueeze(-1)                      # d|p - q| / dp  (NaN at zero distance, like the reference)
        g = dvalues.unsqueeze(-1) * unit                                 # [frames, n_tv, 2]
        grad = torch.zeros((frames, n_tv, 2, 2, N), dtype=torch.float32, device=values.device)  # [f, k, set, xy, point]
        index = idx.long().view(frames, n_tv, 2, 1, 1).expand(frames, n_tv, 2, 2, 1)
        src = torch.stack([g, -g], dim=2).unsqueeze(-1)                  # [f, k, set, xy, 1]
        grad.scatter_(4, index, src)
        return grad.view(frames, channels, 2, N)


class CriticalLoss(nn.Module):
    TV_TO_ARTICULATOR_MAP = {"LA": [LOWER_LIP, UPPER_LIP], "TTCD": [TONGUE, UPPER_INCISOR], "TBCD": [TONGUE, UPPER_INCISOR],
                             "VEL": [SOFT_PALATE, PHARYNX]}

    def __init__(self, TVs, articulators, denormalize_fn=None):
        super().__init__()
        self.TVs = sorted(TVs)
        self.inject_reference = UPPER_INCISOR not in articulators
        if UPPER_INCISOR not in articulators:
            articulators = sorted(articulators + [UPPER_INCISOR])
        self.articulators_indices = {articulator: i for i, articulator in enumerate(articulators)}
        self.denorm_fn = denormalize_fn

    def forward(self, output_shapes, target_shapes, reference_arrays, critical_mask):
        """output_shapes / target_shapes (bs, T, n_articulators, 2, N), reference_arrays (bs, T, 1, 2, N) (the upper incisor,
        injected when it is not predicted), critical_mask (bs, n_TVs, T) -> scalar."""
        if len(self.TVs) == 0:
            return torch.tensor(0, device=target_shapes.device, dtype=torch.float)
        _lib.require_gpu(output_shapes, "output_shapes")
        if self.inject_reference:
            ref_index = self.articulators_indices[UPPER_INCISOR]
            output_shapes = torch.cat([output_shapes[:, :, :ref_index], reference_arrays.to(output_shapes.device),
                                       output_shapes[:, :, ref_index:]], dim=2)
        bs, seq_len, _, _, num_samples = target_shapes.shape
        sets = []
        for TV in self.TVs:
            for articulator in self.TV_TO_ARTICULATOR_MAP[TV]:
                array = output_shapes[..., self.articulators_indices[articulator], :, :]
                if self.denorm_fn and articulator != UPPER_INCISOR:
                    array = self.denorm_fn[articulator](array)
                sets.append(array)
        pairs = torch.stack(sets, dim=2).reshape(bs * seq_len, 2 * len(self.TVs), 2, num_samples)
        critical = _MinPairDistance.apply(pairs).view(bs, seq_len, len(self.TVs)).permute(0, 2, 1)  # (bs, n_TVs, T)
        return critical[critical_mask.to(critical.device) == 1].mean()
