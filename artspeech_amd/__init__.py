"""artspeech_amd -- MI355X-native engine for the phoneme_to_articulation hot path of vribeiro1/artspeech.

The package mirrors the reference's module tree for that path (same class names, constructor
signatures, state_dict keys, argument meaning and error behaviour) on top of hand-written gfx950 HIP
kernels behind a C ABI (include/artspeech_hip.h).  PyTorch is used for device memory, streams,
autograd plumbing and torch.distributed (RCCL) only.
"""
__version__ = "0.1.0"
