"""Per-articulator autoencoders of the principal-components method (reference principal_components/models/autoencoder.py:
83-260): the MLP ``Encoder`` / ``Decoder``, their multi-articulator containers and ``MultiArticulatorAutoencoder``, with the
reference's constructors, ``state_dict`` keys and seed-for-seed initial weights.  Every Linear (+ReLU) of the forward and the
backward is a C-ABI fp32-MFMA GEMM (``GroupedLinear``); slicing / stacking / max over articulators are torch glue.  The PCA
encoders / decoders of the reference (:10-80, closed-form projections fitted by scikit-learn) are not part of this path.
"""
import torch
import torch.nn as nn

from .... import _lib
from ....helpers import make_indices_dict
from ...transformer.ops import GroupedLinear


def _mlp(seq, x):
    """nn.Sequential(Linear, ReLU, Linear, ReLU, Linear) of parameter containers on rows x [R, in] -> [R, out]."""
    _lib.require_gpu(x, "x")
    h = x.reshape(-1, x.shape[-1]).float()
    for idx, relu in ((0, True), (2, True), (4, False)):
        lin = seq[idx]
        h = GroupedLinear.apply(h[None], lin.weight[None], lin.bias[None], (0,), relu)[0]
    return h.reshape(*x.shape[:-1], h.shape[-1])


class Encoder(nn.Module):
    def __init__(self, in_features, num_components, hidden_features):
        super().__init__()
        self.encoder = nn.Sequential(nn.Linear(in_features, hidden_features), nn.ReLU(),
                                     nn.Linear(hidden_features, hidden_features // 2), nn.ReLU(),
                                     nn.Linear(hidden_features // 2, num_components))

    def forward(self, x):
        return _mlp(self.encoder, x)


class Decoder(nn.Module):
    def __init__(self, num_components, out_features, hidden_features):
        super().__init__()
        self.decoder = nn.Sequential(nn.Linear(num_components, hidden_features // 2), nn.ReLU(),
                                     nn.Linear(hidden_features // 2, hidden_features), nn.ReLU(),
                                     nn.Linear(hidden_features, out_features))

    def forward(self, x):
        return _mlp(self.decoder, x)


def _resolve(indices_dict):
    if isinstance(list(indices_dict.values())[0], int):
        indices_dict = make_indices_dict(indices_dict)
    latent_size = max(i for indices in indices_dict.values() for i in indices) + 1
    return indices_dict, latent_size, sorted(indices_dict.keys())


class MultiEncoder(nn.Module):
    """One Encoder per articulator; every latent index takes the maximum over the articulators that own it (:155-173)."""

    def __init__(self, indices_dict, in_features, hidden_features, encoder_cls=Encoder):
        super().__init__()
        if encoder_cls is not Encoder and encoder_cls != "AE":
            raise NotImplementedError("only the MLP encoder (EncoderType.AE) is built on the C ABI")
        self.indices_dict, self.latent_size, self.sorted_articulators = _resolve(indices_dict)
        self.encoders = nn.ModuleDict({articulator: Encoder(in_features=in_features, num_components=len(indices),
                                                            hidden_features=hidden_features)
                                       for articulator, indices in self.indices_dict.items()})

    def forward(self, x):
        """x (bs, n_articulators, in_features), channels in sorted-articulator order -> (bs, latent_size)."""
        bs = x.shape[0]
        spaces = []
        for i, articulator in enumerate(self.sorted_articulators):
            space = torch.full((bs, self.latent_size), -torch.inf, dtype=torch.float32, device=x.device)
            space[..., self.indices_dict[articulator]] = self.encoders[articulator](x[..., i, :])
            spaces.append(space)
        return torch.stack(spaces, dim=1).max(dim=1).values


class MultiDecoder(nn.Module):
    """One Decoder per articulator on its own slice of the latent vector, outputs stacked on dim -2 (:199-213)."""

    def __init__(self, indices_dict, in_features, hidden_features, decoder_cls=Decoder):
        super().__init__()
        if decoder_cls is not Decoder and decoder_cls != "AE":
            raise NotImplementedError("only the MLP decoder (DecoderType.AE) is built on the C ABI")
        self.indices_dict, self.latent_size, self.sorted_articulators = _resolve(indices_dict)
        self.decoders = nn.ModuleDict({articulator: Decoder(num_components=len(indices), out_features=in_features,
                                                            hidden_features=hidden_features)
                                       for articulator, indices in self.indices_dict.items()})

    def forward(self, x):
        """x (..., latent_size) -> (..., n_articulators, in_features)."""
        outs = [self.decoders[articulator](x[..., self.indices_dict[articulator]].contiguous()).unsqueeze(-2)
                for articulator in self.sorted_articulators]
        return torch.cat(outs, dim=-2)


class MultiArticulatorAutoencoder(nn.Module):
    def __init__(self, in_features, indices_dict, hidden_features=64):
        super().__init__()
        self.indices_dict, self.latent_size, self.sorted_articulators = _resolve(indices_dict)
        self.encoders = MultiEncoder(indices_dict=indices_dict, in_features=in_features, hidden_features=hidden_features)
        self.decoders = MultiDecoder(indices_dict=indices_dict, in_features=in_features, hidden_features=hidden_features)

    @property
    def total_parameters(self):
        return sum(p.numel() for p in self.parameters())

    def forward(self, x):
        """x (bs, n_articulators, in_features) -> (outputs (bs, n_articulators, in_features), latent (bs, latent_size))."""
        latent_space = torch.tanh(self.encoders(x))
        return self.decoders(latent_space), latent_space
