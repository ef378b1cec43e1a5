"""``PrincipalComponentsArtSpeech`` (reference principal_components/models/rnn.py:36-109): Embedding -> 2-layer
bidirectional GRU **or LSTM** (``RNNType``) -> Linear + ReLU -> PrincipalComponentsPredictor (three LayerNorm -> Linear
stages) -> tanh, with forward and backward on the C ABI.

Same constructor, ``state_dict`` keys and seed-for-seed initial weights as the reference (the ``nn`` sub-modules are
parameter containers created in the reference's order; their ``forward`` is never called).  autograd only wires the
C-ABI Functions of ``rnn_ops`` / ``transformer.ops`` together; torch ops left are glue (embedding gather, tanh, the
stacking of per-direction parameters).
"""
from functools import reduce

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import _lib
from ....helpers import make_indices_dict
from ... import RNNType
from ...rnn_ops import birnn_stack, check_lengths
from ...transformer.ops import FoldLN, GroupedLinear, Normalize


def _ln_linear(x, ln, lin, relu):
    """Linear(LayerNorm(x)) [+ ReLU] on rows x [R, K]: affine-free normalisation, LayerNorm affine folded into the weights."""
    Wf, bf = FoldLN.apply(lin.weight[None], ln.weight[None], ln.bias[None], lin.bias[None])
    return GroupedLinear.apply(Normalize.apply(x)[None], Wf, bf, (0,), relu)[0]


class PrincipalComponentsPredictor(nn.Module):
    """Parameters of reference rnn.py:11-33; ``forward`` runs LN -> Linear -> ReLU -> LN -> Linear -> ReLU -> LN -> Linear."""

    def __init__(self, in_features, num_components, hidden_features=256):
        super().__init__()
        self.linear = nn.Sequential(
            nn.LayerNorm([in_features]), nn.Linear(in_features, hidden_features), nn.ReLU(),
            nn.LayerNorm([hidden_features]), nn.Linear(hidden_features, hidden_features // 2), nn.ReLU(),
            nn.LayerNorm(hidden_features // 2), nn.Linear(hidden_features // 2, num_components))

    def forward(self, inputs):
        _lib.require_gpu(inputs, "inputs")
        shape = inputs.shape
        x = inputs.reshape(-1, shape[-1])
        seq = self.linear
        x = _ln_linear(x, seq[0], seq[1], True)
        x = _ln_linear(x, seq[3], seq[4], True)
        x = _ln_linear(x, seq[6], seq[7], False)
        return x.reshape(*shape[:-1], x.shape[-1])


class PrincipalComponentsArtSpeech(nn.Module):
    def __init__(self, vocab_size, indices_dict, embed_dim=64, hidden_size=128, rnn_dropout=0., rnn=RNNType.GRU):
        super().__init__()
        if isinstance(list(indices_dict.values())[0], int):
            indices_dict = make_indices_dict(indices_dict)
        self.latent_size = 1 + max(set(reduce(lambda l1, l2: l1 + l2, indices_dict.values())))
        self.embedding = nn.Embedding(vocab_size, embed_dim)
        if isinstance(rnn, str):
            rnn = RNNType[rnn.upper()]
        self.rnn_kind = rnn.name.lower()
        self.rnn = rnn.value(embed_dim, hidden_size, num_layers=2, bidirectional=True, dropout=rnn_dropout, batch_first=True)
        self.rnn_dropout = float(rnn_dropout)
        self.linear = nn.Sequential(nn.Linear(2 * hidden_size, hidden_size), nn.ReLU())
        self.predictor = PrincipalComponentsPredictor(in_features=hidden_size, num_components=self.latent_size, hidden_features=256)

    @property
    def total_parameters(self):
        return sum(p.numel() for p in self.parameters())

    def forward(self, x, lengths):
        """x (bs, seq) token ids on the GPU; lengths sorted in decreasing order (CPU tensor or list, as the reference's
        ``pack_padded_sequence`` takes them) -> (bs, max(lengths), num_components)."""
        _lib.require_gpu(x, "x")
        _lib.require_gpu(self.embedding.weight, "model parameters")
        lengths_cpu, T = check_lengths(lengths, x.shape[0], x.shape[1])
        lengths_dev = lengths_cpu.to(x.device, non_blocking=True)
        B = x.shape[0]
        embed = F.embedding(x[:, :T].long(), self.embedding.weight)  # gather (glue)
        rnn_out = birnn_stack(self.rnn, embed, lengths_dev, self.rnn_kind, self.rnn_dropout, self.training)  # (B, T, 2H), zero padded
        lin = self.linear[0]
        linear_out = GroupedLinear.apply(rnn_out.reshape(1, B * T, -1), lin.weight[None], lin.bias[None], (0,), True)[0]
        components = torch.tanh(self.predictor(linear_out))
        return components.reshape(B, T, self.latent_size)
