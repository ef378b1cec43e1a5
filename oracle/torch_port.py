"""CPU baseline port -- TEST/BENCH INFRASTRUCTURE ONLY (used by bench.py's ``cpu_baseline`` leg and tests).

The reference runs this path on stock PyTorch CPU kernels (oneDNN/ATen GRU, GEMM, LayerNorm).  The
reference's own files cannot travel to the GPU box, so this module restates the same network
functionally on the same stock PyTorch CPU operators, driven by a state_dict with the reference's key
names (encoder_decoder/models.py:99-145 for the graph; phoneme_to_articulation/metrics.py:17-24 and
train_phoneme_to_articulation.py:86-90 for the loss).  It is pinned by tests/test_oracle_golden.py
against the golden fixtures produced by the reference itself.  kind = "port" in bench.py's JSON.
"""
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence


class CpuPort:
    def __init__(self, state_dict, n_art, hidden):
        self.p = {k: v.detach().clone().float().requires_grad_(True) for k, v in state_dict.items()}
        self.n_art, self.hidden = n_art, hidden
        embed = self.p["embedding.weight"].shape[1]
        self.rnn = torch.nn.GRU(embed, hidden, num_layers=2, bidirectional=True, batch_first=True)
        # share storage: the GRU module's parameters ARE our leaf tensors
        for name in list(self.rnn._parameters):
            self.rnn._parameters[name] = torch.nn.Parameter(self.p["rnn." + name].detach().clone())
        self.params = [v for k, v in self.p.items() if not k.startswith("rnn.")] + list(self.rnn.parameters())

    def forward(self, x, lengths):
        p = self.p
        emb = F.embedding(x, p["embedding.weight"])
        packed, _ = self.rnn(pack_padded_sequence(emb, lengths, batch_first=True))
        rnn_out, _ = pad_packed_sequence(packed, batch_first=True)
        lin = F.relu(F.linear(rnn_out, p["linear.0.weight"], p["linear.0.bias"]))
        heads = []
        for a in range(self.n_art):
            q = f"predictors.{a}."
            h = F.layer_norm(lin, lin.shape[-1:], p[q + "linear.0.weight"], p[q + "linear.0.bias"])
            h = F.relu(F.linear(h, p[q + "linear.1.weight"], p[q + "linear.1.bias"]))
            h = F.layer_norm(h, h.shape[-1:], p[q + "linear.3.weight"], p[q + "linear.3.bias"])
            h = F.relu(F.linear(h, p[q + "linear.4.weight"], p[q + "linear.4.bias"]))
            h = F.layer_norm(h, h.shape[-1:], p[q + "linear.6.weight"], p[q + "linear.6.bias"])
            heads.append(torch.stack([F.linear(h, p[q + "x_coords.weight"], p[q + "x_coords.bias"]),
                                      F.linear(h, p[q + "y_coords.weight"], p[q + "y_coords.bias"])], dim=2))
        return torch.sigmoid(torch.stack(heads, dim=2))

    def loss(self, out, targets, lengths):
        dx = out[..., 0, :] - targets[..., 0, :]
        dy = out[..., 1, :] - targets[..., 1, :]
        dist = torch.sqrt(dx ** 2 + dy ** 2)
        mask = torch.arange(out.shape[1])[None, :] < torch.as_tensor(lengths)[:, None]
        return dist[mask].mean()

    def step(self, x, lengths, targets):
        """forward + loss + backward (gradients of every parameter); returns the loss value."""
        for q in self.params:
            q.grad = None
        out = self.forward(x, lengths)
        loss = self.loss(out, targets[:, :out.shape[1]], lengths)
        loss.backward()
        return float(loss), out.detach()
