import torch.nn as nn

from .autoencoder import Decoder, Encoder, MultiArticulatorAutoencoder, MultiDecoder, MultiEncoder  # noqa: F401
from .rnn import PrincipalComponentsArtSpeech, PrincipalComponentsPredictor  # noqa: F401


class PrincipalComponentsArtSpeechWrapper(nn.Module):
    """phonemes -> latent components -> articulator shapes (reference principal_components/models/__init__.py:20-46):
    ``rnn`` = PrincipalComponentsArtSpeech, ``decoder`` = MultiDecoder, ``denorm`` = {articulator: callable}."""

    def __init__(self, rnn, decoder, denorm):
        super().__init__()
        self.rnn = rnn
        self.decoder = decoder
        self.denorm = denorm

    def forward(self, x, lengths):
        """x (bs, seq_len) token ids, lengths sorted descending -> (bs, seq_len, n_articulators, 2, n_samples)."""
        components = self.rnn(x, lengths)
        outputs = self.decoder(components)
        bs, seq_len, n_articulators, features = outputs.shape
        outputs = outputs.reshape(bs, seq_len, n_articulators, 2, features // 2)
        for i, articulator in enumerate(self.decoder.sorted_articulators):
            outputs[:, :, i, :, :] = self.denorm[articulator](outputs[:, :, i, :, :])
        return outputs
