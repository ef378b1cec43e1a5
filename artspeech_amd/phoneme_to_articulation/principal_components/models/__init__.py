from .rnn import PrincipalComponentsArtSpeech, PrincipalComponentsPredictor  # noqa: F401
