"""Articulatory scorer (reference ``phoneme_recognition``): only the DeepSpeech2 forward + top-1 decoding that config 5 of
BASELINE.json puts behind the phoneme-to-articulation models; the recogniser's own training loop is out of scope."""
from .deepspeech2 import DeepSpeech2, top1_phonemes  # noqa: F401
