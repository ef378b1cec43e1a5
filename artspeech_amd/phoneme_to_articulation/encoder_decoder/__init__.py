"""Model-free phoneme-to-articulation (BiGRU encoder-decoder) on MI355X
(reference: phoneme_to_articulation/encoder_decoder/)."""
