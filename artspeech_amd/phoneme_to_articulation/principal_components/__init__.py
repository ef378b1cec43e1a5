"""Principal-components method (reference phoneme_to_articulation/principal_components): the recurrent phoneme -> latent
components model (models/rnn.py) on the C ABI.  The autoencoders, their losses and the dataset of that method are not
part of the hot path (SURVEY section 8f rank 3 names the cell + trunk)."""
