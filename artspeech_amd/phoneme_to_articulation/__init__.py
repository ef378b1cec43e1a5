"""phoneme_to_articulation package of the MI355X engine (reference: phoneme_to_articulation/__init__.py)."""
from enum import Enum

import torch.nn as nn


class RNNType(Enum):
    """Recurrent cell switch (reference phoneme_to_articulation/__init__.py:47-49).  The values are the torch classes only
    because the reference's are (they serve as PARAMETER CONTAINERS here: same keys, shapes and initialisation); the
    recurrences run on the HIP kernels of csrc/gru.hip and csrc/lstm.hip."""
    LSTM = nn.LSTM
    GRU = nn.GRU
