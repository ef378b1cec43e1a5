"""CPU oracle (test infrastructure only).  See artspeech_oracle.py."""
