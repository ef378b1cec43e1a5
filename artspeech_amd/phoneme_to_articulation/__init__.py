"""phoneme_to_articulation package of the MI355X engine (reference: phoneme_to_articulation/__init__.py)."""
