"""Host-side mirror of the reference's Python interface for the hot path (same function names,
argument meaning and error behaviour); the top-level `encoder/` and `decoder/` packages re-export it
so that rhccq.ipynb and encoder/compression/test.py of the reference import unchanged."""
