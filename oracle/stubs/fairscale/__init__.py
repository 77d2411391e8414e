"""Stub of the third-party ``fairscale`` package (not installed here).  TEST INFRASTRUCTURE ONLY: lets the
unmodified reference modules import; at model-parallel size 1 the parallel layers are plain nn.Linear /
nn.Embedding (the reference says so itself, lumina_next_t2i/demo.py:84-85)."""
