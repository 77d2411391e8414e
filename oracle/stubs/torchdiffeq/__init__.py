"""Stub of ``torchdiffeq`` (not installed).  TEST INFRASTRUCTURE ONLY: forwards to the restated fixed-grid
solvers in oracle/odeint_oracle.py (parity unpinned - see that file's header)."""
from oracle.odeint_oracle import odeint  # noqa: F401
