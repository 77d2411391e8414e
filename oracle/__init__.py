"""CPU oracle for the Next-DiT denoising path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import anything from
this package - and there only as the checker / the timed CPU baseline, never as part of the product path
(``lumina-t2x_amd/`` must not import it; a test enforces that).

Pinning status
  * model arithmetic (``nextdit_oracle``): PINNED against the reference's own ``NextDiT`` module
    (``/root/reference/lumina_next_t2i/models/model.py``) imported unmodified behind stub packages in the
    authoring container - ``oracle/make_golden.py`` generated ``tests/golden/*.npz`` from the reference and
    ``tests/test_oracle_golden.py`` checks the restatement against those vectors.
  * ODE stepping (``odeint_oracle``): the reference delegates to ``torchdiffeq.odeint`` (third party,
    unpinned in pyproject.toml:50 / requirements.txt:7, not installed, not vendored) - the fixed-grid
    solvers are restated from torchdiffeq's published algorithm, so this part is **parity unpinned**; it is
    anchored on the reference's call site (transport/integrators.py:97-116), the in-tree midpoint step in
    visual_anagrams/generate.py:212-219 and closed-form known answers.
"""
