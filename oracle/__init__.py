"""CPU oracle for the PPO / replay hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  The only importers
allowed are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` -- and there only as the *checker* / reported CPU baseline,
never as the thing being measured or shipped.  ``elegantrl_amd`` never imports
this package; its HIP path fails loudly when the extension is missing.

Parity status: PINNED.  Every function here was checked against outputs of the
reference itself (``/root/reference`` imported in the authoring container by
``oracle/make_golden.py``); those outputs are committed under ``tests/golden/``
and re-checked by ``tests/test_oracle_golden.py`` on every run.  The reference's
own unit tests pin no numbers (shape/dtype asserts only).
"""
