"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the GaUDI guided-sampling hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it, and
there only as the checker / the reported CPU baseline.  The product path
(``gaudi_amd`` -> ``libgaudi_hip.so``) never imports it and fails loudly without its HIP
extension.

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference itself,
imported in the build container by ``tools/make_golden.py`` and committed as
``tests/golden/*.npz`` (torch 2.10.0 CPU, fp32).  ``tests/test_oracle_golden.py`` checks
every oracle function against those vectors.
"""
