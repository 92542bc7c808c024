"""CPU oracle for the range-image hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (plain torch fp32 functional ops + numpy for the
integer/histogram work) of the reference algorithm on the hot path named in
BASELINE.json.  It is pinned against the reference itself: ``tools/gen_golden.py``
imports the reference modules from ``/root/reference/src`` in the build container,
asserts that every function here agrees with them on seeded inputs, and writes the
golden vectors under ``tests/golden/``.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product (``semanticlidarunc_amd``) never
does: it fails loudly when the HIP library is missing.
"""
