"""CPU oracle for the Diff-Foley Stage-2 sampling path.  TEST INFRASTRUCTURE ONLY.

This package is a plain torch-fp32 *restatement* of the reference algorithm
(file:line citations in every function).  It exists to check the HIP path:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product package (``diff_foley_amd/``) never imports it
and has no CPU fallback.

Pinning: the reference ships no tests, so the oracle is
pinned against outputs of the reference itself, imported on CPU in the build
container by ``tests/golden/make_golden.py`` (which uses ``oracle/ref_import.py``
for the import stubs).  The resulting vectors live in ``tests/golden/*.npz`` and
``tests/test_oracle_golden.py`` checks the oracle against every one of them.
"""
