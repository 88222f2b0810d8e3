"""CPU oracle for the spectral-convolution hot path (TEST INFRASTRUCTURE ONLY).

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / reported baseline.  The product
path (``resolution-pde_amd/``) never imports this package and fails loudly
when the HIP library is missing.

Parity status: PINNED.  ``oracle/reference_path.py`` is checked against the
reference modules imported in the build container (``tests/golden/make_golden.py``
is the committed generating script) and against the committed fixtures in
``tests/golden/*.npz`` produced by that import.
"""
