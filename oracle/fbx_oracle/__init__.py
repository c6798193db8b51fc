"""fbx_oracle -- CPU (NumPy/SciPy) restatement of forest-benchmarking's tomography hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and
there only as the checker / the timed CPU baseline.  The product (``fbx`` package +
``libfbx.so``) never imports, links or executes anything in this directory.

Parity status: PINNED.  Every function here is checked (a) against the reference itself,
imported in the build container through ``tests/golden/_ref_harness.py`` (see
``tests/test_oracle_vs_reference.py``; skipped where /root/reference is absent) and
(b) against the committed golden vectors under ``tests/golden/`` which were produced by
running the reference (``tests/golden/make_goldens.py``) and by transcribing the
known-answer values of the reference's own test-suite (``tests/golden/known_answers.py``).

Third-party arithmetic not present under /root/reference: pyquil==4.5.0
(``lifted_pauli`` / ``lifted_state_operator`` / ``simulation.matrices.STATES``), restated
in :mod:`fbx_oracle.design` from its published algorithm.

Every function cites the reference file:line it follows (paths relative to
``forest/benchmarking/`` of rigetti/forest-benchmarking v0.9.0).
"""
from . import design, superops, measures, estimators, acquisition  # noqa: F401
