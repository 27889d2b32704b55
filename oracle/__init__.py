"""CPU oracle for the MultimodalController hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path (the package
``multimodal-controller-for-generative-models_amd`` / import alias ``mcgen_amd``)
never imports this package and raises if its HIP library is missing.

Parity status: PINNED.  The reference publishes no golden vectors (SURVEY.md
section 4), so the oracle is pinned by vectors produced by importing the
reference's ``models``/``modules`` packages on CPU in the build container
(``tools/gen_golden.py``; committed fixtures under ``tests/golden/``) and
checked by ``tests/test_oracle_golden.py``.
"""
