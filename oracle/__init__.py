"""CPU oracle for the MultimodalController hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path (the package
``multimodal-controller-for-generative-models_amd`` / import alias ``mcgen_amd``)
never imports this package and raises if its HIP library is missing.

Modules: ``mcgan_oracle`` (headline model + train step), ``mcvae_oracle``,
``mcglow_oracle``, ``mcpixelcnn_oracle``, ``vqvae_oracle`` (8(f) rank 1: the frozen
VQ-VAE encode / decode_code in front of MCPixelCNN).

Parity status: PINNED.  The reference publishes no golden vectors (SURVEY.md
section 4), so the oracle is pinned by vectors produced by importing the
reference's ``models``/``modules`` packages on CPU in the build container
(``tools/gen_golden.py``; committed fixtures under ``tests/golden/``) and
checked by ``tests/test_oracle_golden.py`` and ``tests/test_oracle_other_models.py``.
"""
