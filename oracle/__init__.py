"""CPU oracle for the collaborative-sampling refinement hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the reported CPU baseline.
The product package (``collaborative-gan-sampling_amd``) never imports it and
fails loudly when its HIP library is missing.

What it is: a restatement, in this repo's own words, of the reference's
algorithm for the path ``sampling/collaborator.py`` + ``sampling/refiner_cpu.py``
(+ ``policy.py``, ``rejector.py``, ``idpsampler.py``, the ``nsgan/ops.py``
operators and the ``nsgan/GAN.py`` / ``synthetic/GAN.py`` networks the refiner
differentiates), in plain torch-CPU fp32 / numpy f64.

Pinning status (see DESIGN.md "Oracle"):
  * loop / host algorithms (Refiner x2, PolicyAdaptive, Rejector,
    IndependenceSampler, ToyDataset): PINNED against outputs of the reference's
    own classes run in the build container behind a test-only TensorFlow shim
    (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
  * operator arithmetic (conv2d / conv2d_transpose TF-'SAME', batch_norm): the
    reference delegates it to TensorFlow 1.13 / cuDNN (not under
    /root/reference, not installable here) -> PARITY UNPINNED at that boundary;
    restated from the documented TF semantics and
    cross-checked by adjoint identities and a direct-loop numpy evaluation.
"""
