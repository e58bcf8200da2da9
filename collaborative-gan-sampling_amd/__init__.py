"""collaborative-gan-sampling_amd: MI355X-native collaborative-sampling refinement engine.

The hot path of vita-epfl/collaborative-gan-sampling (sampling/collaborator.py +
sampling/refiner_cpu.py) behind the reference's own class / operator surface:

  * ``lib``       ctypes binding of libcgs_hip.so (include/cgs_hip.h), hand-written gfx950 kernels
  * ``ops``       the nsgan/ops.py operator API (bn, conv2d, deconv2d, lrelu, linear, ...)
  * ``nets``      the G head / G tail / D the refiner differentiates (nsgan/GAN.py:59-101 + DCGAN-32/64)
  * ``engine``    the fused K-step refinement loop on one GPU (workspace-resident state, hipGraph replay)
  * ``sampling``  Refiner (x2), PolicyAdaptive, Rejector, IndependenceSampler -- the reference classes
  * ``dist``      z-batch sharding over ranks and the RCCL gather of the refined pool

Import as ``cgs_amd`` (the directory name is not an identifier).  There is no CPU fallback for the
device path: every device op raises if libcgs_hip.so is missing.
"""
__version__ = "0.1.0"
