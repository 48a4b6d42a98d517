"""racformer_amd — MI355X-native (gfx950) implementation of RaCFormer's query-decoder hot path.

Only what the path needs lives here: ``csrc/`` (hand-written HIP kernels behind a C-ABI,
``include/racformer_hip.h``) and host-side mirrors of the reference's operator/plugin surface
(``msmv_sampling``, ``MultiScaleDeformableAttnFunction_fp32``, ``sampling_4d``,
``RaCFormerTransformer``, ``RaCFormer_head``).  The compute path has no CPU fallback: ops raise
if the HIP library is missing or a tensor is not on the GPU.
"""
__version__ = "0.1.0"
