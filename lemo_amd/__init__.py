"""lemo_amd -- MI355X-native (gfx950) implementation of LEMO's temporal SMPL-X fitting hot path.

Host side: Python on PyTorch-ROCm mirroring the reference's ``smplx`` / ``VPoser`` / ``Enc`` / ``AE``
call signatures.  Compute: hand-written HIP kernels behind a C-ABI shared library
(``lemo_amd/csrc`` -> ``liblemo_hip.so``, declared in ``include/lemo_hip.h``).
There is no CPU fallback: importing a compute entry point without the built library raises.
"""
__version__ = '0.1.0'
