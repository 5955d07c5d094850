"""lemo_amd -- MI355X-native (gfx950) implementation of LEMO's temporal SMPL-X fitting hot path.

Host side: Python on PyTorch-ROCm mirroring the reference's ``smplx`` / ``VPoser`` / ``Enc`` / ``AE``
call signatures.  Compute: hand-written HIP kernels behind a C-ABI shared library
(``lemo_amd/csrc`` -> ``liblemo_hip.so``, declared in ``include/lemo_hip.h``).
There is no CPU fallback: importing a compute entry point without the built library raises.
"""
import os as _os

# The HIP runtime maps a process's streams onto 4 hardware queues per device by default; streams that share a queue are
# serialised.  The side-by-side paths (sharding.ConcurrentClips, infill.finetune_and_infill_many, one engine + stream per clip)
# use 3-4 streams next to the caller's own: with 4 queues two clips end up in one queue (3 clips side by side 3680 instead of
# 4060-4330 fitting-iterations/s, 4 clips 3820 instead of 4340: profiles/r03_hw_queues.txt).  Read when the runtime initialises
# the device, i.e. effective if this package is imported before the first HIP call of the process; a value set by the user wins.
# No effect on one clip per GPU (A/B in the same file).
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'
