import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, 'lemo_amd', 'csrc')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def emu_lib():
    """liblemo_emu.so: the UNMODIFIED kernel sources compiled for the host against tests/hipemu
    (test infrastructure -- never loaded by lemo_amd itself)."""
    subprocess.run(['make', '-C', CSRC, '-j8', 'emu'], check=True, capture_output=True)
    from lemo_amd import _hip
    return _hip.HipLib(_hip.EMU_LIB_PATH, is_emu=True)


@pytest.fixture(scope='session')
def hip_lib_built():
    """liblemo_hip.so cross-compiled for gfx950 (hipcc works without a GPU)."""
    so = os.path.join(CSRC, 'liblemo_hip.so')
    if not os.path.exists(so):
        subprocess.run(['make', '-C', CSRC, '-j8', 'all'], check=True, capture_output=True)
    return so


def rel_err(a, b):
    import torch
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
