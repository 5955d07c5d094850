"""The C-ABI library loads and exports every symbol include/lemo_hip.h declares (no compute)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, 'include', 'lemo_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(lemo_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_expected_entry_points():
    names = _declared()
    for need in ('lemo_conv3x3_mfma', 'lemo_vposer_decode_fwd', 'lemo_smplx_pose_fwd', 'lemo_lbs_verts_fwd',
                 'lemo_lbs_verts_bwd', 'lemo_fit_create', 'lemo_fit_step', 'lemo_fit_prepare'):
        assert need in names


def test_gfx950_library_exports_every_declared_symbol(hip_lib_built):
    dll = ctypes.CDLL(hip_lib_built)
    missing = [n for n in _declared() if not hasattr(dll, n)]
    assert not missing, missing
    assert dll.lemo_abi_version() == 1


def test_python_binding_covers_header():
    from lemo_amd import _hip
    assert sorted(_hip.EXPORTED_SYMBOLS) == _declared()


def test_library_contains_gfx950_code_object(hip_lib_built):
    blob = open(hip_lib_built, 'rb').read()
    assert b'gfx950' in blob and b'conv3x3_mfma_kernel' in blob


def test_product_refuses_cpu_tensors(hip_lib_built):
    import pytest
    import torch
    from lemo_amd import _hip
    from lemo_amd.rotation import convert_to_3D_all
    with pytest.raises(_hip.LemoHipError):
        convert_to_3D_all(torch.randn(4, 6))            # no CPU fallback
