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
    assert dll.lemo_abi_version() == 5


def test_library_exports_nothing_but_the_header(hip_lib_built):
    """built with -fvisibility=hidden + csrc/exports.map: the dynamic symbol table IS the header (117 C++ launchers and
    the kernel handles used to be exported next to the 84 C names)"""
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', hip_lib_built], check=True, capture_output=True, text=True).stdout
    exported = sorted(line.split()[-1] for line in out.strip().split('\n') if line.strip())
    assert exported == _declared(), sorted(set(exported) ^ set(_declared()))


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


def test_descriptor_layouts_match_the_header():
    """the ctypes mirrors of lemo_fit_desc / lemo_prox_desc must have the C structs' size and field offsets: compiled from
    include/lemo_hip.h with the host compiler (a mismatch is a silent wrong-pointer bug otherwise)"""
    import ctypes as C
    import subprocess
    import tempfile
    from lemo_amd import _hip
    fields = {'lemo_fit_desc': (_hip.FitDesc, ['enc_w3', 'enc_w3_inv', 'target', 'transl', 'act', 'per_frame', 'verts_side', 'transl_side']),
              'lemo_prox_desc': (_hip.ProxDesc, ['enc_w3_inv', 'sdf', 'pose_embedding', 'losses']),
              'lemo_ae_desc': (_hip.AeDesc, ['lr', 'ws', 'ws_floats'])}
    src = '#include <cstdio>\n#include <cstddef>\n#include "lemo_hip.h"\nint main(){\n'
    for name, (_, fl) in fields.items():
        src += f'printf("%zu", sizeof({name}));' + ''.join(f'printf(" %zu", offsetof({name}, {f}));' for f in fl) + 'printf("\\n");\n'
    src += 'return 0;}\n'
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, 'o.cpp'), 'w').write(src)
        subprocess.run(['g++', '-I', os.path.join(root, 'include'), os.path.join(td, 'o.cpp'), '-o', os.path.join(td, 'o')], check=True)
        lines = subprocess.run([os.path.join(td, 'o')], check=True, capture_output=True, text=True).stdout.strip().split('\n')
    for line, (name, (cls, fl)) in zip(lines, fields.items()):
        want = [int(v) for v in line.split()]
        got = [C.sizeof(cls)] + [getattr(cls, f).offset for f in fl]
        assert got == want, (name, got, want)
