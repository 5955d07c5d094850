"""marker-image encode / decode kernels on the host emulator against the oracle and the golden vectors that
tests/golden/make_golden.py produced with the reference's own utils/utils.py."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from oracle import markers_oracle as MO


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(GOLDEN, 'markers_decode.npz'))


def test_oracle_matches_reference_golden(gold):
    img, piv = MO.get_local_markers_4chan(gold['body'].astype(np.float64), gold['contact'].astype(np.float64))
    assert rel_err(torch.from_numpy(img), torch.from_numpy(gold['image'].astype(np.float64))) < 1e-6
    assert abs(float(piv[0]) - float(gold['rot_0_pivot'][0])) < 1e-7
    glob = MO.reconstruct_global_body(gold['decode_in'].astype(np.float64), gold['rot_0_pivot'])
    assert rel_err(torch.from_numpy(glob), torch.from_numpy(gold['global_body'].astype(np.float64))) < 1e-6


def test_kernels_vs_golden_and_round_trip(emu_lib, gold):
    from lemo_amd.markers import get_local_markers_4chan, reconstruct_global_body
    body, contact = gold['body'], gold['contact']
    before = body.copy()
    img, piv = get_local_markers_4chan(body, contact, _lib=emu_lib)
    assert np.array_equal(body, before)                                 # inputs are not modified (the reference's are)
    assert img.shape == (4, 119, 3 * 68 + 4) and piv.shape == (1,)
    assert rel_err(torch.from_numpy(img), torch.from_numpy(gold['image'].astype(np.float64))) < 1e-5
    assert abs(float(piv[0]) - float(gold['rot_0_pivot'][0])) < 1e-6
    glob = reconstruct_global_body(gold['decode_in'], gold['rot_0_pivot'], _lib=emu_lib)
    assert glob.shape == (119, 68, 3)
    assert rel_err(torch.from_numpy(glob), torch.from_numpy(gold['global_body'].astype(np.float64))) < 1e-5
    # decode(encode(x)) gives the clip back up to the floor shift and the start pose (x, y of the first pelvis)
    shifted = gold['body'][:-1].astype(np.float64).copy()
    shifted[:, :, 2] -= gold['body'][:, :, 2].min()
    shifted[:, :, :2] -= gold['body'][0, 0, :2]
    assert np.abs(glob - shifted).max() < 2e-4


def test_ragged_and_bad_shapes(emu_lib):
    from lemo_amd.markers import get_local_markers_4chan, reconstruct_global_body
    from lemo_amd._hip import LemoHipError
    rng = np.random.default_rng(0)
    # shortest clip the encode takes (T = 2 -> one output frame) against the oracle
    body = rng.normal(0, 0.3, (2, 68, 3)); body[:, 57] += [0, -0.4, 0]; body[:, 27] += [0, 0.4, 0]
    contact = np.ones((2, 4))
    img, piv = get_local_markers_4chan(body, contact, _lib=emu_lib)
    ref, rp = MO.get_local_markers_4chan(body, contact)
    assert rel_err(torch.from_numpy(img), torch.from_numpy(ref)) < 1e-5 and abs(float(piv[0] - rp[0])) < 1e-6
    with pytest.raises(LemoHipError):
        get_local_markers_4chan(rng.normal(size=(300, 68, 3)), np.ones((300, 4)), _lib=emu_lib)      # T > 256
    with pytest.raises(LemoHipError):
        get_local_markers_4chan(rng.normal(size=(10, 30, 3)), np.ones((10, 4)), _lib=emu_lib)        # direction markers missing
    one = reconstruct_global_body(rng.normal(size=(1, 5, 3)), np.array([0.3]), _lib=emu_lib)        # T = 1: rotation only
    assert one.shape == (1, 3, 3)
