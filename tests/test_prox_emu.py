"""CPU suite: the PROX twin of the hot path (temp_prox/fitting_temp_slide.py:239-311 + the S2/S3-active
parts of SMPLifyLoss) through the HIP kernels on the host emulator vs the PROX oracle: all 14 entries of
loss_dict, gradients with and without the first-window erase, and 2 Adam steps."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from lemo_amd._hip import ptr


def test_sdf_sample_matches_grid_sample(emu_lib):
    from lemo_amd.scene import sdf_sample
    g = torch.Generator().manual_seed(0)
    sdf = torch.randn(16, 18, 20, generator=g)
    pts = (torch.rand(50, 7, 3, generator=g) * 2.6 - 1.3).requires_grad_(True)       # some points outside -> border
    gmin, gmax = (-1., -1.1, -0.9), (1., 1.2, 1.1)
    v = sdf_sample(pts, sdf, gmin, gmax, _lib=emu_lib)
    norm = (pts - torch.tensor(gmin)) / (torch.tensor(gmax) - torch.tensor(gmin)) * 2 - 1
    ref = F.grid_sample(sdf[None, None], norm[..., [2, 1, 0]].view(1, -1, 1, 1, 3), padding_mode='border',
                        align_corners=False).view(50, 7)
    assert float((v - ref).abs().max()) < 1e-5
    w = torch.randn(50, 7, generator=g)
    (v * w).sum().backward()
    gp = pts.grad.clone(); pts.grad = None
    (ref * w).sum().backward()
    assert rel_err(gp, pts.grad) < 1e-5


def test_enc_module_state_dict_and_values(emu_lib):
    from lemo_amd.assets import load_assets
    from lemo_amd.priors import Enc
    from oracle import lemo_oracle as O
    A = load_assets()
    enc = Enc(downsample=False, z_channel=64, _lib=emu_lib)
    assert set(enc.state_dict().keys()) == set(A['enc_w_torch'].keys())
    enc.load_state_dict(A['enc_w_torch'])
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(1, 1, 20, 29, generator=g) * 0.05).requires_grad_(True)
    z, s0, s1, s2, s3, s4 = enc(x)
    zr = O.enc_forward(A['enc_w_torch'], x)
    assert tuple(s0) == (1, 1, 20, 29) and tuple(s1) == (1, 32, 20, 29) and tuple(s4) == (1, 64, 20, 29)
    assert rel_err(z.detach(), zr.detach()) < 1e-5
    wz = torch.randn(z.shape, generator=g)
    (z * wz).sum().backward()
    g1 = x.grad.clone(); x.grad = None
    (zr * wz).sum().backward()
    assert rel_err(g1, x.grad) < 1e-4
    x.grad = None
    l = enc.smooth_loss(x)
    lr = torch.mean((zr.detach()[..., 1:] - zr.detach()[..., :-1]) ** 2)
    assert abs(float(l) - float(lr)) <= 1e-5 * float(lr)
    with pytest.raises(NotImplementedError):
        Enc(downsample=True)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('stage,first', [('S3', False), ('S2', True)])
def test_prox_iteration_vs_oracle(emu_lib, stage, first):
    import __graft_entry__ as ge
    from lemo_amd.prox import LOSS_KEYS
    prob = ge.prox_small_problem(stage=stage)
    of = ge.prox_oracle_for(prob, first_batch_flag=first)
    old = of.closure()
    fit, bm = ge.prox_fitter_for(prob, 'cpu', first_batch_flag=first, lib=emu_lib)
    ld = fit.closure()
    assert tuple(ld.keys()) == LOSS_KEYS
    for k in LOSS_KEYS:
        a, b = float(ld[k]), float(old[k])
        assert abs(a - b) <= 1e-5 * abs(b) + 1e-12, (k, a, b)
    # the synthetic scene really exercises the scene terms
    assert float(old['sdf_penetration_loss']) > 0 and float(old['loss_fric_tangent']) > 0
    if stage == 'S3':
        assert float(old['motion_infill_loss']) > 0 and float(old['motion_infill_contact_loss']) > 0
    pairs = [(fit.pose_embedding.grad, of.pose_embedding.grad)] + \
            [(getattr(bm, n).grad, of.p[n].grad) for n in ('transl', 'global_orient', 'left_hand_pose', 'expression', 'jaw_pose')]
    for a, b in pairs:
        assert rel_err(a, b) < 2e-4
    n_erase = int(prob['B'] * 0.15)
    if first:
        assert float(fit.pose_embedding.grad[:n_erase].abs().max()) > 0
    else:
        assert float(fit.pose_embedding.grad[:n_erase].abs().max()) == 0.0 and float(bm.transl.grad[:n_erase].abs().max()) == 0.0
    for _ in range(2):
        o = of.step()
        l = fit.step()
    assert abs(float(l['total_loss']) - o['total_loss']) <= 1e-4 * abs(o['total_loss'])
    assert float((fit.pose_embedding.detach() - of.pose_embedding.detach()).abs().max()) < 1e-4
