"""Drop-in boundary: one AMASS iteration composed from the MODULE API -- lemo_amd.compat ``smplx.create``,
``VPoser.decode``, ``Enc``, ``convert_to_3D_rot`` with torch autograd doing the backward through the HIP autograd Functions --
on the emulator library, against tests/golden/dropin_amass_small.npz, which holds what the REFERENCE's own loop-body text
(opt_amass_temp.py:355-453) produced on oracle-backed objects.  tests/golden/make_golden.py additionally exec's that text
against these very modules in the build container (rows dropin.* of tests/golden/dropin_vs_reference.txt)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, rel_err


def test_dropin_report_rows():
    rows = dict(l.split('\t') for l in open(os.path.join(GOLDEN, 'dropin_vs_reference.txt')).read().strip().splitlines())
    for k in ('marker', 'vposer', 'shape', 'hand', 'contact', 'smooth', 'total'):
        assert float(rows['dropin.' + k]) <= 1e-5
    for k in ('g_transl', 'g_rot6d', 'g_other'):
        assert float(rows['dropin.' + k]) <= 2e-4
    assert float(rows['dropin.p75_after3']) <= 1e-5
    # PROX side of the boundary: the reference's own FittingMonitor closure + SMPLifyLoss + optim_factory
    # (fitting_temp_slide.py:220-311, 564-1062) ran unmodified on lemo_amd.compat smplx / lemo_amd.vposer.VPoser /
    # lemo_amd.priors.Enc (emulator library) -- tests/golden/make_golden.py::pin_dropin_prox, vs prox_iter.npz
    for stage in ('S2', 'S3'):
        for w in ('first', 'later'):
            t = f'dropin_prox.{stage}_{w}.'
            assert float(rows[t + 'loss_dict']) <= 5e-6
            for k in ('g_pose_embedding', 'g_transl', 'g_global_orient'):
                assert float(rows[t + k]) <= 2e-4
            assert float(rows[t + 'params_after3']) <= 1e-4


@pytest.mark.timeout(900)
def test_amass_iteration_from_module_api_small(emu_lib):
    import functools
    import __graft_entry__ as ge
    from lemo_amd import compat, rotation
    from lemo_amd.priors import Enc
    from lemo_amd.vposer import VPoser
    g = np.load(os.path.join(GOLDEN, 'dropin_amass_small.npz'))
    prob = ge.small_problem()
    B = prob['B']
    compat.install()
    import smplx                                                       # resolves to lemo_amd.compat.smplx
    assert smplx.__name__.startswith('lemo_amd.compat')
    smplx_model = smplx.create(prob['model'], model_type='smplx', gender='male', ext='npz', num_pca_comps=12, batch_size=B,
                               extra_joint_ids=list(range(21)), _lib=emu_lib)
    vposer_model = VPoser(_lib=emu_lib).eval()
    vposer_model.load_state_dict({**vposer_model.state_dict(), **{k: torch.from_numpy(v) for k, v in prob['vposer_w'].items()}})
    smooth_encoder = Enc(_lib=emu_lib)
    smooth_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in prob['enc_w'].items()})
    c3d = functools.partial(rotation.convert_to_3D_rot, _lib=emu_lib)
    ip = torch.from_numpy(np.array(prob['seq']['init_params'], np.float32))
    transl = ip[:, 0:3].clone().requires_grad_(True)
    rot6d = rotation.convert_to_6D_all(ip[:, 3:6]).detach().clone().requires_grad_(True)
    shape_t, other = ip[:, 6:16].clone(), ip[:, 16:].clone().requires_grad_(True)
    ids = {k: torch.as_tensor(np.asarray(v, np.int64)) for k, v in prob['ids'].items()}
    Xmean, Xstd = torch.from_numpy(prob['Xmean']).view(1, 1, -1), torch.from_numpy(prob['Xstd'])
    markers_rec, contact = torch.from_numpy(g['markers_rec']), torch.from_numpy(prob['seq']['contact_lbl'])
    opt = torch.optim.Adam([transl, rot6d, other], lr=0.01)

    def iteration():
        opt.zero_grad()
        p72 = c3d(torch.cat([transl, rot6d, shape_t, other], dim=-1))
        body_pose = vposer_model.decode(p72[:, 16:48], output_type='aa').view(B, -1)
        out = smplx_model(return_verts=True, transl=p72[:, 0:3], global_orient=p72[:, 3:6], betas=p72[:, 6:16], body_pose=body_pose,
                          left_hand_pose=p72[:, 48:60], right_hand_pose=p72[:, 60:])
        verts, joints = out.vertices, out.joints
        ms = verts[:, ids['markers81']]
        j0 = joints[0].detach()
        x = j0[2] - j0[1]
        x = torch.cat([x[:2], x.new_zeros(1)]); x = x / torch.norm(x)
        z = x.new_tensor([0., 0., 1.])
        y = torch.linalg.cross(z, x); y = y / torch.norm(y)
        gm = torch.matmul(ms - ms[0].detach()[0], torch.stack([x, y, z], dim=1))
        img = ((gm.reshape(B, -1).unsqueeze(0) - Xmean) / Xstd).permute(0, 2, 1).unsqueeze(1)
        mz = smooth_encoder(F.pad(img[..., 1:] - img[..., :-1], (8, 8, 1, 1), 'reflect'))[0]
        L = dict(smooth=torch.mean((mz[..., 1:] - mz[..., :-1]) ** 2), marker=F.l1_loss(verts[:, ids['markers67']], markers_rec),
                 vposer=torch.mean(p72[:, 16:48] ** 2), shape=torch.mean(p72[:, 6:16] ** 2), hand=torch.mean(p72[:, 48:] ** 2))
        vel = (verts[1:] - verts[:-1]) * 30
        c = verts.new_zeros(())
        for k, name in enumerate(('left_heel', 'right_heel', 'left_toe', 'right_toe')):
            sp = torch.norm(vel[:, ids[name]][contact[:-1, k] == 1], dim=-1)
            if (sp - 0.1).gt(0).sum().item() >= 1:
                c = c + sp[sp > 0.1].abs().mean()
        L['contact'] = c
        L['total'] = L['marker'] + 0.02 * L['vposer'] + 0.01 * L['shape'] + 0.01 * L['hand'] + 0.03 * c + 1e6 * L['smooth']
        L['total'].backward()
        return L

    L = iteration()
    for k in ('marker', 'vposer', 'shape', 'hand', 'contact', 'smooth', 'total'):
        assert abs(float(L[k]) - float(g[k])) <= 1e-5 * abs(float(g[k])), (k, float(L[k]), float(g[k]))
    assert rel_err(transl.grad, g['g_transl']) < 2e-4 and rel_err(rot6d.grad, g['g_rot6d']) < 2e-4 and rel_err(other.grad, g['g_other']) < 2e-4
    opt.step()
    p75 = torch.cat([transl, rot6d, shape_t, other], -1).detach()
    assert float((p75 - torch.from_numpy(g['p75_after1'])).abs().max()) < 2e-6
