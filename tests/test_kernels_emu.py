"""CPU suite: the UNMODIFIED HIP kernel sources, compiled for the host against tests/hipemu
(fibers + MFMA lane-map emulation), checked against the oracle at small sizes.  This validates
index arithmetic / lane maps / host orchestration without a GPU; the real parity gate is
tests/test_gpu_parity.py (-m gpu) on the MI355X."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, rel_err
from lemo_amd import synthetic
from lemo_amd._hip import ptr
from lemo_amd.priors import cg8p_alloc, from_cg8p, pack_conv3x3, pack_conv3x3_bwd, to_cg8p
from oracle import lemo_oracle as O


@pytest.mark.parametrize('variant', [0, 1, 2])
@pytest.mark.parametrize('ci,co', [(8, 32), (32, 64), (64, 64), (64, 32)])
def test_conv3x3_mfma_forward_and_backward_data(emu_lib, ci, co, variant):
    if variant >= 1 and ci % 16:
        pytest.skip('balanced / LDS variants need Cin % 16 == 0')
    from lemo_amd.priors import pack_conv3x3_gmajor, pack_conv3x3_bwd_gmajor
    g = torch.Generator().manual_seed(ci + co)
    H, W = 7, 41                                        # P = 287 = 2 full 128-px blocks + 31 (ragged tail)
    x, w, b = torch.randn(ci, H, W, generator=g), torch.randn(co, ci, 3, 3, generator=g) * 0.1, torch.randn(co, generator=g)
    ref = F.leaky_relu(F.conv2d(x[None], w, b, padding=1), 0.2)[0]
    xin, out, wt = to_cg8p(x), cg8p_alloc(co, H, W, 'cpu'), torch.from_numpy(pack_conv3x3(w.numpy()))
    if variant == 2:
        wt2 = torch.from_numpy(pack_conv3x3_gmajor(w.numpy()))
        assert emu_lib.conv3x3_mfma_lds(ptr(xin), ptr(wt), ptr(wt2), ptr(b), None, ptr(out), H, W, ci, co, 0, None) == 0
    else:
        assert emu_lib.conv3x3_mfma(ptr(xin), ptr(wt), ptr(b), None, ptr(out), H, W, ci, co, 0, variant, None) == 0
    assert rel_err(from_cg8p(out, H, W), ref) < 2e-6
    assert float(out.reshape(co // 8, H + 2, W + 2, 8)[:, 0].abs().max()) == 0.0       # border untouched
    if ci % 32 == 0:
        dy, aux = torch.randn(co, H, W, generator=g), torch.randn(ci, H, W, generator=g)
        xr = x.clone().requires_grad_(True)
        F.conv2d(xr[None], w, b, padding=1).backward(dy[None])
        refdx = xr.grad * torch.where(aux > 0, 1.0, 0.2)
        dyb, auxb, dxb = to_cg8p(dy), to_cg8p(aux), cg8p_alloc(ci, H, W, 'cpu')
        wtb = torch.from_numpy(pack_conv3x3_bwd(w.numpy()))
        if variant == 2:
            wtb2 = torch.from_numpy(pack_conv3x3_bwd_gmajor(w.numpy()))
            assert emu_lib.conv3x3_mfma_lds(ptr(dyb), ptr(wtb), ptr(wtb2), None, ptr(auxb), ptr(dxb), H, W, co, ci, 1, None) == 0
        else:
            assert emu_lib.conv3x3_mfma(ptr(dyb), ptr(wtb), None, ptr(auxb), ptr(dxb), H, W, co, ci, 1, variant, None) == 0
        assert rel_err(from_cg8p(dxb, H, W), refdx) < 2e-6


@pytest.mark.parametrize('H,W,ci,co', [(7, 41, 64, 64), (36, 57, 64, 64), (7, 41, 32, 64), (7, 41, 64, 32), (9, 30, 32, 32)])
def test_conv3x3_split_bf16_forward_and_backward_data(emu_lib, H, W, ci, co):
    """conv variant 3 (fp32-exact 3-way bf16 operand split on the bf16 MFMA) against torch fp32 AND float64:
    its error must be of the size of the fp32 convolution's own rounding error, not a bf16-sized one.
    (7, 41): 2 blocks + 31 remainder pixels = 128 patches -> 64 patches per block (generic patch loop);
    (36, 57): 16 blocks + 4 remainder pixels = one patch per block (the headline-shape code path);
    Cin / Cout 32: the one-phase (Cin 32) and 4-wave (Cout 32) instantiations."""
    from lemo_amd.priors import pack_conv3x3_split, pack_conv3x3_bwd_split, bf16_split3
    g = torch.Generator().manual_seed(H * W + ci + 2 * co)
    x, w, b = torch.randn(ci, H, W, generator=g), torch.randn(co, ci, 3, 3, generator=g) * 0.1, torch.randn(co, generator=g)
    hi, mid, lo = bf16_split3(w.numpy())
    assert np.abs(mid).max() <= 2.0 ** -8 * np.abs(w.numpy()).max() and np.abs(lo).max() <= 2.0 ** -16 * np.abs(w.numpy()).max()
    assert emu_lib.conv3x3_split_supported(H, W, ci, co) == 1 and emu_lib.conv3x3_split_supported(H, W, 16, 64) == 0
    ref64 = F.leaky_relu(F.conv2d(x[None].double(), w.double(), b.double(), padding=1), 0.2)[0]
    ref32 = F.leaky_relu(F.conv2d(x[None], w, b, padding=1), 0.2)[0]
    xin, out = to_cg8p(x), cg8p_alloc(co, H, W, 'cpu')
    wt, w3 = torch.from_numpy(pack_conv3x3(w.numpy())), torch.from_numpy(pack_conv3x3_split(w.numpy()).view(np.int16))
    assert emu_lib.conv3x3_mfma_split(ptr(xin), ptr(w3), ptr(wt), ptr(b), None, ptr(out), H, W, ci, co, 0, None) == 0
    got = from_cg8p(out, H, W)
    e_split, e_f32 = rel_err(got.double(), ref64), rel_err(ref32.double(), ref64)
    assert e_split < 2e-6 and e_split < 4 * e_f32, (e_split, e_f32)
    assert float(out.reshape(co // 8, H + 2, W + 2, 8)[:, 0].abs().max()) == 0.0       # border untouched
    dy, aux = torch.randn(co, H, W, generator=g), torch.randn(ci, H, W, generator=g)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr[None], w, b, padding=1).backward(dy[None])
    refdx = xr.grad * torch.where(aux > 0, 1.0, 0.2)
    dyb, auxb, dxb = to_cg8p(dy), to_cg8p(aux), cg8p_alloc(ci, H, W, 'cpu')
    wtb, wb3 = torch.from_numpy(pack_conv3x3_bwd(w.numpy())), torch.from_numpy(pack_conv3x3_bwd_split(w.numpy()).view(np.int16))
    assert emu_lib.conv3x3_mfma_split(ptr(dyb), ptr(wb3), ptr(wtb), None, ptr(auxb), ptr(dxb), H, W, co, ci, 1, None) == 0
    assert rel_err(from_cg8p(dxb, H, W), refdx) < 2e-6


def test_conv_rejects_bad_shapes(emu_lib):
    t = torch.zeros(64)
    assert emu_lib.conv3x3_mfma(ptr(t), ptr(t), ptr(t), None, ptr(t), 4, 4, 12, 32, 0, 0, None) != 0
    assert emu_lib.conv3x3_mfma(ptr(t), ptr(t), None, None, ptr(t), 4, 4, 8, 32, 0, 0, None) != 0
    assert emu_lib.conv3x3_mfma(ptr(t), ptr(t), ptr(t), None, ptr(t), 4, 4, 8, 32, 0, 1, None) != 0


def test_first_layer_and_smooth_loss(emu_lib):
    g = torch.Generator().manual_seed(3)
    H, W = 9, 21
    x, w, b = torch.randn(1, H, W, generator=g), torch.randn(32, 1, 3, 3, generator=g), torch.randn(32, generator=g)
    ref = F.leaky_relu(F.conv2d(x[None], w, b, padding=1), 0.2)[0]
    x0 = torch.zeros(H + 2, W + 2); x0[1:-1, 1:-1] = x[0]
    w9, out = w.reshape(32, 9).contiguous(), cg8p_alloc(32, H, W, 'cpu')
    assert emu_lib.conv3x3_c1(ptr(x0), ptr(w9), ptr(b), ptr(out), H, W, 32, None) == 0
    assert rel_err(from_cg8p(out, H, W), ref) < 1e-6
    dpre = torch.randn(32, H, W, generator=g)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr[None], w, b, padding=1).backward(dpre[None])
    dpb, dx0 = to_cg8p(dpre), torch.zeros(H, W)
    assert emu_lib.conv3x3_c1_bwd(ptr(dpb), ptr(w9), ptr(dx0), H, W, 32, None) == 0
    assert rel_err(dx0, xr.grad[0]) < 1e-5
    z = torch.randn(64, H, W, generator=g).requires_grad_(True)
    zz = F.leaky_relu(z, 0.2)
    loss = torch.mean((zz[..., 1:] - zz[..., :-1]) ** 2)
    (loss * 1e6).backward()
    nb, cnt = emu_lib.smooth_loss_blocks(H, W, 64), 64 * H * (W - 1)
    part, dp, zb = torch.zeros(nb), cg8p_alloc(64, H, W, 'cpu'), to_cg8p(zz.detach())
    assert emu_lib.smooth_loss(ptr(zb), ptr(dp), ptr(part), H, W, 64, 1e6 * 2 / cnt, None) == 0
    assert abs(float(part.double().sum() / cnt) - float(loss)) < 1e-6 * float(loss)
    assert rel_err(from_cg8p(dp, H, W), z.grad) < 1e-6


def test_rot6d_and_vposer_vs_golden(emu_lib):
    from lemo_amd.vposer import VPoser, make_vposer_weights
    r = np.load(os.path.join(GOLDEN, 'rot6d.npz'))
    x6 = torch.from_numpy(r['rot6d_in']).contiguous()
    N = x6.shape[0]
    aa = torch.empty(N, 3)
    assert emu_lib.rot6d_to_aa_fwd(ptr(x6), 6, N, ptr(aa), None) == 0
    assert float((aa - torch.from_numpy(r['rot6d_aa'])).abs().max()) < 5e-6
    gen = torch.Generator().manual_seed(1)
    wv = torch.randn(N, 3, generator=gen)
    xr = x6.clone().requires_grad_(True)
    (O.convert_to_3D_all(xr) * wv).sum().backward()
    dx = torch.empty(N, 6)
    assert emu_lib.rot6d_to_aa_bwd(ptr(x6), 6, ptr(wv), N, ptr(dx), None) == 0
    err = (dx - xr.grad).abs().max(1).values / (xr.grad.abs().max(1).values + 1e-9)
    assert float(err.max()) < 1e-4                       # incl. the near-identity and near-pi rows

    g = np.load(os.path.join(GOLDEN, 'vposer_decode.npz'))
    vp = VPoser(_lib=emu_lib).eval()
    w = make_vposer_weights(2)
    vp.load_state_dict({**vp.state_dict(), **{k: torch.from_numpy(v) for k, v in w.items()}})
    Z = torch.from_numpy(g['Z']).clone().requires_grad_(True)
    aa = vp.decode(Z, 'aa')
    assert aa.shape == (16, 1, 21, 3) and rel_err(aa.detach(), g['aa']) < 1e-4
    assert rel_err(vp.decode(Z, 'matrot').detach(), g['matrot']) < 1e-4
    wa = torch.randn(aa.shape, generator=gen)
    (aa * wa).sum().backward()
    Z2 = torch.from_numpy(g['Z']).clone().requires_grad_(True)
    (O.vposer_decode({k: torch.from_numpy(v) for k, v in w.items()}, Z2, 'aa') * wa).sum().backward()
    assert rel_err(Z.grad, Z2.grad) < 1e-4
    with pytest.raises(RuntimeError):
        vp.train().decode(Z, 'aa')


def test_smplx_module_forward_backward_vs_golden(emu_lib):
    """smplx-compatible module: vertices, 127 joints, full pose and every input gradient."""
    from lemo_amd.body_model import create
    g = np.load(os.path.join(GOLDEN, 'lbs_small.npz'))
    m = synthetic.make_synthetic_smplx(seed=int(g['model_seed']), V=int(g['model_V']), F=1200)
    model = create(m, model_type='smplx', gender='male', batch_size=4, num_pca_comps=12,
                   extra_joint_ids=g['extra_ids'].tolist(), _lib=emu_lib, some_unknown_kwarg=1)
    assert model.get_num_verts() == 640 and model.faces_tensor.shape == (1200, 3)
    p = {k: torch.from_numpy(g[k]).clone().requires_grad_(True) for k in ('betas', 'global_orient', 'body_pose', 'lh', 'rh', 'transl')}
    out = model(betas=p['betas'], global_orient=p['global_orient'], body_pose=p['body_pose'], left_hand_pose=p['lh'],
                right_hand_pose=p['rh'], transl=p['transl'], return_verts=True, return_full_pose=True)
    assert out.vertices.shape == (4, 640, 3) and out.joints.shape == (4, 127, 3)
    assert rel_err(out.vertices.detach(), g['verts']) < 1e-4          # north_star: <= 1e-4 rel on vertices
    assert rel_err(out.joints.detach(), g['joints']) < 1e-4
    assert rel_err(out.full_pose.detach(), g['full_pose']) < 1e-6
    ((out.vertices * torch.from_numpy(g['wv'])).sum() + (out.joints * torch.from_numpy(g['wj'])).sum()).backward()
    for k in p:
        assert rel_err(p[k].grad, g['g_' + k]) < 1e-4, k
    # parameters are used when arguments are omitted; reset_params fills / zeroes
    model.reset_params(transl=g['transl'], betas=g['betas'], body_pose=np.zeros((4, 63)), unknown=1)
    assert torch.allclose(model.transl.detach(), torch.from_numpy(g['transl'])) and float(model.jaw_pose.abs().max()) == 0
    o2 = model(global_orient=p['global_orient'].detach(), body_pose=p['body_pose'].detach(),
               left_hand_pose=p['lh'].detach(), right_hand_pose=p['rh'].detach())
    assert rel_err(o2.vertices.detach(), g['verts']) < 1e-4
    mapper = lambda j: j[:, [0, 5, 126]]
    model.joint_mapper = mapper
    assert model(global_orient=p['global_orient'].detach(), body_pose=p['body_pose'].detach()).joints.shape == (4, 3, 3)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('full,coherent', [(True, False), (False, False), (True, True)])
def test_fit_iteration_vs_oracle(emu_lib, full, coherent):
    """whole AMASS iteration through the native engine (C-ABI lemo_fit_*): six loss scalars, total,
    gradients, and parameters after 3 Adam steps -- full-vertex and active-vertex forward; on the i.i.d.-joint synthetic model
    and on the one with the licensed model's index locality (vertices with 1 .. 4 skinning joints, ELL rows padded with weight 0)."""
    import __graft_entry__ as ge
    from lemo_amd.fitting import AmassTemporalFitter
    prob = ge.small_problem(coherent=coherent)
    ofit, markers = ge.oracle_for(prob)
    total, parts, _, verts = ofit.losses()
    total.backward()
    fit = AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'],
                              prob['B'], 'cpu', full_vertices=full, lib=emu_lib)
    fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    fit.forward()
    # the engine builds the marker image and applies the first encoder layer in one launch (marker_c1_kernel); the
    # stand-alone layer (C-ABI lemo_conv3x3_c1) on the image it published must give the same activations bit for bit
    from lemo_amd.priors import ENC_CHANNELS
    ref1 = torch.zeros_like(fit.act[1])
    assert emu_lib.conv3x3_c1(ptr(fit.ws['x0']), ptr(fit.enc.w[0]), ptr(fit.enc.b[0]), ptr(ref1), fit.H, fit.W, ENC_CHANNELS[1], None) == 0
    assert float(fit.ws['x0'].abs().max()) > 0 and torch.equal(ref1, fit.act[1])
    fit.backward()
    L = fit.losses()
    for k in ('marker', 'vposer', 'shape', 'hand', 'contact', 'smooth'):
        assert abs(L[k] - float(parts[k])) <= 1e-5 * abs(float(parts[k])), (k, L[k], float(parts[k]))
    assert abs(L['total'] - float(total)) <= 1e-5 * float(total)
    if full:
        assert rel_err(fit.vertices(), verts.detach()) < 1e-4
    g = fit.grads_with_priors()
    for k, ref in (('transl', ofit.transl.grad), ('rot6d', ofit.rot6d.grad), ('other', ofit.other.grad)):
        assert rel_err(g[k], ref) < 2e-4, k
    fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    ofit.opt.zero_grad()
    # One Adam step must agree to fp32 rounding.  Later steps may not: the contact term selects `x[x > thr]`
    # (opt_amass_temp.py:429-443), so a velocity within an ulp of the threshold is in the mean on one side and out on
    # the other; that moves a handful of gradient entries by ~1e-3 relative and parameters by ~1e-4 (lr = 1e-2).
    ofit.step()
    fit.step(1, use_graph=False)
    assert float((fit.params75() - ofit.params75()).abs().max()) < 1e-6
    for _ in range(2):
        ofit.step()
    fit.step(2, use_graph=False)
    d = (fit.params75() - ofit.params75()).abs()
    assert float(d.max()) < 1e-3 and float(d.mean()) < 2e-5, (float(d.max()), float(d.mean()))
    assert int(fit.step_ctr.item()) == 3


@pytest.mark.timeout(900)
def test_fit_gradient_long_clip_vs_oracle(emu_lib):
    """B = 20 (> 18 difference columns): dverts_vertex takes its prefetched form, where the reflected copies of the
    smoothness image are mutually exclusive (loss_device.hpp); the 14-frame problem above takes the generic loop."""
    import __graft_entry__ as ge
    from lemo_amd.fitting import AmassTemporalFitter
    prob = ge.small_problem(B=20)
    ofit, markers = ge.oracle_for(prob)
    total, parts, _, _ = ofit.losses()
    total.backward()
    fit = AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'],
                              prob['B'], 'cpu', full_vertices=False, lib=emu_lib)
    fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    fit.forward()
    fit.backward()
    L = fit.losses()
    assert abs(L['total'] - float(total)) <= 1e-5 * float(total)
    g = fit.grads_with_priors()
    for k, ref in (('transl', ofit.transl.grad), ('rot6d', ofit.rot6d.grad), ('other', ofit.other.grad)):
        assert rel_err(g[k], ref) < 2e-4, k


@pytest.mark.timeout(900)
@pytest.mark.parametrize('V,foot,nmin', [(640, 75, 256), (1300, 262, 1024)])
def test_fit_gradient_large_vertex_set_vs_oracle(emu_lib, V, foot, nmin):
    """a loss-carrying set of more than 256 vertices: the fused d(verts) stage of the LBS backward (256 lanes) takes
    more than one trip per lane; more than 1024: the set no longer fits the staged kernel, and the engine falls back
    to dverts_assemble + the dense LBS backward."""
    import __graft_entry__ as ge
    from lemo_amd.fitting import AmassTemporalFitter
    prob = ge.small_problem(B=12, V=V, foot=foot)
    ofit, markers = ge.oracle_for(prob)
    total, parts, _, _ = ofit.losses()
    total.backward()
    fit = AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'],
                              prob['B'], 'cpu', full_vertices=True, lib=emu_lib)
    assert fit.n > nmin
    fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    fit.forward()
    fit.backward()
    L = fit.losses()
    for k in ('marker', 'contact', 'smooth'):
        assert abs(L[k] - float(parts[k])) <= 1e-5 * abs(float(parts[k])), (k, L[k], float(parts[k]))
    g = fit.grads_with_priors()
    for k, ref in (('transl', ofit.transl.grad), ('rot6d', ofit.rot6d.grad), ('other', ofit.other.grad)):
        assert rel_err(g[k], ref) < 2e-4, k


def test_contact_term_empty_selection_is_exactly_zero(emu_lib):
    """K15: `x[x>thr].mean()` with an empty selection must be exactly 0 (opt_amass_temp.py:429-443)."""
    import __graft_entry__ as ge
    from lemo_amd.fitting import AmassTemporalFitter
    prob = ge.small_problem()
    _, markers = ge.oracle_for(prob)
    fit = AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'],
                              prob['B'], 'cpu', full_vertices=False, lib=emu_lib)
    p = prob['seq']['init_params'].copy()
    p[:] = p[0]                                          # a static pose: every contact velocity is 0
    fit.load_sequence(p, markers, np.ones_like(prob['seq']['contact_lbl']))
    fit.forward()
    fit.backward()
    assert fit.losses()['contact'] == 0.0
    assert not torch.isnan(fit.grads()['other']).any()
    fit.load_sequence(prob['seq']['init_params'], markers, np.zeros_like(prob['seq']['contact_lbl']))
    fit.forward()
    assert fit.losses()['contact'] == 0.0                # no contact labels at all


@pytest.mark.parametrize('H,W,ci,co', [(7, 41, 64, 64), (36, 57, 64, 64), (7, 41, 32, 64), (7, 41, 64, 32), (9, 30, 32, 32)])
def test_conv3x3_split_f16_forward_and_backward_data(emu_lib, H, W, ci, co):
    """conv variant 4 (two error-compensated fp16 pieces per operand, three products, per-workgroup power-of-two scaling)
    against torch fp32 AND float64: its error must be of the size of the fp32 convolution's own rounding error.  The
    'range' input spans 8 orders of magnitude across the image and 3 more across the two channel phases -- what the
    per-workgroup, per-phase scale is for (a per-tensor scale would flush most of it to fp16 denormals) -- and the result is
    exactly homogeneous under power-of-two scalings of the input."""
    from lemo_amd.priors import pack_conv3x3_split_f16, pack_conv3x3_bwd_split_f16, f16_split2
    g = torch.Generator().manual_seed(H * W + ci + 2 * co + 1)
    x, w, b = torch.randn(ci, H, W, generator=g), torch.randn(co, ci, 3, 3, generator=g) * 0.1, torch.randn(co, generator=g)
    hi, lo, winv = f16_split2(w.numpy())
    assert np.abs(hi.astype(np.float64) + lo.astype(np.float64) - w.numpy().astype(np.float64) / winv).max() <= 2.0 ** -22 * np.abs(w.numpy() / winv).max()
    assert 2.0 ** 14 <= np.abs(w.numpy() / winv).max() < 2.0 ** 15
    wt = torch.from_numpy(pack_conv3x3(w.numpy()))
    pf, fi = pack_conv3x3_split_f16(w.numpy())
    w4 = torch.from_numpy(pf.view(np.int16))
    for case in (('plain', 'range') if H >= 30 else ('plain',)):
        xx = x.clone()
        if case == 'range':
            # magnitudes fall by 8 orders from the first image row to the last (smoothly: a workgroup's tile is a few rows of
            # a vertical strip + halo and takes its scale from the largest value it staged) and channels 32.. sit 1e-3 lower
            xx *= (10.0 ** (-8.0 * torch.arange(H) / H))[None, :, None]
            xx[32:] *= 1e-3
        bb = b * (0 if case == 'range' else 1)
        ref64 = F.leaky_relu(F.conv2d(xx[None].double(), w.double(), bb.double(), padding=1), 0.2)[0]
        ref32 = F.leaky_relu(F.conv2d(xx[None], w, bb, padding=1), 0.2)[0]
        xin, out = to_cg8p(xx), cg8p_alloc(co, H, W, 'cpu')
        assert emu_lib.conv3x3_mfma_split_f16(ptr(xin), ptr(w4), fi, ptr(wt), ptr(bb), None, ptr(out), H, W, ci, co, 0, None) == 0
        got = from_cg8p(out, H, W).double()
        if case == 'plain':
            e_split, e_f32 = rel_err(got, ref64), rel_err(ref32.double(), ref64)
            assert e_split < 2e-6 and e_split < 4 * e_f32, (e_split, e_f32)
        else:
            # every row against the largest reference value within +-12 rows (what a tile can hold): fp32-sized everywhere,
            # from 1e0-sized rows down to 1e-8-sized ones -- a per-tensor scale would leave the lower rows in fp16 denormals
            for y in range(H):
                loc = ref64[:, max(0, y - 12):y + 13].abs().max()
                e = float((got[:, y] - ref64[:, y]).abs().max() / loc)
                assert e < 3e-6, (y, e)
            # exact homogeneity: the per-workgroup scales are powers of two, so scaling the input by 2^k scales the output by
            # exactly 2^k (bias 0, LeakyReLU is homogeneous) -- no overflow into fp16 inf, no underflow, at 2^-30 and 2^+20
            for k in (-30, 20):
                xs, outk = to_cg8p(xx * 2.0 ** k), cg8p_alloc(co, H, W, 'cpu')
                assert emu_lib.conv3x3_mfma_split_f16(ptr(xs), ptr(w4), fi, ptr(wt), ptr(bb), None, ptr(outk), H, W, ci, co, 0, None) == 0
                assert torch.equal(outk, out * 2.0 ** k), k
        assert float(out.reshape(co // 8, H + 2, W + 2, 8)[:, 0].abs().max()) == 0.0       # border untouched
    dy, aux = torch.randn(co, H, W, generator=g) * 1e-6, torch.randn(ci, H, W, generator=g)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr[None], w, b, padding=1).backward(dy[None])
    refdx = xr.grad * torch.where(aux > 0, 1.0, 0.2)
    dyb, auxb, dxb = to_cg8p(dy), to_cg8p(aux), cg8p_alloc(ci, H, W, 'cpu')
    pb, bi = pack_conv3x3_bwd_split_f16(w.numpy())
    wtb, wb4 = torch.from_numpy(pack_conv3x3_bwd(w.numpy())), torch.from_numpy(pb.view(np.int16))
    assert emu_lib.conv3x3_mfma_split_f16(ptr(dyb), ptr(wb4), bi, ptr(wtb), None, ptr(auxb), ptr(dxb), H, W, co, ci, 1, None) == 0
    assert rel_err(from_cg8p(dxb, H, W), refdx) < 2e-6
    # all-zero input: scale clamps, output = lrelu(bias) exactly
    z = cg8p_alloc(ci, H, W, 'cpu')
    assert emu_lib.conv3x3_mfma_split_f16(ptr(z), ptr(w4), fi, ptr(wt), ptr(b), None, ptr(out), H, W, ci, co, 0, None) == 0
    assert torch.equal(from_cg8p(out, H, W), F.leaky_relu(b, 0.2)[:, None, None].expand(co, H, W))


@pytest.mark.parametrize('coherent', [False, True])
def test_dense_vertex_backward_large_set(emu_lib, coherent):
    """vertex sets above 1024 vertices (the PROX window differentiates through all of them) take the chunked dense
    backward (joint-major gather per 512-vertex chunk, per-chunk partials reduced in order: no atomics): every input
    gradient against the oracle's autograd.  i.i.d. skinning weights: ~35 entries per joint and chunk (the thread-per-output
    form of the joint-major sums, LBS_JOINT_SMALL); weights with index locality: a few joints with hundreds of entries each (a
    wave per joint) next to joints with a handful -- both forms in one chunk"""
    from lemo_amd.body_model import create
    from oracle import lemo_oracle as O
    m = synthetic.make_synthetic_smplx(seed=5, V=1300, F=600, coherent=coherent)
    B = 3
    g = torch.Generator().manual_seed(4)
    mk = lambda *s, sc=0.3: (torch.randn(*s, generator=g) * sc)
    vals = dict(betas=mk(B, 10, sc=0.5), global_orient=mk(B, 3), body_pose=mk(B, 63), lh=mk(B, 12, sc=0.1), rh=mk(B, 12, sc=0.1),
                transl=mk(B, 3))
    wv = torch.randn(B, 1300, 3, generator=g)
    model = create(m, batch_size=B, num_pca_comps=12, extra_joint_ids=list(range(21)), _lib=emu_lib)
    p = {k: v.clone().requires_grad_(True) for k, v in vals.items()}
    out = model(betas=p['betas'], global_orient=p['global_orient'], body_pose=p['body_pose'], left_hand_pose=p['lh'],
                right_hand_pose=p['rh'], transl=p['transl'])
    (out.vertices * wv).sum().backward()
    so = O.SmplxOracle(m, extra_joint_ids=list(range(21)))
    q = {k: v.clone().requires_grad_(True) for k, v in vals.items()}
    v_ref, _, _ = so.forward(q['betas'], q['global_orient'], q['body_pose'], q['lh'], q['rh'], q['transl'])
    (v_ref * wv).sum().backward()
    assert rel_err(out.vertices.detach(), v_ref.detach()) < 1e-4
    for k in p:
        assert rel_err(p[k].grad, q[k].grad) < 1e-4, k
    db = model._device_body(torch.device('cpu'))
    st, tt = db.vertex_set('all', np.arange(1300), frames=B)
    assert 'jcsr_chunk' in tt and st.part_frames >= B                   # the deterministic path was the one that ran
    # scratch is private per caller: the cached struct carries none, two requests never share a buffer, and a
    # larger request does not touch the smaller one's; the immutable tables are shared
    shared, stt = db.vertex_set('all', np.arange(1300))
    assert shared.part_frames == 0 and not shared.part and 'part' not in stt
    st2, tt2 = db.vertex_set('all', np.arange(1300), frames=4 * B)
    assert tt2['part'].data_ptr() != tt['part'].data_ptr() and tt2['gemm_part'].data_ptr() != tt['gemm_part'].data_ptr()
    assert st.part == tt['part'].data_ptr() and st.part_frames == B and tt['part'].shape[0] == B
    assert tt2['Dk'].data_ptr() == tt['Dk'].data_ptr() and st2.DkG == st.DkG == shared.DkG


@pytest.mark.parametrize('ci,co,ks', [(64, 64, 4), (40, 32, 7), (128, 64, 8)])
def test_conv3x3_splitk_matches_reference(emu_lib, ci, co, ks):
    """variant 0 with the K reduction split over ks grid slices + the combine pass (uneven slice sizes included),
    forward and backward-data epilogues"""
    g = torch.Generator().manual_seed(ci + co + ks)
    H, W = 9, 17
    x, w, b = torch.randn(ci, H, W, generator=g), torch.randn(co, ci, 3, 3, generator=g) * 0.1, torch.randn(co, generator=g)
    ref = F.leaky_relu(F.conv2d(x[None], w, b, padding=1), 0.2)[0]
    xin, out, wt = to_cg8p(x), cg8p_alloc(co, H, W, 'cpu'), torch.from_numpy(pack_conv3x3(w.numpy()))
    part = torch.full((ks * (co // 8) * (H + 2) * (W + 2) * 8,), float('nan'))
    assert emu_lib.conv3x3_mfma_splitk(ptr(xin), ptr(wt), ptr(b), None, ptr(out), ptr(part), ks, H, W, ci, co, 0, None) == 0
    assert rel_err(from_cg8p(out, H, W), ref) < 2e-6
    assert float(out.reshape(co // 8, H + 2, W + 2, 8)[:, 0].abs().max()) == 0.0
    aux = torch.randn(co, H, W, generator=g)
    auxb = to_cg8p(aux)
    assert emu_lib.conv3x3_mfma_splitk(ptr(xin), ptr(wt), None, ptr(auxb), ptr(out), ptr(part), ks, H, W, ci, co, 1, None) == 0
    ref1 = F.conv2d(x[None], w, None, padding=1)[0] * torch.where(aux > 0, 1.0, 0.2)
    assert rel_err(from_cg8p(out, H, W), ref1) < 2e-6
    assert emu_lib.conv3x3_mfma_splitk(ptr(xin), ptr(wt), ptr(b), None, ptr(out), ptr(part), 9 * (ci // 8) + 1, H, W, ci, co, 0, None) != 0


@pytest.mark.timeout(600)
def test_real_file_shaped_model_400_shapedirs_and_kw_above_8(emu_lib):
    """a model shaped like the licensed SMPLX_*.npz -- shapedirs [V,3,400] (300 shape + 100 expression directions: the
    expression block is read at [300:310]) and skinning rows with more than 8 non-zeros (the ELL tail loop of the vertex
    kernel, KW > 8) -- forward and every input gradient against the oracle"""
    from lemo_amd.body_model import create
    from oracle import lemo_oracle as O
    m = synthetic.make_synthetic_smplx(seed=9, V=640, F=1200, n_shape=400, nnz_w=11)
    assert m['shapedirs'].shape == (640, 3, 400) and int((m['weights'] != 0).sum(1).max()) > 8
    B = 3
    g = torch.Generator().manual_seed(6)
    mk = lambda *s, sc=0.3: (torch.randn(*s, generator=g) * sc)
    vals = dict(betas=mk(B, 10, sc=0.5), expression=mk(B, 10, sc=0.5), global_orient=mk(B, 3), body_pose=mk(B, 63),
                lh=mk(B, 12, sc=0.1), rh=mk(B, 12, sc=0.1), transl=mk(B, 3), jaw=mk(B, 3, sc=0.1))
    wv, wj = torch.randn(B, 640, 3, generator=g), torch.randn(B, 127, 3, generator=g)
    model = create(m, batch_size=B, num_pca_comps=12, extra_joint_ids=list(range(21)), _lib=emu_lib)
    assert model._device_body(torch.device('cpu')).data.KW > 8
    p = {k: v.clone().requires_grad_(True) for k, v in vals.items()}
    out = model(betas=p['betas'], expression=p['expression'], global_orient=p['global_orient'], body_pose=p['body_pose'],
                left_hand_pose=p['lh'], right_hand_pose=p['rh'], transl=p['transl'], jaw_pose=p['jaw'])
    ((out.vertices * wv).sum() + (out.joints * wj).sum()).backward()
    so = O.SmplxOracle(m, extra_joint_ids=list(range(21)))
    q = {k: v.clone().requires_grad_(True) for k, v in vals.items()}
    v_ref, j_ref, _ = so.forward(q['betas'], q['global_orient'], q['body_pose'], q['lh'], q['rh'], q['transl'],
                                 expression=q['expression'], jaw_pose=q['jaw'])
    ((v_ref * wv).sum() + (j_ref * wj).sum()).backward()
    assert rel_err(out.vertices.detach(), v_ref.detach()) < 1e-4 and rel_err(out.joints.detach(), j_ref.detach()) < 1e-4
    # the expression directions really are columns 300..309: zeroing them changes the vertices
    m0 = dict(m, shapedirs=m['shapedirs'].copy()); m0['shapedirs'][:, :, 300:310] = 0
    v0, _, _ = O.SmplxOracle(m0, extra_joint_ids=list(range(21))).forward(vals['betas'], vals['global_orient'], vals['body_pose'],
                                                                         vals['lh'], vals['rh'], vals['transl'], expression=vals['expression'])
    assert float((v0 - v_ref.detach()).abs().max()) > 1e-3
    for k in p:
        assert rel_err(p[k].grad, q[k].grad) < 2e-4, k


def test_presplit_blend_operand_is_bit_identical(emu_lib):
    """lemo_lbs_verts_fwd_xs (per-frame features pre-split into bf16 pieces by the pose kernel, lemo_pose_ws.XgS) against
    lemo_lbs_verts_fwd (every workgroup converts them itself): same pieces, same products in the same order -> same bits;
    and XgS really is the exact 3-way split of Xg"""
    import ctypes as C
    from lemo_amd._hip import ptr
    from lemo_amd.body_model import BodyModelData, DeviceBody, alloc_pose_ws
    lib = emu_lib
    data = BodyModelData(synthetic.make_synthetic_smplx(seed=3, V=200, F=300))
    db = DeviceBody(data, 'cpu', blend_f16=False)
    B = 5
    ws, tt, Bp = alloc_pose_ws(B, data.nj, 'cpu', False)
    g = torch.Generator().manual_seed(1)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).contiguous()
    go, body, lh, rh, betas = r(B, 3), r(B, 63), r(B, 12), r(B, 12), r(B, 10)
    z3 = torch.zeros(B, 3)
    expr = torch.zeros(B, 10)
    from lemo_amd import _hip
    pin = _hip.PoseIn(ptr(go), ptr(body), ptr(z3), ptr(z3), ptr(z3), ptr(lh), ptr(rh), 12, ptr(betas), 10, ptr(expr))
    lib.check(lib.smplx_pose_fwd(C.byref(db.body), C.byref(pin), C.byref(ws), B, None))
    # XgS[k >> 4][piece][frame][(k >> 3) & 1][k & 7] (bf16 bits) sums back to Xg[k >> 3][frame][k & 7] exactly
    pieces = (tt['XgS'].to(torch.int32) & 0xFFFF) << 16
    xs = pieces.view(torch.float32).double().sum(1)                                      # [K/16][Bp][2][8]
    xg = tt['Xg'].view(-1, 2, Bp, 8).permute(0, 2, 1, 3).double()                        # [K/16][Bp][2][8]
    assert torch.equal(xs, xg) and float(xg.abs().max()) > 0
    out = []
    for pre in (False, True):
        v, vp = torch.empty(B, data.V, 3), torch.empty(B, data.V, 3)
        if pre:
            lib.check(lib.lbs_verts_fwd_xs(C.byref(db.skin), ptr(tt['Xg']), ptr(tt['XgS']), Bp, ptr(tt['A']), data.nj, None, None, data.V, B,
                                           ptr(v), ptr(vp), None))
        else:
            lib.check(lib.lbs_verts_fwd(C.byref(db.skin), ptr(tt['Xg']), Bp, ptr(tt['A']), data.nj, None, None, data.V, B, ptr(v), ptr(vp), None))
        out.append((v, vp))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert float(out[0][0].abs().max()) > 0


def test_f16_presplit_blend_is_fp32_sized(emu_lib):
    """the blend GEMM with BOTH operands pre-split into two fp16 pieces (lemo_skin_const.DgH made on the host, XgS written in its
    fp16 form by the pose kernel; 3 fp16 MFMA products per k-chunk) against a float64 evaluation of the same blend + skinning:
    the error is the fp32 kernel's (fp32-MFMA path, blend_fp32 = 1) to within a small factor, far inside the 1e-4 vertex gate;
    the pieces carry their operands to 2^-22; ragged frame count, all vertices and a vertex subset"""
    import ctypes as C
    from lemo_amd import _hip
    from lemo_amd._hip import ptr
    from lemo_amd.body_model import BodyModelData, DeviceBody, alloc_pose_ws, split_f16_pairs
    lib = emu_lib
    data = BodyModelData(synthetic.make_synthetic_smplx(seed=3, V=200, F=300))
    h, inv = split_f16_pairs(data.Dg)
    hp = h.view(np.float16).reshape(data.Dg.shape[0], data.Dg.shape[1], 2, 2, 4).astype(np.float64)
    back = (hp[:, :, :, 0, :] + hp[:, :, :, 1, :]).reshape(data.Dg.shape) * inv
    assert np.abs(back - data.Dg).max() <= np.abs(data.Dg).max() * 2.0 ** -21
    B = 5
    res = {}
    for f16 in (True, False):
        db = DeviceBody(data, 'cpu', blend_f16=f16)
        assert (db.skin.DgH is not None) == f16
        ws, tt, Bp = alloc_pose_ws(B, data.nj, 'cpu', f16)
        g = torch.Generator().manual_seed(1)
        r = lambda *s: (torch.randn(*s, generator=g) * 0.3).contiguous()
        go, body, lh, rh, betas = r(B, 3), r(B, 63), r(B, 12), r(B, 12), r(B, 10)
        z3, expr, tr = torch.zeros(B, 3), r(B, 10), r(B, 3)
        pin = _hip.PoseIn(ptr(go), ptr(body), ptr(z3), ptr(z3), ptr(z3), ptr(lh), ptr(rh), 12, ptr(betas), 10, ptr(expr))
        lib.check(lib.smplx_pose_fwd(C.byref(db.body), C.byref(pin), C.byref(ws), B, None))
        if f16:                                            # XgS[k >> 4][piece][frame][half][8] (fp16 bits) sums back to Xg to 2^-22
            pc = tt['XgS'].view(torch.float16).double().sum(1)
            xg = tt['Xg'].view(-1, 2, Bp, 8).permute(0, 2, 1, 3).double()
            assert float((pc - xg).abs().max()) <= float(xg.abs().max()) * 2.0 ** -21 and float(xg.abs().max()) > 0.1
        v, vp = torch.empty(B, data.V, 3), torch.empty(B, data.V, 3)
        lib.check(lib.lbs_verts_fwd_xs(C.byref(db.skin), ptr(tt['Xg']), ptr(tt['XgS']), Bp, ptr(tt['A']), data.nj, ptr(tr), None, data.V, B,
                                       ptr(v), ptr(vp), None))
        ids = torch.arange(3, data.V, 7, dtype=torch.int32)
        vs = torch.empty(B, len(ids), 3)
        lib.check(lib.lbs_verts_fwd_xs(C.byref(db.skin), ptr(tt['Xg']), ptr(tt['XgS']), Bp, ptr(tt['A']), data.nj, ptr(tr), ptr(ids), len(ids), B,
                                       ptr(vs), None, None))
        assert torch.equal(vs, v[:, ids.long()])
        # float64 reference from the kernel's own inputs: Xg features, D, A, skinning weights
        X = tt['Xg'].double().permute(1, 0, 2).reshape(Bp, -1)[:B]                       # [B][512]
        vpr = (X @ torch.from_numpy(data.D).double()).reshape(B, data.V, 3) + torch.from_numpy(data.v_template).double()
        A = tt['A'].double().reshape(B, data.nj, 3, 4)
        Wd = torch.zeros(data.V, data.nj, dtype=torch.float64)
        wi, wv = torch.from_numpy(data.w_idx).long(), torch.from_numpy(data.w_val).double()
        Wd.scatter_add_(1, wi, wv)
        T = torch.einsum('vj,bjrc->bvrc', Wd, A)
        ref = torch.einsum('bvrc,bvc->bvr', T[..., :3], vpr) + T[..., 3] + tr.double()[:, None]
        res[f16] = (float((v.double() - ref).abs().max()), float((vp.double() - vpr).abs().max()), float(ref.abs().max()))
    (e16, p16, scale), (e32, p32, _) = res[True], res[False]
    assert e16 < 4e-7 * max(1.0, scale) and p16 < 4e-7 * max(1.0, scale), res
    assert e16 < 4 * e32 + 1e-7 and p16 < 4 * p32 + 1e-7, res


@pytest.mark.parametrize('grouped', [False, True])
def test_splitk_gemm_matches_float64(emu_lib, grouped):
    """lemo_gemm_nt16_splitk (the feature-gradient GEMM of the all-vertex LBS backward): K slabs on the bf16 matrix cores with
    exactly split fp32 operands against a float64 product -- fp32-sized error (the three dropped products are < 2^-24 each);
    the k-chunk-major copy of A (`A_grouped`) gives the same bits as the row-major one; ragged N and a K that the slabs do not
    divide evenly"""
    from lemo_amd._hip import ptr
    lib = emu_lib
    g = torch.Generator().manual_seed(5)
    for M, N, K, S in ((128, 37, 16 * 41, 6), (192, 100, 16 * 23, 11), (512, 7, 16 * 67, 64)):
        _splitk_case(lib, g, M, N, K, S, grouped)          # 128-row workgroups (M % 128 == 0) and the 64-row form; S > 8 * 8: several reduce rounds


def _splitk_case(lib, g, M, N, K, S, grouped):
    from lemo_amd._hip import ptr
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) * 0.5
    ref = (B.double() @ A.double().t())                                            # C[n][m]
    part = torch.zeros(lib.gemm_nt16_splitk_part_floats(M, S))
    outs = []
    for use_g in (False, grouped):
        Cm = torch.zeros(N, M)
        Ag = A.view(M, K // 16, 16).permute(1, 0, 2).contiguous() if use_g else None
        lib.check(lib.gemm_nt16_splitk(ptr(A), K, ptr(B), K, M, N, K, ptr(Cm), M, ptr(part), S, ptr(Ag) if use_g else None, None))
        outs.append(Cm)
    assert torch.equal(outs[0], outs[1])
    assert float((outs[1].double() - ref).abs().max() / ref.abs().max()) < 2e-6


@pytest.mark.timeout(1800)
@pytest.mark.parametrize('variant', [5, 7, 9])
def test_wide_clip_keeps_the_fused_pairs_and_matches_the_oracle(emu_lib, variant):
    """B = 130 (W = 145 > 134): the single-layer split kernels do not take the image, the fused pairs do -- the engine keeps conv
    variant 5 / 7 / 9 (pairs fused -- 7, 9: head and tail fused as well, any width -- the other encoder launches on the fp32-input kernel layer
    by layer) and the iteration still matches the oracle: losses, gradients, one Adam step"""
    import warnings
    import __graft_entry__ as ge
    from lemo_amd.fitting import AmassTemporalFitter
    prob = ge.small_problem(B=130)
    ofit, markers = ge.oracle_for(prob)
    total, parts, _, verts = ofit.losses()
    total.backward()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        fit = AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'],
                                  prob['B'], 'cpu', full_vertices=True, lib=emu_lib, conv_variant=variant)
    assert fit.conv_variant == variant and any('wider than' in str(x.message) for x in w)
    assert not emu_lib.conv3x3_split_supported(fit.H, fit.W, 64, 64) and emu_lib.conv3x3_pair_supported(fit.H, fit.W, 64, 64, 64)
    fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    fit.forward()
    fit.backward()
    L = fit.losses()
    for k in ('marker', 'vposer', 'shape', 'hand', 'contact', 'smooth'):
        assert abs(L[k] - float(parts[k])) <= 1e-5 * abs(float(parts[k])), (k, L[k], float(parts[k]))
    g = fit.grads_with_priors()
    for k, ref in (('transl', ofit.transl.grad), ('rot6d', ofit.rot6d.grad), ('other', ofit.other.grad)):
        assert rel_err(g[k], ref) < 2e-4, k
    fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    ofit.opt.zero_grad()
    ofit.step()
    fit.step(1, use_graph=False)
    dp = (fit.params75() - ofit.params75()).abs()                   # lr 1e-2: a few entries with rounding-sized gradients among 9750
    assert float(dp.max()) < 2e-5 and float(dp.mean()) < 2e-8, (float(dp.max()), float(dp.mean()))


def test_wide_clip_warns_about_the_slower_encoder_path(emu_lib):
    """clips longer than 120 frames: the image is wider than the split-f16 single-layer kernels stage (W <= 134); the engine still
    fits them (fp32-input fallback for five launches) but says so once at construction; the reference's clip length does not warn"""
    import warnings
    from lemo_amd.priors import warn_if_wide_image
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        assert warn_if_wide_image(emu_lib, 245, 134, 5) is False            # T = 120
        assert warn_if_wide_image(emu_lib, 245, 115, 5) is False            # the 100-frame PROX window
        assert warn_if_wide_image(emu_lib, 245, 253, 2) is False            # fp32 variants have no such limit
    with pytest.warns(RuntimeWarning, match='wider than'):
        assert warn_if_wide_image(emu_lib, 245, 253, 5) is True
