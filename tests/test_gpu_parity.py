"""GPU parity gate (-m gpu, runs on the MI355X through liblemo_hip.so / the C ABI).

Tolerances are BASELINE.json's: <= 1e-4 relative on vertices / markers, <= 1e-5 relative on each
loss scalar, for one iteration from identical inputs; checked against the oracle (recomputed here
on the host) and against the committed golden vectors."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, rel_err
from lemo_amd import synthetic
from lemo_amd.assets import load_assets

pytestmark = pytest.mark.gpu
LOSS_TOL = 1e-5
VERT_TOL = 1e-4


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs the MI355X'
    from lemo_amd import _hip
    assert not _hip.get_lib().is_emu                      # the product library, loaded from the tree
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def full_problem(dev):
    from lemo_amd.fitting import AmassTemporalFitter
    from lemo_amd.vposer import make_vposer_weights
    A = load_assets()
    g = np.load(os.path.join(GOLDEN, 'amass_iter.npz'))
    model = synthetic.make_synthetic_smplx(seed=0)
    seq = synthetic.make_synthetic_sequence(0, B=119)
    mk = lambda full, variant=None: AmassTemporalFitter(model, make_vposer_weights(2), A['enc_w'], A['ids'], A['Xmean'],
                                                        A['Xstd'], 119, dev, full_vertices=full, conv_variant=variant)
    return dict(A=A, g=g, model=model, seq=seq, make=mk)


@pytest.fixture(scope='module')
def kink_exposure(full_problem):
    """COMPUTED per-frame bound on what the objective's kinks can do to the iteration-0 gradient at BASELINE size
    (oracle/f64.py: flip_sensitivity for the encoder's LeakyReLU units, KinkProbe for L1 residuals / contact speeds),
    evaluated once in float64 at the golden (6) inputs -- what the gradient gates use instead of a constant"""
    from oracle.f64 import amass_fit_oracle_f64, default_f64, flip_sensitivity, KinkProbe
    from lemo_amd.vposer import make_vposer_weights
    torch.set_num_threads(32)
    A, g, seq = full_problem['A'], full_problem['g'], full_problem['seq']
    o64 = amass_fit_oracle_f64(full_problem['model'], make_vposer_weights(2), A['enc_w'], A['ids'], A['Xmean'], A['Xstd'],
                               seq['init_params'], g['markers_rec'], seq['contact_lbl'])
    S = flip_sensitivity(o64)
    with default_f64(), KinkProbe(o64, tol_act=0.0) as pr:           # residual / speed kinks only (tol_act 0: no unit listed)
        o64.losses()
    l1 = sorted({f for e, kind, _, fr in pr.events if kind == 'l1' for f in fr})
    contact = any(kind.startswith('contact') for _, kind, _, _ in pr.events)
    print(f'\ncomputed kink exposure at golden (6): S median {max(float(v.median()) for v in S.values()):.1e} max '
          f'{max(float(v.max()) for v in S.values()):.1e}; frames with an L1 residual within 3e-6 m of zero: {l1}; contact speed near 0.1: {contact}')
    return dict(S=S, l1=l1, contact=contact)


def test_rot6d_vposer_golden(dev):
    from lemo_amd.rotation import convert_to_3D_all
    from lemo_amd.vposer import VPoser, make_vposer_weights
    r = np.load(os.path.join(GOLDEN, 'rot6d.npz'))
    aa = convert_to_3D_all(torch.from_numpy(r['rot6d_in']).to(dev))
    assert float((aa.cpu() - torch.from_numpy(r['rot6d_aa'])).abs().max()) < 5e-6
    g = np.load(os.path.join(GOLDEN, 'vposer_decode.npz'))
    vp = VPoser().eval()
    vp.load_state_dict({**vp.state_dict(), **{k: torch.from_numpy(v) for k, v in make_vposer_weights(2).items()}})
    vp = vp.to(dev)
    Z = torch.from_numpy(g['Z']).to(dev).requires_grad_(True)
    aa = vp.decode(Z, 'aa')
    assert rel_err(aa.detach().cpu(), g['aa']) < VERT_TOL
    assert rel_err(vp.decode(Z, 'matrot').detach().cpu(), g['matrot']) < VERT_TOL
    # backward: d(sum w aa)/dZ through the HIP rotation head + MLP vs autograd on the oracle's decode
    from oracle import lemo_oracle as O
    wgt = torch.randn(aa.shape, generator=torch.Generator().manual_seed(1))
    (aa * wgt.to(dev)).sum().backward()
    Zo = torch.from_numpy(g['Z']).clone().requires_grad_(True)
    (O.vposer_decode(O.make_vposer_weights(2), Zo, 'aa') * wgt).sum().backward()
    assert rel_err(Z.grad.cpu(), Zo.grad) < 1e-4
    Z2 = torch.from_numpy(g['Z']).to(dev).requires_grad_(True)
    wm = torch.randn(g['matrot'].shape, generator=torch.Generator().manual_seed(2))
    (vp.decode(Z2, 'matrot') * wm.to(dev)).sum().backward()
    Zo2 = torch.from_numpy(g['Z']).clone().requires_grad_(True)
    (O.vposer_decode(O.make_vposer_weights(2), Zo2, 'matrot') * wm).sum().backward()
    assert rel_err(Z2.grad.cpu(), Zo2.grad) < 1e-4


def test_smplx_module_golden(dev):
    from lemo_amd.body_model import create
    g = np.load(os.path.join(GOLDEN, 'lbs_small.npz'))
    m = synthetic.make_synthetic_smplx(seed=int(g['model_seed']), V=int(g['model_V']), F=1200)
    model = create(m, batch_size=4, extra_joint_ids=g['extra_ids'].tolist()).to(dev)
    p = {k: torch.from_numpy(g[k]).to(dev).requires_grad_(True) for k in ('betas', 'global_orient', 'body_pose', 'lh', 'rh', 'transl')}
    out = model(betas=p['betas'], global_orient=p['global_orient'], body_pose=p['body_pose'], left_hand_pose=p['lh'],
                right_hand_pose=p['rh'], transl=p['transl'], return_full_pose=True)
    assert rel_err(out.vertices.detach().cpu(), g['verts']) < VERT_TOL
    assert rel_err(out.joints.detach().cpu(), g['joints']) < VERT_TOL
    ((out.vertices * torch.from_numpy(g['wv']).to(dev)).sum() + (out.joints * torch.from_numpy(g['wj']).to(dev)).sum()).backward()
    for k in p:
        assert rel_err(p[k].grad.cpu(), g['g_' + k]) < 1e-4, k


@pytest.mark.parametrize('coherent', [False, True])
def test_smplx_module_full_size_vs_oracle(dev, coherent):
    """V=10475, B=119: every vertex and joint against the oracle (<= 1e-4 rel), on the i.i.d.-joint synthetic model and on the one
    with the licensed model's index locality (1 .. 4 skinning joints per vertex, lemo_amd.synthetic._coherent_skinning)."""
    from lemo_amd.body_model import create
    from oracle import lemo_oracle as O
    m = synthetic.make_synthetic_smplx(seed=0, coherent=coherent)
    seq = synthetic.make_synthetic_sequence(1, B=119)
    p = torch.from_numpy(seq['init_params'])
    gen = torch.Generator().manual_seed(0)
    body = torch.randn(119, 63, generator=gen) * 0.3
    so = O.SmplxOracle(m)
    with torch.no_grad():
        v_ref, j_ref, _ = so.forward(p[:, 6:16], p[:, 3:6], body, p[:, 48:60], p[:, 60:], p[:, 0:3])
    model = create(m, batch_size=119).to(dev)
    out = model(betas=p[:, 6:16].to(dev), global_orient=p[:, 3:6].to(dev), body_pose=body.to(dev),
                left_hand_pose=p[:, 48:60].to(dev), right_hand_pose=p[:, 60:].to(dev), transl=p[:, 0:3].to(dev))
    assert out.vertices.shape == (119, 10475, 3) and out.joints.shape == (119, 127, 3)
    assert rel_err(out.vertices.cpu(), v_ref) < VERT_TOL
    assert rel_err(out.joints.cpu(), j_ref) < VERT_TOL
    # blend GEMM variants: bf16 matrix cores with exactly split fp32 operands (default) vs fp32 MFMA -- the error
    # against the (fp32, CPU) oracle must be of the same size
    e_split = rel_err(out.vertices.cpu(), v_ref)
    try:
        model._device_body(dev).skin.blend_fp32 = 1                       # per-model constant (lemo_skin_const), no process-wide switch
        out0 = model(betas=p[:, 6:16].to(dev), global_orient=p[:, 3:6].to(dev), body_pose=body.to(dev),
                     left_hand_pose=p[:, 48:60].to(dev), right_hand_pose=p[:, 60:].to(dev), transl=p[:, 0:3].to(dev))
        e_f32 = rel_err(out0.vertices.cpu(), v_ref)
        e_ab = rel_err(out.vertices.cpu(), out0.vertices.cpu())
    finally:
        model._device_body(dev).skin.blend_fp32 = 0
    print(f'\nvertices vs oracle: split-bf16 blend GEMM {e_split:.3e}, fp32-MFMA blend GEMM {e_f32:.3e}, A vs B {e_ab:.3e}')
    assert e_split < 3 * e_f32 + 1e-6 and e_ab < 2e-6


def test_split_bf16_conv_error_is_fp32_sized(dev):
    """conv variant 3 multiplies exact fp32 operands as 3 bf16 pieces each (6 bf16-MFMA products, fp32 accumulate).
    Against a float64 convolution its error must be the size of an fp32 convolution's own rounding error: not
    larger than 1.5x the fp32-MFMA kernel's on the same data (measured: smaller), and < 2e-6 of max|out|."""
    import torch.nn.functional as F
    from lemo_amd import _hip
    from lemo_amd._hip import ptr
    from lemo_amd.priors import (cg8p_alloc, from_cg8p, to_cg8p, pack_conv3x3, pack_conv3x3_gmajor, pack_conv3x3_split,
                                 pack_conv3x3_bwd, pack_conv3x3_bwd_split)
    lib = _hip.get_lib()
    A = load_assets()
    H, W = 245, 134
    assert lib.conv3x3_split_supported(H, W, 64, 64) == 1
    w, b = np.asarray(A['enc_w']['enc_blc4.main.2.weight'], np.float32), np.asarray(A['enc_w']['enc_blc4.main.2.bias'], np.float32)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(64, H, W, generator=g).abs() * 0.3                       # post-LeakyReLU-like (mostly positive) input
    ref = F.leaky_relu(F.conv2d(x[None].double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1), 0.2)[0]
    t = lambda a: torch.from_numpy(a).to(dev)
    wt, wt2, w3, bd, xin = t(pack_conv3x3(w)), t(pack_conv3x3_gmajor(w)), t(pack_conv3x3_split(w).view(np.int16)), t(b), to_cg8p(x).to(dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    o2, o3 = cg8p_alloc(64, H, W, dev), cg8p_alloc(64, H, W, dev)
    lib.check(lib.conv3x3_mfma_lds(ptr(xin), ptr(wt), ptr(wt2), ptr(bd), None, ptr(o2), H, W, 64, 64, 0, s))
    lib.check(lib.conv3x3_mfma_split(ptr(xin), ptr(w3), ptr(wt), ptr(bd), None, ptr(o3), H, W, 64, 64, 0, s))
    torch.cuda.synchronize()
    e2 = float((from_cg8p(o2.cpu(), H, W).double() - ref).abs().max() / ref.abs().max())
    e3 = float((from_cg8p(o3.cpu(), H, W).double() - ref).abs().max() / ref.abs().max())
    r3 = float((from_cg8p(o3.cpu(), H, W).double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f'\nmax err / max|ref| vs float64: fp32-MFMA {e2:.3e}   split-bf16 {e3:.3e} (rms {r3:.3e})')
    assert e3 < 2e-6 and e3 <= 1.5 * e2 and r3 < 5e-7
    assert float(o3.reshape(8, H + 2, W + 2, 8)[:, 0].abs().max()) == 0.0          # zero border untouched
    # backward-data epilogue (x lrelu'(saved activation)), remainder pixels included (P % 128 = 62)
    dy, aux = torch.randn(64, H, W, generator=g), torch.randn(64, H, W, generator=g)
    refdx = F.conv_transpose2d(dy[None].double(), torch.from_numpy(w).double(), padding=1)[0] * torch.where(aux > 0, 1.0, 0.2).double()
    wb, wb3 = t(pack_conv3x3_bwd(w)), t(pack_conv3x3_bwd_split(w).view(np.int16))
    dyb, auxb, dxb = to_cg8p(dy).to(dev), to_cg8p(aux).to(dev), cg8p_alloc(64, H, W, dev)
    lib.check(lib.conv3x3_mfma_split(ptr(dyb), ptr(wb3), ptr(wb), None, ptr(auxb), ptr(dxb), H, W, 64, 64, 1, s))
    torch.cuda.synchronize()
    got = from_cg8p(dxb.cpu(), H, W).double()
    assert float((got - refdx).abs().max() / refdx.abs().max()) < 2e-6
    assert float((got[-1, -62:] - refdx[-1, -62:]).abs().max() / refdx.abs().max()) < 2e-6      # the remainder patches


def test_split_f16_conv_error_is_fp32_sized(dev):
    """conv variant 4 multiplies each fp32 operand as two error-compensated fp16 pieces (3 f16-MFMA products, fp32
    accumulate, per-workgroup power-of-two scaling).  Against a float64 convolution its error must be the size of an fp32
    convolution's own rounding error: <= 1.5x the fp32-MFMA kernel's on the same data and < 2e-6 of max|out| -- on
    activation-sized data AND on gradient-sized data (1e-6: deep in fp16's denormal range without the scaling)."""
    import torch.nn.functional as F
    from lemo_amd import _hip
    from lemo_amd._hip import ptr
    from lemo_amd.priors import (cg8p_alloc, from_cg8p, to_cg8p, pack_conv3x3, pack_conv3x3_gmajor, pack_conv3x3_split_f16,
                                 pack_conv3x3_bwd, pack_conv3x3_bwd_split_f16)
    lib = _hip.get_lib()
    A = load_assets()
    H, W = 245, 134
    w, b = np.asarray(A['enc_w']['enc_blc4.main.2.weight'], np.float32), np.asarray(A['enc_w']['enc_blc4.main.2.bias'], np.float32)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(64, H, W, generator=g).abs() * 0.3
    ref = F.leaky_relu(F.conv2d(x[None].double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1), 0.2)[0]
    t = lambda a: torch.from_numpy(a).to(dev)
    pf, fi = pack_conv3x3_split_f16(w)
    wt, wt2, w4, bd, xin = t(pack_conv3x3(w)), t(pack_conv3x3_gmajor(w)), t(pf.view(np.int16)), t(b), to_cg8p(x).to(dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    o2, o4 = cg8p_alloc(64, H, W, dev), cg8p_alloc(64, H, W, dev)
    lib.check(lib.conv3x3_mfma_lds(ptr(xin), ptr(wt), ptr(wt2), ptr(bd), None, ptr(o2), H, W, 64, 64, 0, s))
    lib.check(lib.conv3x3_mfma_split_f16(ptr(xin), ptr(w4), fi, ptr(wt), ptr(bd), None, ptr(o4), H, W, 64, 64, 0, s))
    torch.cuda.synchronize()
    e2 = float((from_cg8p(o2.cpu(), H, W).double() - ref).abs().max() / ref.abs().max())
    e4 = float((from_cg8p(o4.cpu(), H, W).double() - ref).abs().max() / ref.abs().max())
    r4 = float((from_cg8p(o4.cpu(), H, W).double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f'\nmax err / max|ref| vs float64: fp32-MFMA {e2:.3e}   split-f16 {e4:.3e} (rms {r4:.3e})')
    assert e4 < 2e-6 and e4 <= 1.5 * e2 and r4 < 6e-7
    assert float(o4.reshape(8, H + 2, W + 2, 8)[:, 0].abs().max()) == 0.0          # zero border untouched
    dy, aux = torch.randn(64, H, W, generator=g) * 1e-6, torch.randn(64, H, W, generator=g)
    refdx = F.conv_transpose2d(dy[None].double(), torch.from_numpy(w).double(), padding=1)[0] * torch.where(aux > 0, 1.0, 0.2).double()
    pb, bi = pack_conv3x3_bwd_split_f16(w)
    wb, wb4 = t(pack_conv3x3_bwd(w)), t(pb.view(np.int16))
    dyb, auxb, dxb = to_cg8p(dy).to(dev), to_cg8p(aux).to(dev), cg8p_alloc(64, H, W, dev)
    lib.check(lib.conv3x3_mfma_split_f16(ptr(dyb), ptr(wb4), bi, ptr(wb), None, ptr(auxb), ptr(dxb), H, W, 64, 64, 1, s))
    torch.cuda.synchronize()
    got = from_cg8p(dxb.cpu(), H, W).double()
    assert float((got - refdx).abs().max() / refdx.abs().max()) < 2e-6
    assert float((got[-1, -62:] - refdx[-1, -62:]).abs().max() / refdx.abs().max()) < 2e-6      # the remainder patches


DX0_GATE = 7e-6          # 3 x the largest measured value (round 5: 1.2e-6 .. 2.2e-6 across the seven families; was a silent 1e-4)


@pytest.mark.parametrize('variant', [0, 1, 2, 3, 4, 5, 7, 8, 9, 10])
def test_encoder_full_size_golden(dev, variant):
    """10-layer MFMA conv stack at 245x134 with the real runs/15217 weights: z, loss, input grad.  Variants 0-2 run
    every layer on the fp32 MFMA; 3 / 4 run the nine MFMA layers (forward and backward-data) on the split-bf16 / split-f16 kernel;
    5 / 6 run the SHIPPED chain (csrc/enc_chain.hpp): fused pairs (3,4) (5,6) (7,8) forward and (9,8) (7,6) (5,4) backward on the 8-wave /
    4-wave pair kernel, the other launches layer by layer on the split-f16 kernel (the default chain had no
    standalone z / loss / dx0 golden run)."""
    from lemo_amd import _hip
    from lemo_amd._hip import ptr
    from lemo_amd.priors import ENC_CHANNELS, EncWeights, cg8p_alloc, from_cg8p, _conv_layer
    lib = _hip.get_lib()
    A = load_assets()
    g = np.load(os.path.join(GOLDEN, 'enc_smooth.npz'))
    H, W = 245, 134
    enc = EncWeights(A['enc_w'], dev)
    x = torch.from_numpy(g['x'])[0, 0]
    x0 = torch.zeros(H + 2, W + 2); x0[1:-1, 1:-1] = x
    x0 = x0.to(dev).contiguous()
    act = [None] + [cg8p_alloc(ENC_CHANNELS[l], H, W, dev) for l in range(1, 11)]
    s = torch.cuda.current_stream(dev).cuda_stream
    lib.check(lib.conv3x3_c1(ptr(x0), ptr(enc.w[0]), ptr(enc.b[0]), ptr(act[1]), H, W, 32, s))
    pair = {5: lib.conv3x3_pair_f16, 7: lib.conv3x3_pair_f16, 8: lib.conv3x3_pair_f16, 9: lib.conv3x3_pair_f16}.get(variant)      # (7, 8, 9: + the fused tails below)
    # variant 8 differs from 7 only in its head launch, which starts from vertices (test_fused_marker_image_and_first_layer[8] holds its
    # act[1..3] against the separate launches); from an image, as here, its chain is 7's: the run pins that split_pack(l, bwd, 8) serves it
    P = (lambda l, bwd: enc.split_pack(l, bwd, variant)) if pair else None
    for l in range(1, 10):
        if pair and l in (3, 5, 7):
            (pa, ia), (pb, ib) = P(l, False), P(l + 1, False)
            lib.check(pair(ptr(act[l]), ptr(pa), ia, ptr(enc.b[l]), None, ptr(act[l + 1]), ptr(pb), ib, ptr(enc.b[l + 1]), None, ptr(act[l + 2]), H, W, 0, None, s))
        elif pair and l in (4, 6, 8):
            continue
        elif variant >= 2:
            _conv_layer(lib, enc, l, False, act[l], act[l + 1], None, H, W, variant, s)
        else:
            lib.check(lib.conv3x3_mfma(ptr(act[l]), ptr(enc.w[l]), ptr(enc.b[l]), None, ptr(act[l + 1]), H, W,
                                       ENC_CHANNELS[l], ENC_CHANNELS[l + 1], 0, variant, s))
    z = from_cg8p(act[10], H, W)
    assert abs(float(z.double().sum()) - float(g['z_sum'])) < 1e-5 * float(g['z_abs_sum'])
    assert rel_err(z[::8, ::16, ::16].cpu(), g['z_sub']) < 1e-5
    cnt = 64 * H * (W - 1)
    nb = lib.smooth_loss_blocks(H, W, 64)
    part, d0, d1 = torch.zeros(nb, device=dev), cg8p_alloc(64, H, W, dev), cg8p_alloc(64, H, W, dev)
    lib.check(lib.smooth_loss(ptr(act[10]), ptr(d0), ptr(part), H, W, 64, 2.0 / cnt, s))
    loss = float(part.double().sum() / cnt)
    assert abs(loss - float(g['loss_smooth'])) <= LOSS_TOL * float(g['loss_smooth'])
    cur = [d0, d1]
    ci = 0
    for l in range(9, {7: 1, 8: 1, 9: 2, 10: 2}.get(variant, 0), -1):
        if pair and l in (9, 7, 5):
            (pa, ia), (pb, ib) = P(l, True), P(l - 1, True)
            lib.check(pair(ptr(cur[ci]), ptr(pa), ia, None, ptr(act[l]), None, ptr(pb), ib, None, ptr(act[l - 1]), ptr(cur[1 - ci]), H, W, 1, None, s))
        elif pair and l in (8, 6, 4):
            continue
        elif variant >= 2:
            _conv_layer(lib, enc, l, True, cur[ci], cur[1 - ci], act[l], H, W, variant, s)
        else:
            lib.check(lib.conv3x3_mfma(ptr(cur[ci]), ptr(enc.wbwd[l]), None, ptr(act[l]), ptr(cur[1 - ci]), H, W,
                                       ENC_CHANNELS[l + 1], ENC_CHANNELS[l], 1, variant, s))
        ci = 1 - ci
    dx0 = torch.zeros(H * W, device=dev)
    if variant in (7, 8):        # layer 1 backward-data + layer 0 adjoint in one launch (csrc/conv_head_kernels.hip)
        pb, ib = P(1, True)
        lib.check(lib.enc_tail(ptr(cur[ci]), ptr(pb), ib, ptr(act[1]), ptr(enc.w[0]), ptr(dx0), H, W, s))
    elif variant in (9, 10):     # layer 2 and layer 1 backward-data + layer 0 adjoint in one launch (10: behind seven Winograd launches)
        P = P or (lambda l, bwd: enc.split_pack(l, bwd, variant))
        (p2, i2), (pb, ib) = P(2, True), P(1, True)
        lib.check(lib.enc_tail3(ptr(cur[ci]), ptr(p2), i2, ptr(act[2]), ptr(pb), ib, ptr(act[1]), ptr(enc.w[0]), ptr(dx0), H, W, s))
    else:
        lib.check(lib.conv3x3_c1_bwd(ptr(cur[ci]), ptr(enc.w[0]), ptr(dx0), H, W, 32, s))
    e_dx0 = rel_err(dx0.view(H, W).cpu(), g['gx'][0, 0])
    print(f'\nencoder chain variant {variant}: d loss / d image vs the reference-generated golden: {e_dx0:.2e} of max; loss rel '
          f'{abs(loss - float(g["loss_smooth"])) / float(g["loss_smooth"]):.1e}')
    # measured 1.2e-6 (split-f16 families, pairs included) .. 2.2e-6 (fp32-input MFMA) of the largest entry: gate at 3 x the largest
    assert e_dx0 < DX0_GATE, (variant, e_dx0)


def test_fit_small_vs_oracle_eager_and_graph(dev):
    import __graft_entry__ as ge
    from lemo_amd.fitting import AmassTemporalFitter
    prob = ge.small_problem()
    ofit, markers = ge.oracle_for(prob)
    total, parts, _, verts = ofit.losses()
    total.backward()
    fits = []
    for full in (True, False):
        fit = AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'],
                                  prob['Xstd'], prob['B'], dev, full_vertices=full)
        fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
        fit.forward(); fit.backward()
        torch.cuda.synchronize()
        L = fit.losses()
        for k in ('marker', 'vposer', 'shape', 'hand', 'contact', 'smooth'):
            assert abs(L[k] - float(parts[k])) <= LOSS_TOL * abs(float(parts[k])), (k, L[k], float(parts[k]))
        g = fit.grads_with_priors()
        for k, ref in (('transl', ofit.transl.grad), ('rot6d', ofit.rot6d.grad), ('other', ofit.other.grad)):
            assert rel_err(g[k].cpu(), ref) < 2e-4, k
        fits.append(fit)
    assert rel_err(fits[0].vertices().cpu(), verts.detach()) < VERT_TOL
    # Adam trajectories (eager launches and hipGraph replay) against the oracle.  One step must agree to fp32 rounding.
    # Later steps may not: the contact term selects `x[x > thr]` (opt_amass_temp.py:429-443), so a velocity within an
    # ulp of the threshold enters the mean on one side and not on the other, which changes a handful of gradient
    # entries by ~1e-3 relative and, through lr = 1e-2, parameters by ~1e-4 (measured: which kernel family trips a
    # flip is a matter of rounding pattern, tools/traj_check.py).  Hence: tight after 1 step, bounded after 5.
    s = torch.cuda.Stream(dev)
    for fit in fits:
        fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    with torch.cuda.stream(s):
        fits[0].step(1, use_graph=False)
        fits[1].step(1, use_graph=True)
    torch.cuda.synchronize()
    ofit.opt.zero_grad()
    ofit.step()
    for fit in fits:
        assert float((fit.params75().cpu() - ofit.params75()).abs().max()) < 2e-6
    with torch.cuda.stream(s):
        fits[0].step(4, use_graph=False)
        fits[1].step(4, use_graph=True)
    torch.cuda.synchronize()
    for _ in range(4):
        ofit.step()
    for fit in fits:
        d = (fit.params75().cpu() - ofit.params75()).abs()
        print(f'\nparams after 5 Adam steps vs oracle: max {float(d.max()):.2e} mean {float(d.mean()):.2e}')
        assert float(d.max()) < 1e-2 and float(d.mean()) < 2e-4          # no entry off by more than one lr-sized step
        assert int(fit.step_ctr.item()) == 5
        fit.forward()
        torch.cuda.synchronize()
        with torch.no_grad():
            ref_total = float(ofit.losses()[0])
        assert abs(fit.losses()['total'] - ref_total) < 2e-3 * ref_total


def test_side_full_forward_full_size(full_problem, dev):
    """lemo_fit_desc.verts_side (round 6) at BASELINE size, real second stream, graph replay: the engine whose all-vertex forward runs beside
    the per-frame launches must walk the set-U engine's trajectory bit for bit (same loss path), and after 1, 25 and 100 replayed iterations
    vertices() must be -- bit for bit, all 119 x 10475 rows -- the in-line all-vertex forward at the parameters the last iteration's forward
    saw, although Adam has rewritten the translation since and the next pose stage is what the side launch is joined in front of (a race
    with either would show here); ten repetitions of the 25-iteration graph give the same bits"""
    from lemo_amd.fitting import AmassTemporalFitter
    from lemo_amd.vposer import make_vposer_weights
    A, g, seq, model = full_problem['A'], full_problem['g'], full_problem['seq'], full_problem['model']
    mk = lambda full, side: AmassTemporalFitter(model, make_vposer_weights(2), A['enc_w'], A['ids'], A['Xmean'], A['Xstd'], 119, dev,
                                                full_vertices=full, side_full_forward=side)
    side, active, ref = mk(True, True), mk(False, False), mk(True, False)
    assert side.side_full and tuple(side.vertices().shape) == (119, 10475, 3)
    for f in (side, active):
        f.load_sequence(seq['init_params'], g['markers_rec'], seq['contact_lbl'])
    ref.load_sequence(seq['init_params'], g['markers_rec'], seq['contact_lbl'])
    s = torch.cuda.Stream(dev)
    B = 119

    def check(tag):
        torch.cuda.synchronize()
        assert torch.equal(side.params75(), active.params75()), tag
        ref.P['transl'].copy_(side.snap[:B * 3].view(B, 3)); ref.P['rot6d'].copy_(side.snap[B * 3:B * 9].view(B, 6))
        ref.P['other'].copy_(side.snap[B * 9:].view(B, 56))
        ref._after_write() if hasattr(ref, '_after_write') else None
        ref.forward()
        torch.cuda.synchronize()
        assert torch.equal(side.vertices(), ref.vertices()), (tag, float((side.vertices() - ref.vertices()).abs().max()))

    done = 0
    for n in (1, 24, 75):
        with torch.cuda.stream(s):
            side.step(n, use_graph=True); active.step(n, use_graph=True)
        done += n
        check(f'after {done} iterations')
    st = side.save_state()
    first = None
    for _ in range(10):
        side.load_state(st)
        with torch.cuda.stream(s):
            side.step(25, use_graph=True)
        torch.cuda.synchronize()
        v = side.vertices().clone()
        first = v if first is None else first
        assert torch.equal(v, first)


def test_wino_conv_full_size_vs_float64(dev):
    """conv variant 10 (csrc/conv_wino_kernels.hip) at the encoder's own size (245 x 134: 256 workgroups of 32 tiles + the direct last
    row of the odd H), real runs/15217 weights of layer 5: forward and backward-data against torch float64 on the host -- error of the
    size of an fp32 convolution's own rounding -- agreement with the split-f16 direct kernel (variant 4), zero border kept, ten launches
    bit-identical (fixed summation orders)"""
    from lemo_amd import _hip
    from lemo_amd._hip import ptr
    from lemo_amd.priors import EncWeights, cg8p_alloc, from_cg8p, to_cg8p, enc_layer_keys
    lib = _hip.get_lib()
    A = load_assets()
    enc = EncWeights(A['enc_w'], dev)
    keys = enc_layer_keys()
    H, W, l = 245, 134, 5
    g = torch.Generator().manual_seed(9)
    x = torch.randn(64, H, W, generator=g) * 0.3
    w, b = torch.from_numpy(A['enc_w'][keys[l] + '.weight']), torch.from_numpy(A['enc_w'][keys[l] + '.bias'])
    y64 = F.leaky_relu(F.conv2d(x[None].double(), w.double(), b.double(), padding=1), 0.2)[0]
    y32 = F.leaky_relu(F.conv2d(x[None], w, b, padding=1), 0.2)[0]
    s = torch.cuda.current_stream(dev).cuda_stream
    xin, out, ref4 = to_cg8p(x).to(dev), cg8p_alloc(64, H, W, dev), cg8p_alloc(64, H, W, dev)
    pu, iu = enc.split_pack(l, False, 10)
    p4, i4 = enc.split_pack(l, False, 4)
    assert lib.conv3x3_wino_supported(H, W, 64, 64) == 1
    assert lib.conv3x3_wino_f16(ptr(xin), ptr(pu), iu, ptr(enc.w[l]), ptr(enc.b[l]), None, ptr(out), H, W, 0, None, s) == 0
    assert lib.conv3x3_mfma_split_f16(ptr(xin), ptr(p4), i4, ptr(enc.w[l]), ptr(enc.b[l]), None, ptr(ref4), H, W, 64, 64, 0, s) == 0
    torch.cuda.synchronize()
    for _ in range(10):
        o2 = cg8p_alloc(64, H, W, dev)
        assert lib.conv3x3_wino_f16(ptr(xin), ptr(pu), iu, ptr(enc.w[l]), ptr(enc.b[l]), None, ptr(o2), H, W, 0, None, s) == 0
        torch.cuda.synchronize()
        assert torch.equal(o2, out)
    e, f = rel_err(from_cg8p(out.cpu(), H, W).double(), y64), rel_err(y32.double(), y64)
    print(f'\nWinograd layer forward vs float64: {e:.2e} (torch fp32 conv: {f:.2e}); vs the split-f16 direct kernel {rel_err(out.cpu(), ref4.cpu()):.2e}')
    assert e < 2e-6 and e < 3 * f + 2e-7 and rel_err(out.cpu(), ref4.cpu()) < 2e-6
    o = out.cpu().reshape(8, H + 2, W + 2, 8)
    assert float(o[:, 0].abs().max()) == 0 and float(o[:, -1].abs().max()) == 0 and float(o[:, :, 0].abs().max()) == 0 and float(o[:, :, -1].abs().max()) == 0
    d1 = torch.randn(64, H, W, generator=g) * 1e-6
    a0 = torch.randn(64, H, W, generator=g)
    refb = F.conv_transpose2d(d1[None].double(), w.double(), padding=1)[0] * torch.where(a0 > 0, 1.0, 0.2).double()
    fb = rel_err((F.conv_transpose2d(d1[None], w, padding=1)[0] * torch.where(a0 > 0, 1.0, 0.2)).double(), refb)
    pub, iub = enc.split_pack(l, True, 10)
    d0 = cg8p_alloc(64, H, W, dev)
    assert lib.conv3x3_wino_f16(ptr(to_cg8p(d1).to(dev)), ptr(pub), iub, ptr(enc.wbwd[l]), None, ptr(to_cg8p(a0).to(dev)), ptr(d0), H, W, 1, None, s) == 0
    torch.cuda.synchronize()
    eb = rel_err(from_cg8p(d0.cpu(), H, W).double(), refb)
    print(f'Winograd layer backward-data vs float64: {eb:.2e} (torch fp32: {fb:.2e})')
    assert eb < 2e-6 and eb < 3 * fb + 2e-7


@pytest.mark.parametrize('kernel', ['conv3x3_pair_f16'])
def test_fused_pair_conv_full_size_vs_float64(dev, kernel):
    """conv variant 5 (csrc/conv_pair_kernels.hip) and variant 6 (four-wave workgroups, csrc/conv_pair4_kernels.hip: same arithmetic and
    summation order, so the same bits -- run ten times: its two co-resident workgroups per CU are where a race would show) at the encoder's own size (245 x 134, real runs/15217 weights of layers 3 / 4):
    forward pair (intermediate AND output) and backward-data pair against torch float64 on the host; error of the size of an fp32
    convolution's own rounding, and agreement with two single-layer launches of the same arithmetic (variant 4)"""
    from lemo_amd import _hip
    from lemo_amd._hip import ptr
    from lemo_amd.priors import EncWeights, cg8p_alloc, from_cg8p, to_cg8p, enc_layer_keys
    lib = _hip.get_lib()
    A = load_assets()
    enc = EncWeights(A['enc_w'], dev)
    keys = enc_layer_keys()
    H, W = 245, 134
    g = torch.Generator().manual_seed(7)
    x = torch.randn(64, H, W, generator=g) * 0.3
    w1, b1 = torch.from_numpy(A['enc_w'][keys[3] + '.weight']), torch.from_numpy(A['enc_w'][keys[3] + '.bias'])
    w2, b2 = torch.from_numpy(A['enc_w'][keys[4] + '.weight']), torch.from_numpy(A['enc_w'][keys[4] + '.bias'])
    a1_64 = F.leaky_relu(F.conv2d(x[None].double(), w1.double(), b1.double(), padding=1), 0.2)
    a2_64 = F.leaky_relu(F.conv2d(a1_64, w2.double(), b2.double(), padding=1), 0.2)[0]
    a1_32 = F.leaky_relu(F.conv2d(x[None], w1, b1, padding=1), 0.2)
    a2_32 = F.leaky_relu(F.conv2d(a1_32, w2, b2, padding=1), 0.2)[0]
    s = torch.cuda.current_stream(dev).cuda_stream
    xin, mid, out = to_cg8p(x).to(dev), cg8p_alloc(64, H, W, dev), cg8p_alloc(64, H, W, dev)
    pa, ia = enc.split_pack(3, False, 5)
    pb, ib = enc.split_pack(4, False, 5)
    pair = getattr(lib, kernel)
    assert pair(ptr(xin), ptr(pa), ia, ptr(enc.b[3]), None, ptr(mid), ptr(pb), ib, ptr(enc.b[4]), None, ptr(out), H, W, 0, None, s) == 0
    torch.cuda.synchronize()
    e_mid, e_out = rel_err(from_cg8p(mid.cpu(), H, W).double(), a1_64[0]), rel_err(from_cg8p(out.cpu(), H, W).double(), a2_64)
    f_mid, f_out = rel_err(a1_32[0].double(), a1_64[0]), rel_err(a2_32.double(), a2_64)
    print(f'\nfused pair forward vs float64: mid {e_mid:.2e} out {e_out:.2e} (torch fp32 conv: {f_mid:.2e} / {f_out:.2e})')
    assert e_mid < 2e-6 and e_out < 2e-6 and e_out < 3 * f_out + 2e-7
    o = out.cpu().reshape(8, H + 2, W + 2, 8)
    assert float(o[:, 0].abs().max()) == 0 and float(o[:, -1].abs().max()) == 0 and float(o[:, :, 0].abs().max()) == 0 and float(o[:, :, -1].abs().max()) == 0
    s_mid, s_out = cg8p_alloc(64, H, W, dev), cg8p_alloc(64, H, W, dev)
    assert lib.conv3x3_mfma_split_f16(ptr(xin), ptr(pa), ia, ptr(enc.w[3]), ptr(enc.b[3]), None, ptr(s_mid), H, W, 64, 64, 0, s) == 0
    assert lib.conv3x3_mfma_split_f16(ptr(s_mid), ptr(pb), ib, ptr(enc.w[4]), ptr(enc.b[4]), None, ptr(s_out), H, W, 64, 64, 0, s) == 0
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), s_out.cpu()) < 1e-6 and rel_err(mid.cpu(), s_mid.cpu()) < 1e-6
    # backward-data pair through layers (4, 3)
    d2, a0 = torch.randn(64, H, W, generator=g) * 1e-6, torch.randn(64, H, W, generator=g)
    # reference chain in float64 with the lrelu' branches taken from the SAME saved activations the kernel reads (a unit of the 2.1 M
    # that sits within rounding of its kink would otherwise be a coin toss between the two sides)
    a1_dev = from_cg8p(mid.cpu(), H, W)
    dmid = F.conv_transpose2d(d2[None].double(), w2.double(), padding=1)[0] * torch.where(a1_dev > 0, 1.0, 0.2).double()
    ref = F.conv_transpose2d(dmid[None], w1.double(), padding=1)[0] * torch.where(a0 > 0, 1.0, 0.2).double()
    qa, ja = enc.split_pack(4, True, 5)
    qb, jb = enc.split_pack(3, True, 5)
    d0 = cg8p_alloc(64, H, W, dev)
    assert pair(ptr(to_cg8p(d2).to(dev)), ptr(qa), ja, None, ptr(mid), None, ptr(qb), jb, None, ptr(to_cg8p(a0).to(dev)), ptr(d0), H, W, 1, None, s) == 0
    torch.cuda.synchronize()
    e_b = rel_err(from_cg8p(d0.cpu(), H, W).double(), ref)
    print(f'fused pair backward-data vs float64: {e_b:.2e}')
    assert e_b < 2e-6


@pytest.mark.parametrize('conv_variant', [5, 7, 8, 9, 10, 4, 3, 2])
def test_fit_full_size_golden(full_problem, kink_exposure, dev, conv_variant):
    """golden (6): B=119, V=10475, real encoder weights: six losses + total, verts, grads, params after
    1 and 10 Adam steps (graph replay); with the default split-bf16 encoder kernels (3) and the fp32-MFMA ones (2)."""
    g, seq = full_problem['g'], full_problem['seq']
    fit = full_problem['make'](True, conv_variant)
    assert fit.conv_variant == conv_variant
    fit.load_sequence(seq['init_params'], g['markers_rec'], seq['contact_lbl'])
    fit.forward(); fit.backward()
    torch.cuda.synchronize()
    L = fit.losses()
    for k in ('marker', 'vposer', 'shape', 'hand', 'contact', 'smooth'):
        ref = float(g['loss_' + k])
        assert abs(L[k] - ref) <= LOSS_TOL * abs(ref), (k, L[k], ref)
    assert abs(L['total'] - float(g['total'])) <= LOSS_TOL * float(g['total'])
    v = fit.vertices()
    assert rel_err(v[:, ::97].cpu(), g['verts_sub']) < VERT_TOL
    assert abs(float(v.double().sum()) - float(g['verts_sum'])) < 1e-6 * float(v.double().abs().sum())
    assert rel_err(fit.params72().cpu(), g['p72_0']) < 1e-5
    gr = fit.grads_with_priors()
    # vs the reference's fp32 CPU result, frame by frame, against a COMPUTED bound: both sides are fp32-accurate evaluations of
    # an objective with kinks, so frame f may differ by  2e-5 (rounding, of the group's largest entry) + 2 x 2 x S[f],
    # S = how far f's gradient moves when every LeakyReLU unit of the encoder within 3e-6 x layer-max of its kink takes the
    # other branch (float64, oracle/f64.py::flip_sensitivity; one factor 2 per side).  Frames holding an L1 residual within
    # 3e-6 m of zero (or any frame, if a contact speed sits at the 0.1 m/s threshold) get the constant 2e-3 instead.  The
    # median over frames of the per-frame maximum -- the arithmetic where nothing flipped -- stays at 2e-5.
    KE = kink_exposure
    for k in ('transl', 'rot6d', 'other'):
        a, b = gr[k].cpu().double(), torch.from_numpy(g['g_' + k]).double()
        e = ((a - b).abs() / b.abs().max()).max(1).values
        bound = 2e-5 + 4.0 * KE['S'][k]
        if KE['contact']:
            bound = torch.full_like(bound, 2e-3)
        for f in KE['l1']:
            bound[f] = max(float(bound[f]), 2e-3)
        bad = (e > bound).nonzero().flatten().tolist()
        assert not bad, (k, bad, [float(e[i]) for i in bad], [float(bound[i]) for i in bad])
        assert float(e.median()) < 2e-5, (k, float(e.median()))
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        fit.step(1, use_graph=True)
    torch.cuda.synchronize()
    # first Adam step moves every entry by ~lr*g/(|g|+eps): entries with |g| ~ eps amplify fp noise in g
    assert float((fit.params75().cpu() - torch.from_numpy(g['p75_after1'])).abs().max()) < 3e-4
    assert float((fit.params75().cpu() - torch.from_numpy(g['p75_after1'])).abs().mean()) < 2e-6
    with torch.cuda.stream(s):
        fit.step(9, use_graph=True)
    torch.cuda.synchronize()
    # (round 4) steps beyond the first are asserted ONE AT A TIME from the reference's own optimiser states -- iterations 0, 1, 10,
    # 30, 60, 61 (lr switch), 62, 99 of the 100-step loop: tests/test_gpu_teacher.py::test_amass_loop_teacher_forced_full_size, next
    # state within 5e-3 x lr of the reference's on every regular entry (measured 1.7e-3), Adam moments bit for bit.  The free-running
    # 10-step distance is printed; its old gates (max < 1e-2, mean < 1e-4) bounded nothing.  What stays
    # asserted here: the loss of the 10th iteration (the trajectory as a whole descends the same valley).
    d10 = (fit.params75().cpu() - torch.from_numpy(g['p75_after10'])).abs()
    print(f'\nconv variant {conv_variant}: free-running 10 steps vs the fixture: max {float(d10.max()):.2e} mean {float(d10.mean()):.2e}')
    assert abs(fit.losses()['total'] - float(g['total_hist'][9])) < 1e-3 * float(g['total_hist'][9])


@pytest.mark.parametrize('conv_variant', [5, 7, 8, 9])
def test_fused_marker_image_and_first_layer(full_problem, dev, conv_variant):
    """the engine's one-launch marker image + first encoder layer (marker_c1_kernel; variant 7: enc_head_kernel, which also carries layer 1)
    against the stand-alone layer (C-ABI lemo_conv3x3_c1) applied to the image it published: identical activations, at the full
    245 x 134 size; variant 7's act[2] against the single-layer kernel on that act[1]."""
    from lemo_amd._hip import ptr
    from lemo_amd.priors import ENC_CHANNELS
    g, seq = full_problem['g'], full_problem['seq']
    fit = full_problem['make'](True, conv_variant)
    fit.load_sequence(seq['init_params'], g['markers_rec'], seq['contact_lbl'])
    fit.forward()
    ref1 = torch.zeros_like(fit.act[1])
    lib = fit.lib
    lib.check(lib.conv3x3_c1(ptr(fit.ws['x0']), ptr(fit.enc.w[0]), ptr(fit.enc.b[0]), ptr(ref1), fit.H, fit.W, ENC_CHANNELS[1],
                             lib.stream(dev)))
    torch.cuda.synchronize()
    assert float(fit.ws['x0'].abs().max()) > 0
    assert torch.equal(ref1, fit.act[1])
    if conv_variant >= 7:
        from lemo_amd.priors import _conv_layer
        ref2 = torch.zeros_like(fit.act[2])
        _conv_layer(lib, fit.enc, 1, False, fit.act[1], ref2, None, fit.H, fit.W, 4, lib.stream(dev))
        torch.cuda.synchronize()
        assert rel_err(fit.act[2].cpu(), ref2.cpu()) < 1e-6
    if conv_variant >= 8:                                    # ... and layer 2 (32 -> 64) against the single-layer kernel on that act[2]
        ref3 = torch.zeros_like(fit.act[3])
        _conv_layer(lib, fit.enc, 2, False, fit.act[2], ref3, None, fit.H, fit.W, 4, lib.stream(dev))
        torch.cuda.synchronize()
        assert rel_err(fit.act[3].cpu(), ref3.cpu()) < 1e-6


def test_compat_lbs_function(dev):
    """smplx.lbs.lbs served by lemo_amd.compat on the GPU library: vertices, joints and gradients vs the oracle's lbs
    (pinned to the reference's lbs.py), SMPL-X-sized model (V = 10475), 16 shape coefficients."""
    from lemo_amd.compat.smplx.lbs import lbs
    from oracle import lemo_oracle as O
    m = synthetic.make_synthetic_smplx(seed=0)
    so = O.SmplxOracle(m, extra_joint_ids=[0])
    sd = torch.cat([so.shapedirs, so.expr_dirs], dim=-1)[:, :, :16].contiguous()
    a = dict(v_template=so.v_template, shapedirs=sd, posedirs=so.posedirs, J_regressor=so.J_regressor,
             parents=so.parents, lbs_weights=so.lbs_weights)
    g = torch.Generator().manual_seed(5)
    B = 4
    betas, pose = torch.randn(B, 16, generator=g) * 0.5, torch.randn(B, 165, generator=g) * 0.2
    bc, pc = betas.clone().requires_grad_(True), pose.clone().requires_grad_(True)
    v_ref, j_ref = O.lbs(bc, pc, **a)
    wv = torch.randn(v_ref.shape, generator=g)
    (v_ref * wv).sum().backward()
    bd, pd = betas.to(dev).requires_grad_(True), pose.to(dev).requires_grad_(True)
    ad = {k: v.to(dev) for k, v in a.items()}
    verts, joints = lbs(bd, pd, ad['v_template'], ad['shapedirs'], ad['posedirs'], ad['J_regressor'], ad['parents'],
                        ad['lbs_weights'])
    assert rel_err(verts.detach().cpu(), v_ref.detach()) < 1e-4 and rel_err(joints.detach().cpu(), j_ref.detach()) < 1e-4
    (verts * wv.to(dev)).sum().backward()
    assert rel_err(bd.grad.cpu(), bc.grad) < 1e-4 and rel_err(pd.grad.cpu(), pc.grad) < 1e-4


def test_active_vertex_forward_is_identical(full_problem, dev):
    """forwarding only the 253 vertices the losses read gives the same losses and gradients."""
    from lemo_amd.priors import from_cg8p
    g, seq = full_problem['g'], full_problem['seq']
    res = []
    for full in (True, False):
        fit = full_problem['make'](full)
        fit.load_sequence(seq['init_params'], g['markers_rec'], seq['contact_lbl'])
        fit.forward(); fit.backward()
        torch.cuda.synchronize()
        acts = [from_cg8p(fit.act[l], fit.H, fit.W).cpu() for l in range(1, 11)]
        res.append((fit.losses(), {k: v.clone() for k, v in fit.grads().items()}, acts))
    for k in res[0][0]:
        assert abs(res[0][0][k] - res[1][0][k]) <= 1e-6 * abs(res[0][0][k]) + 1e-12, k
    # The two forwards differ in the last bits (blend GEMM of 10475 vs 253 rows), so a LeakyReLU unit of the encoder's 21 M that sits within
    # rounding of its kink can take different branches in the two runs (round 5: with conv variant 7 one unit of layer 7 does on this clip
    # -- the very unit float64 disagrees with the engine about, profiles/r05_gates.txt seed 0, frame 35, 5.6e-4).  The frames such a unit
    # reaches are excluded -- after checking that the unit IS at its kink -- and every other frame must agree to 5e-5 (the orientation
    # gradient is a sum over all vertices with cancellation and carries the last-bit differences at ~2e-5 of its max).
    import kink_attribution as KA
    B = res[0][1]['transl'].shape[0]
    skip = set()
    for l, (a, b) in enumerate(zip(res[0][2], res[1][2]), start=1):
        m = float(a.abs().max())
        for c, y, x in ((a > 0) != (b > 0)).nonzero().tolist():
            assert abs(float(a[c, y, x])) <= 1e-5 * m and abs(float(b[c, y, x])) <= 1e-5 * m, (l, c, y, x, float(a[c, y, x]), float(b[c, y, x]))
            skip.update(KA.frames_in_reach(l, x, B))
    assert len(skip) <= B // 3, sorted(skip)
    keep = torch.tensor([f not in skip for f in range(B)])
    print(f'\nactive vs all-vertex forward: {len(skip)} frames within reach of a unit at its kink excluded')
    for k in res[0][1]:
        ref = res[0][1][k]
        assert float((res[1][1][k][keep] - ref[keep]).abs().max() / ref.abs().max()) < 5e-5, k


def test_translation_invariance_property(full_problem, dev):
    """size-independent property at full size: moving the whole clip and its target markers by a
    constant leaves every loss term unchanged (the smoothness feature is canonicalised to marker 0 of
    frame 0, opt_amass_temp.py:376-377)."""
    g, seq = full_problem['g'], full_problem['seq']
    fit = full_problem['make'](True)
    out = []
    for off in (np.zeros(3, np.float32), np.array([0.5, -0.25, 0.125], np.float32)):
        p = seq['init_params'].copy(); p[:, 0:3] += off
        fit.load_sequence(p, g['markers_rec'] + off, seq['contact_lbl'])
        fit.forward()
        torch.cuda.synchronize()
        out.append(fit.losses())
    for k in ('marker', 'contact', 'smooth'):
        assert abs(out[0][k] - out[1][k]) <= 2e-4 * abs(out[0][k]), (k, out[0][k], out[1][k])


@pytest.mark.timeout(900)
def test_mpjpe_after_full_fit(full_problem, dev):
    """BASELINE metric 'MPJPE vs ref': 100-step fit on the GPU vs the oracle from identical inputs;
    mean over frames and the first 22 joints of ||J_gpu - J_oracle||, in mm."""
    from oracle import lemo_oracle as O
    g, seq, A = full_problem['g'], full_problem['seq'], full_problem['A']
    steps = 100
    fit = full_problem['make'](True)
    fit.load_sequence(seq['init_params'], g['markers_rec'], seq['contact_lbl'])
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        fit.step(steps, use_graph=True)
        fit.forward()
    torch.cuda.synchronize()
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    so = O.SmplxOracle(full_problem['model'])
    ofit = O.AmassFitOracle(so, O.make_vposer_weights(2), A['enc_w_torch'], A['ids'], A['Xmean'], A['Xstd'],
                            seq['init_params'], g['markers_rec'], seq['contact_lbl'], faithful=False)
    first = ofit.step()
    for _ in range(steps - 1):
        last = ofit.step()
    with torch.no_grad():
        p72 = O.convert_to_3D_rot(ofit.params75())
        _, j_ref, _ = ofit._body(p72)
    mpjpe = O.mpjpe_mm(fit.posed_joints().cpu(), j_ref[:, :55])
    Lg = fit.losses()['total']
    print(f'MPJPE gpu-vs-oracle after {steps} steps: {mpjpe:.4f} mm ; total loss {first["total"]:.4f} -> oracle {last["total"]:.4f} / gpu {Lg:.4f}')
    assert Lg < 0.7 * first['total']                    # the fit actually descends
    assert mpjpe < 2.0, mpjpe


@pytest.mark.parametrize('stage,first,coherent', [('S3', False, False), ('S2', True, False), ('S3', False, True)])
def test_prox_iteration_vs_oracle(dev, stage, first, coherent):
    """PROX twin (a11/a12/a14): 14 loss_dict entries, gradients, first-window erase, 3 Adam steps (the third case: on the synthetic
    model with the licensed model's index locality -- chunks of the all-vertex LBS backward that touch few joints)."""
    import __graft_entry__ as ge
    from lemo_amd.prox import LOSS_KEYS
    prob = ge.prox_small_problem(stage=stage, coherent=coherent)
    of = ge.prox_oracle_for(prob, first_batch_flag=first)
    old = of.closure()
    fit, bm = ge.prox_fitter_for(prob, dev, first_batch_flag=first)
    ld = fit.closure()
    for k in LOSS_KEYS:
        a, b = float(ld[k]), float(old[k])
        assert abs(a - b) <= LOSS_TOL * abs(b) + 1e-12, (k, a, b)
    assert float(old['sdf_penetration_loss']) > 0 and float(old['loss_fric_tangent']) > 0
    for a, b in [(fit.pose_embedding.grad, of.pose_embedding.grad)] + \
                [(getattr(bm, n).grad, of.p[n].grad) for n in ('transl', 'global_orient', 'left_hand_pose', 'expression')]:
        assert rel_err(a.cpu(), b) < 2e-4
    n_erase = int(prob['B'] * 0.15)
    assert (float(fit.pose_embedding.grad[:n_erase].abs().max()) > 0) == first
    for _ in range(3):
        o = of.step()
        l = fit.step()
    assert abs(float(l['total_loss']) - o['total_loss']) <= 1e-4 * abs(o['total_loss'])
    assert float((fit.pose_embedding.detach().cpu() - of.pose_embedding.detach()).abs().max()) < 1e-4


def test_prox_full_size_window_runs(dev):
    """B = 100 window, V = 10475, 256^3 SDF (BASELINE configs 4-5 shape): finite losses, loss decreases;
    prints the iteration rate of the module-level (autograd) PROX path."""
    import time
    import __graft_entry__ as ge
    from lemo_amd.assets import load_assets
    from lemo_amd.prox import S3_WEIGHTS, load_prox_tables
    A = load_assets()
    B = 100
    small = ge.prox_small_problem(B=B, stage='S3')
    rng = np.random.default_rng(3)
    D = 256
    zz = np.linspace(-3, 6, D, dtype=np.float32)
    prob = dict(small, model=synthetic.make_synthetic_smplx(seed=0), V=10475, ids=A['ids'], Xmean=A['Xmean'], Xstd=A['Xstd'],
                fric_ids=load_prox_tables()['contact_fric_verts_ids'],
                sdf=(np.broadcast_to(zz[None, None, :], (D, D, D)) - 1.40).astype(np.float32).copy(), weights=S3_WEIGHTS)
    mask = np.ones((B, 67), np.float32); mask[40:60, :22] = 0
    prob['infill'] = dict(marker_mask=mask, body_markers_rec=(rng.standard_normal((B - 1, 67, 3)) * 0.3).astype(np.float32),
                          contact_lbl_rec=(rng.random((B - 1, 4)) < 0.7).astype(np.float32))
    fit, bm = ge.prox_fitter_for(prob, dev, first_batch_flag=False)
    l0 = float(fit.step(1)['total_loss'])
    torch.cuda.synchronize()
    t0 = time.time()
    n = 20
    ld = fit.step(n, use_graph=False)
    torch.cuda.synchronize()
    dt = time.time() - t0
    l1 = float(ld['total_loss'])
    print(f'PROX S3 window B=100 V=10475, eager launches: {n / dt:.1f} iterations/s ({dt / n * 1e3:.2f} ms/iteration); total {l0:.3f} -> {l1:.3f}')
    assert np.isfinite(l1) and l1 < l0
    # captured iteration replayed: same arithmetic, the host out of the loop
    fit_g, _ = ge.prox_fitter_for(prob, dev, first_batch_flag=False)
    fit_e, _ = ge.prox_fitter_for(prob, dev, first_batch_flag=False)
    lg, le = fit_g.step(8, use_graph=True), fit_e.step(8, use_graph=False)
    torch.cuda.synchronize()
    # (not bit-for-bit: torch's scatter / grid_sample backward ops of this MODULE path accumulate with atomics, so even two
    # eager runs differ in the last bits -- the printed 20-step loss above varies in its 6th digit from run to run -- and the
    # loss has thresholds (contact, friction, sdf < 0.01) that turn a last-bit difference into an Adam-step-sized one in a few
    # entries now and then: one of six full-suite runs of round 2 tripped the former max < 5e-3 bound.  Bounds: the mean, and
    # the drift two Adam trajectories can build up in 8 steps of lr 0.005 (they may move in opposite directions).  The NATIVE engine is deterministic and checked bit for bit, tests/test_gpu_r2.py.)
    dpe = (fit_g.pose_embedding.detach() - fit_e.pose_embedding.detach()).abs()
    # against the spread of TWO EAGER runs (the former max bound, 0.1, was vacuous): the graph run may not be further
    # from an eager run than 3x what a second eager run is, in the mean and in the 99 % quantile; a capture bug in a few frames
    # (3200 entries, 32 per frame) shows in the quantile long before it shows in the mean
    fit_e2, _ = ge.prox_fitter_for(prob, dev, first_batch_flag=False)
    le2 = fit_e2.step(8, use_graph=False)
    torch.cuda.synchronize()
    dee = (fit_e2.pose_embedding.detach() - fit_e.pose_embedding.detach()).abs()
    q = lambda t: float(torch.quantile(t.flatten().float(), 0.99))
    print(f'module path, 8 steps: |graph - eager| mean {float(dpe.mean()):.2e} q99 {q(dpe):.2e} max {float(dpe.max()):.2e}; '
          f'|eager - eager| mean {float(dee.mean()):.2e} q99 {q(dee):.2e} max {float(dee.max()):.2e}')
    assert float(dpe.mean()) <= 3 * float(dee.mean()) + 2e-5 and q(dpe) <= 3 * q(dee) + 2e-4
    assert float(dpe.max()) <= 2 * 8 * 0.005 * 1.25              # (what two Adam trajectories can drift apart at all)
    spread = abs(float(le2['total_loss']) - float(le['total_loss']))
    assert abs(float(lg['total_loss']) - float(le['total_loss'])) <= 3 * spread + 1e-3 * abs(float(le['total_loss']))
    n = 103
    torch.cuda.synchronize()
    t0 = time.time()
    ld = fit_g.step(n, use_graph=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f'PROX S3 window, iteration captured once and replayed: {n / dt:.1f} iterations/s ({dt / n * 1e3:.2f} ms/iteration incl. capture)')
    assert np.isfinite(float(ld['total_loss']))


def test_infill_ae_full_size_golden_and_finetune(dev):
    """a13 / N1: infilling AE at the reference shape [1,4,210,135] (z [1,256,7,5], AE.py:93-108): golden forward,
    every parameter gradient vs the oracle, and the timing of the 60-step finetune + eval of one clip."""
    import time
    from lemo_amd.infill import AE, finetune_and_infill
    from oracle import lemo_oracle as O
    g = np.load(os.path.join(GOLDEN, 'ae_infill.npz'))
    w = {k: torch.from_numpy(v) for k, v in synthetic.make_ae_weights(7).items()}
    ae = AE().to(dev)
    ae.load_state_dict(w)
    x = torch.from_numpy(g['x'].astype(np.float32)).to(dev)
    out, z = ae(x)
    assert out.shape == (1, 1, 210, 135) and z.shape == (1, 256, 7, 5)
    assert rel_err(z.detach().cpu(), g['z']) < 1e-4 and rel_err(out.detach().cpu()[0, 0, ::7, ::5], g['out_sub']) < 1e-4
    assert abs(float(out.double().sum()) - float(g['out_sum'])) < 1e-4 * float(out.double().abs().sum())
    # gradients w.r.t. all 40 parameter tensors.  Max-pool argmax and the LeakyReLU mask are discontinuous: at
    # 28k pixels x 20 layers fp32 noise between the CPU and the MFMA summation order flips a few near-ties, which
    # moves gradient mass between neighbouring pixels (bias gradients stay exact, weight gradients move ~1e-4..1e-3;
    # tools/ae_debug.py, tools/wgrad_debug.py: each kernel alone is exact to 1e-6 at these sizes).  So: norm-wise
    # tolerance at full size, element-wise 1e-4 at 64x40 where no flip occurs.
    gen = torch.Generator().manual_seed(0)
    mask = (torch.rand(210, 135, generator=gen) > 0.3)
    wo = torch.randn(210, 135, generator=gen)
    (out[0, 0] * wo.to(dev)).sum().backward()
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    xc = x.cpu()
    oo, _ = O.ae_forward(wr, xc)
    (oo[0, 0] * wo).sum().backward()
    for k, p in ae.named_parameters():
        a, b = p.grad.cpu().double(), wr[k].grad.double()
        assert float((a - b).norm() / b.norm()) < 2e-2, k
        if k.startswith('dec_blc'):
            assert rel_err(a, b) < 1e-4, k
    ae2 = AE().to(dev)
    ae2.load_state_dict(w)
    xs = torch.randn(1, 4, 64, 40, generator=gen)
    ws = torch.randn(64, 40, generator=gen)
    (ae2(xs.to(dev))[0][0, 0] * ws.to(dev)).sum().backward()
    wr2 = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    (O.ae_forward(wr2, xs)[0][0, 0] * ws).sum().backward()
    for k, p in ae2.named_parameters():
        assert rel_err(p.grad.cpu(), wr2[k].grad) < 1e-4, k
    wdev0 = {k: v.to(dev) for k, v in w.items()}
    finetune_and_infill(ae, wdev0, x, mask.to(dev), steps=8, lr=3e-6)       # one-time: gather tables, LDS opt-ins, allocator
    torch.cuda.synchronize()
    t0 = time.time()
    rec, zz = finetune_and_infill(ae, wdev0, x, mask.to(dev), steps=60, lr=3e-6)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f'infilling AE (training step captured in a graph): 60 finetune steps + eval at [1,4,210,135]: {dt * 1e3:.1f} ms per clip ({dt / 61 * 1e3:.2f} ms per pass)')
    assert rec.shape == (1, 1, 208, 119) and torch.isfinite(rec).all()
    # graph replay == eager launches: same kernels in the same order on the same data (12 steps each way)
    wdev = {k: v.to(dev) for k, v in w.items()}
    r_g, _ = finetune_and_infill(ae, wdev, x, mask.to(dev), steps=12, lr=3e-6, use_graph=True)
    p_g = {k: v.detach().clone() for k, v in ae.state_dict().items()}
    torch.cuda.synchronize()
    t0 = time.time()
    r_e, _ = finetune_and_infill(ae, wdev, x, mask.to(dev), steps=12, lr=3e-6, use_graph=False)
    torch.cuda.synchronize()
    print(f'eager launches: {(time.time() - t0) / 13 * 1e3:.2f} ms per pass')
    assert torch.equal(r_g, r_e)
    for k, v in ae.state_dict().items():
        assert torch.equal(v, p_g[k]), k
    moved = max(float((p_g[k] - wdev[k]).abs().max()) for k in p_g)
    assert 1e-6 < moved < 1e-3                                    # 12 Adam steps of lr 3e-6 actually happened


def test_marker_image_encode_decode_golden(dev):
    """SURVEY N2 on the GPU: utils/utils.py::get_local_markers_4chan / reconstruct_global_body against the golden
    vectors produced by the reference's own functions; device tensors in -> device tensors out."""
    from lemo_amd.markers import get_local_markers_4chan, reconstruct_global_body
    g = np.load(os.path.join(GOLDEN, 'markers_decode.npz'))
    img, piv = get_local_markers_4chan(torch.from_numpy(g['body']).to(dev), torch.from_numpy(g['contact']).to(dev))
    assert img.is_cuda and img.shape == (4, 119, 208) and piv.dtype == torch.float64
    assert rel_err(img.cpu(), g['image']) < 1e-5
    assert abs(float(piv[0]) - float(g['rot_0_pivot'][0])) < 1e-6
    glob = reconstruct_global_body(torch.from_numpy(g['decode_in']).to(dev), piv)
    assert rel_err(glob.cpu(), g['global_body']) < 1e-5
    img_np, _ = get_local_markers_4chan(g['body'], g['contact'])          # numpy in -> float64 numpy out, like the reference
    assert isinstance(img_np, np.ndarray) and img_np.dtype == np.float64
