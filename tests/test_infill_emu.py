"""CPU suite: infilling autoencoder (models/AE.py) forward, all 40 parameter gradients and the finetune
loop (opt_amass_temp.py:160-215) through the HIP kernels on the host emulator vs the oracle."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from lemo_amd import synthetic
from lemo_amd._hip import ptr
from oracle import lemo_oracle as O


def _weights():
    return {k: torch.from_numpy(v) for k, v in synthetic.make_ae_weights(7).items()}


@pytest.mark.timeout(600)
def test_ae_forward_and_all_parameter_gradients(emu_lib):
    from lemo_amd.infill import AE
    w = _weights()
    ae = AE(downsample=True, in_channel=4, kernel=3, _lib=emu_lib)
    assert set(ae.state_dict().keys()) == set(w.keys())
    ae.load_state_dict(w)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 34, 21, generator=g)                       # odd sizes: 34x21 -> 17x11 -> 9x6 -> 5x3 -> 3x2 -> 2x1
    out, z = ae(x)
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    oo, zo = O.ae_forward(wr, x)
    assert out.shape == (1, 1, 34, 21) and z.shape == (1, 256, 2, 1)
    assert rel_err(out.detach(), oo.detach()) < 1e-5 and rel_err(z.detach(), zo.detach()) < 1e-5
    wo, wz = torch.randn(out.shape, generator=g), torch.randn(z.shape, generator=g) * 0.1
    ((out * wo).sum() + (z * wz).sum()).backward()
    ((oo * wo).sum() + (zo * wz).sum()).backward()
    for k, p in ae.named_parameters():
        assert rel_err(p.grad, wr[k].grad) < 1e-4, k
    with pytest.raises(NotImplementedError):
        AE(downsample=False)


def test_pool_and_stuffing_adjoints(emu_lib):
    """<pool_bwd(g), x'> and <stuff_bwd(g), x> against torch on ragged sizes."""
    import torch.nn.functional as F
    from lemo_amd.priors import cg8p_alloc, from_cg8p, to_cg8p
    g = torch.Generator().manual_seed(1)
    C, H, W = 16, 7, 10
    x = torch.randn(C, H, W, generator=g).requires_grad_(True)
    ref = F.max_pool2d(x[None], 3, 2, 1)[0]
    Ho, Wo = ref.shape[1:]
    xb, out = to_cg8p(x.detach()), cg8p_alloc(C, Ho, Wo, 'cpu')
    idx = torch.empty(C // 8, Ho * Wo, 8, dtype=torch.uint8)
    assert emu_lib.maxpool3s2_fwd(ptr(xb), H, W, ptr(out), ptr(idx), C, None) == 0
    assert torch.equal(from_cg8p(out, Ho, Wo), ref.detach())
    go = torch.randn(C, Ho, Wo, generator=g)
    ref.backward(go)
    gob, din = to_cg8p(go), cg8p_alloc(C, H, W, 'cpu')
    assert emu_lib.maxpool3s2_bwd(ptr(gob), ptr(idx), None, ptr(din), H, W, C, None) == 0
    assert rel_err(from_cg8p(din, H, W), x.grad) < 1e-6
    # stuffing: ConvTranspose2d(k3,s2,p1)(z, output_size) == conv_transpose2d(stuffed, stride 1)
    z = torch.randn(8, 4, 5, generator=g)
    wt = torch.randn(8, 8, 3, 3, generator=g)
    for tH, tW in ((7, 9), (8, 10), (7, 10)):
        S = cg8p_alloc(8, tH, tW, 'cpu')
        zb = to_cg8p(z)
        assert emu_lib.stuff2_fwd(ptr(zb), 4, 5, ptr(S), tH, tW, 8, None) == 0
        a = F.conv_transpose2d(from_cg8p(S, tH, tW)[None], wt, stride=1, padding=1)
        b = F.conv_transpose2d(z[None], wt, stride=2, padding=1, output_padding=(tH - 7, tW - 9))
        assert rel_err(a, b) < 1e-6


@pytest.mark.timeout(900)
def test_finetune_loop_matches_torch_adam(emu_lib):
    """3 finetune steps (forward, masked L1, backward, Adam lr 3e-6 -> raised to 1e-3 to be visible) + eval forward."""
    from lemo_amd.infill import AE, finetune_and_infill
    w = _weights()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 4, 18, 22, generator=g)
    mask = torch.rand(18, 22, generator=g) > 0.3
    ae = AE(_lib=emu_lib)
    rec, z = finetune_and_infill(ae, w, x, mask, steps=3, lr=1e-3)
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    opt = torch.optim.Adam(list(wr.values()), lr=1e-3)
    for _ in range(3):
        opt.zero_grad()
        r, _ = O.ae_forward(wr, x)
        (r[0, 0] - x[0, 0])[mask].abs().mean().backward()
        opt.step()
    with torch.no_grad():
        r, zr = O.ae_forward(wr, x)
    assert rec.shape == (1, 1, 16, 6)
    assert rel_err(rec, r[:, :, 1:-1, 8:-8]) < 1e-3 and rel_err(z, zr) < 1e-3
    for k, p in ae.named_parameters():
        assert float((p.detach() - wr[k].detach()).abs().max()) < 2e-5, k


def test_finetune_session_carries_nothing_from_clip_to_clip(emu_lib):
    """The finetune loop keeps its parameters, Adam state, inputs and workspace at fixed addresses across clips of one shape
    (lemo_amd.infill._FinetuneSession).  A second clip must start from the pretrained weights with a fresh optimiser
    (opt_amass_temp.py:160-164): its result equals what a brand-new session computes, bit for bit, whatever ran before."""
    from lemo_amd import infill
    from lemo_amd.infill import AE, finetune_and_infill
    w = _weights()
    g = torch.Generator().manual_seed(4)
    xa, xb = torch.randn(1, 4, 18, 22, generator=g), torch.randn(1, 4, 18, 22, generator=g)
    ma, mb = torch.rand(18, 22, generator=g) > 0.3, torch.rand(18, 22, generator=g) > 0.5
    ae = AE(_lib=emu_lib)
    infill._SESSIONS.clear()
    finetune_and_infill(ae, w, xa, ma, steps=3, lr=1e-3)                     # clip A first: the session now holds A's state
    rec_b, z_b = finetune_and_infill(ae, w, xb, mb, steps=2, lr=1e-3)         # clip B on the SAME session
    p_b = {k: v.detach().clone() for k, v in ae.state_dict().items()}
    assert len(infill._SESSIONS) == 1
    infill._SESSIONS.clear()
    ae2 = AE(_lib=emu_lib)
    rec_f, z_f = finetune_and_infill(ae2, w, xb, mb, steps=2, lr=1e-3)        # clip B on a fresh session
    assert torch.equal(rec_b, rec_f) and torch.equal(z_b, z_f)
    for k, v in ae2.state_dict().items():
        assert torch.equal(v, p_b[k]), k
    infill._SESSIONS.clear()


def test_finetune_many_clips_equals_solo_and_sessions_are_bounded(emu_lib, monkeypatch):
    """finetune_and_infill_many: clips of one shape go AE_CLIPS at a time into ONE engine whose launches carry all of them (the
    clip is the last grid dimension; own parameters / Adam state / step counter / workspace slice per clip) == clip i through
    finetune_and_infill to rounding (bit for bit for equal grouping) -- full groups, the tail group, a clip of another shape in the
    middle, the model left with the LAST clip's weights; and the session cache is an LRU of at most _MAX_SESSIONS entries"""
    from lemo_amd import infill
    from lemo_amd.infill import AE, finetune_and_infill, finetune_and_infill_many
    w = _weights()
    g = torch.Generator().manual_seed(6)
    shapes = [(18, 22), (18, 22), (18, 30), (18, 22), (18, 22), (18, 22)]         # five of one shape (2 + 2 + 1) and an odd one
    xs = [torch.randn(1, 4, h, ww, generator=g) for h, ww in shapes]
    ms = [torch.rand(h, ww, generator=g) > 0.3 for h, ww in shapes]
    infill._SESSIONS.clear()
    ae = AE(_lib=emu_lib)
    solo = [tuple(t.clone() for t in finetune_and_infill(ae, w, x, m, steps=2, lr=1e-3)) for x, m in zip(xs, ms)]
    p_last = {k: v.detach().clone() for k, v in ae.state_dict().items()}
    infill._SESSIONS.clear()
    monkeypatch.setattr(infill, 'AE_CLIPS', 2)
    ae_many = AE(_lib=emu_lib)
    many = finetune_and_infill_many(ae_many, w, xs, ms, steps=2, lr=1e-3)
    assert sorted(k[-1] for k in infill._SESSIONS) == [1, 1, 2]           # a 2-clip engine (used twice), the tail's and the odd shape's
    # Round 5: the convolutions' launch shapes are chosen for the clips IN FLIGHT (csrc/ae_engine.hip::ae_conv_shape), so a clip in a
    # K-clip engine sums its K slices in another order than alone: equal to the solo run to rounding, bit-identical between two runs of
    # the same grouping and between the slots of one engine (the padded tail repeats its last clip: same bits in both slots)
    close = lambda a, b: float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7
    for (ra, za), (rb, zb) in zip(solo, many):
        assert ra.shape == rb.shape and close(ra, rb) and close(za, zb)
    for k, v in ae_many.state_dict().items():
        assert close(v, p_last[k]), k
    again = finetune_and_infill_many(AE(_lib=emu_lib), w, xs, ms, steps=2, lr=1e-3)
    for (ra, za), (rb, zb) in zip(many, again):
        assert torch.equal(ra, rb) and torch.equal(za, zb)
    infill._SESSIONS.clear()
    monkeypatch.setattr(infill, 'AE_CLIPS', 4)
    many = finetune_and_infill_many(AE(_lib=emu_lib), w, xs[:2] + xs[3:], ms[:2] + ms[3:], steps=2, lr=1e-3)    # 4 + 1
    for (ra, za), (rb, zb) in zip(solo[:2] + solo[3:], many):
        assert close(ra, rb) and close(za, zb)
    # engine sizes are powers of two: three clips run as 2 + 1, seven on an 8-clip engine with the last one repeated (slot dropped)
    assert infill._engine_groups(3) == [(2, 2), (1, 1)] and infill._engine_groups(7) == [(7, 8)] and infill._engine_groups(5) == [(4, 4), (1, 1)]
    monkeypatch.setattr(infill, 'AE_CLIPS', 8)
    idx7 = [0, 1, 3, 4, 5, 0, 1]
    many7 = finetune_and_infill_many(AE(_lib=emu_lib), w, [xs[i] for i in idx7], [ms[i] for i in idx7], steps=2, lr=1e-3)
    for i, (rb, zb) in zip(idx7, many7):
        assert close(solo[i][0], rb) and close(solo[i][1], zb)
    assert torch.equal(many7[0][0], many7[5][0]) and torch.equal(many7[1][1], many7[6][1])      # the same clip in two slots of one engine: same bits
    for t in range(infill._MAX_SESSIONS + 3):                          # other clip shapes: the cache stays bounded
        finetune_and_infill(ae, w, torch.randn(1, 4, 18, 24 + 2 * t, generator=g), torch.ones(18, 24 + 2 * t) > 0, steps=0)
    assert len(infill._SESSIONS) == infill._MAX_SESSIONS
    infill._SESSIONS.clear()


def test_engine_rejects_bad_clip_indices(emu_lib):
    """lemo_ae_*_clip: clip outside [0, desc.clips) is an argument error; a workspace too small for the clips fails create"""
    import ctypes as C
    from lemo_amd import _hip
    from lemo_amd._hip import ptr
    n = int(emu_lib.ae_ws_floats(18, 22))
    ws = torch.zeros(2 * n)
    assert not emu_lib.ae_create(C.byref(_hip.AeDesc(18, 22, 1e-3, ptr(ws), 2 * n, 3)))
    h = emu_lib.ae_create(C.byref(_hip.AeDesc(18, 22, 1e-3, ptr(ws), 2 * n, 2)))
    assert h
    flat, x, moc = torch.zeros(int(emu_lib.ae_n_param())), torch.zeros(4, 18, 22), torch.zeros(18, 22)
    assert emu_lib.ae_load_clip(h, 2, ptr(flat), ptr(x), ptr(moc), None) != 0
    assert emu_lib.ae_load_clip(h, -1, ptr(flat), ptr(x), ptr(moc), None) != 0
    assert emu_lib.ae_params_clip(h, 2, ptr(flat), None) != 0
    emu_lib.ae_destroy(h)


@pytest.mark.parametrize('mt,pt,ks', [(0, 0, 0), (1, 1, 1), (1, 1, 4), (1, 4, 2), (2, 2, 2), (1, 2, 8), (2, 1, 8), (3, 1, 1), (3, 4, 1), (3, 2, 8)])
def test_engine_conv_all_geometries(emu_lib, mt, pt, ks):
    """lemo_ae_conv (K split over the waves of a workgroup, summed in wave order) against torch in its three geometries: plain;
    output written into the even pixels of a twice finer image (= zero-stuffed input of the next stride-2 transposed conv);
    input read at the even pixels of a finer image with the epilogue operand read there too (= adjoint of the stuffing fused
    into the backward-data convolution).  Ragged pixel counts, every epilogue."""
    import torch.nn.functional as F
    from lemo_amd.priors import cg8p_alloc, from_cg8p, to_cg8p, pack_conv3x3
    g = torch.Generator().manual_seed(11)
    cin, cout, H, W = 32, 64, 7, 9                                     # 63 pixels: one full tile + a ragged one
    x = torch.randn(cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    wt = torch.from_numpy(pack_conv3x3(w.numpy()))
    ref = F.conv2d(x[None], w, b, padding=1)[0]
    # plain, epilogues 0 / 2 / 1
    aux = torch.randn(cout, H, W, generator=g)
    for epi, want in ((0, F.leaky_relu(ref, 0.2)), (2, ref),
                      (1, F.conv2d(x[None], w, None, padding=1)[0] * torch.where(aux > 0, 1.0, 0.2))):
        out = cg8p_alloc(cout, H, W, 'cpu')
        assert emu_lib.ae_conv(ptr(to_cg8p(x)), ptr(wt), ptr(b), ptr(to_cg8p(aux)), ptr(out), H, W, 0, 0, 1, 1, cin, cout, epi, mt, pt, ks, None) == 0
        assert rel_err(from_cg8p(out, H, W), want) < 2e-6, epi
    # stuffed output: (y, x) -> (2y, 2x) of a 14 x 17 image; everything else stays zero
    fH, fW = 14, 17
    out = cg8p_alloc(cout, fH, fW, 'cpu')
    assert emu_lib.ae_conv(ptr(to_cg8p(x)), ptr(wt), ptr(b), None, ptr(out), H, W, fH, fW, 1, 2, cin, cout, 0, mt, pt, ks, None) == 0
    S = from_cg8p(out, fH, fW)
    want = torch.zeros(cout, fH, fW)
    want[:, 0:2 * H:2, 0:2 * W:2] = F.leaky_relu(ref, 0.2)
    assert rel_err(S, want) < 2e-6 and float(S[:, 1::2].abs().max()) == 0.0 and float(S[:, :, 1::2].abs().max()) == 0.0
    # strided input: the conv of a fine image evaluated at its even pixels only, times lrelu' of a stuffed operand
    xf = torch.randn(cin, fH, fW, generator=g)
    auxf = torch.zeros(cout, fH, fW)
    auxf[:, 0:2 * H:2, 0:2 * W:2] = aux
    full = F.conv2d(xf[None], w, None, padding=1)[0]
    want = full[:, 0:2 * H:2, 0:2 * W:2] * torch.where(aux > 0, 1.0, 0.2)
    out = cg8p_alloc(cout, H, W, 'cpu')
    assert emu_lib.ae_conv(ptr(to_cg8p(xf)), ptr(wt), None, ptr(to_cg8p(auxf)), ptr(out), H, W, fH, fW, 2, 1, cin, cout, 1, mt, pt, ks, None) == 0
    assert rel_err(from_cg8p(out, H, W), want) < 2e-6


@pytest.mark.parametrize('mt,pt,ks', [(0, 0, 0), (1, 1, 1), (1, 1, 4), (1, 4, 2), (2, 2, 2), (1, 2, 8), (2, 1, 8), (3, 1, 1), (3, 4, 1), (3, 2, 8)])
def test_engine_conv_f16_all_geometries(emu_lib, mt, pt, ks):
    """lemo_ae_conv_f16 (round 6: the engine's convolutions on two fp16 pieces per operand, scales from tensor maxima) in the three
    geometries of test_engine_conv_all_geometries, every epilogue, against torch in FLOAT64: error of the size of an fp32 convolution's
    own rounding (2^-22 operands), also with the input's maximum given as a bound 4 x too large (the max-pool adjoint's case) and with an
    input whose magnitudes span six decades; the launch leaves max |out| in its slot"""
    import torch.nn.functional as F
    from lemo_amd.priors import cg8p_alloc, from_cg8p, to_cg8p, pack_conv3x3
    g = torch.Generator().manual_seed(12)
    cin, cout, H, W = 32, 64, 7, 9
    x = torch.randn(cin, H, W, generator=g) * torch.logspace(0, -6, H).view(1, H, 1)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    b = torch.randn(cout, generator=g) * 1e-3
    wt = torch.from_numpy(pack_conv3x3(w.numpy()))
    wmax = w.abs().max().reshape(1).clone()
    ref = F.conv2d(x[None].double(), w.double(), b.double(), padding=1)[0]
    f32 = F.conv2d(x[None], w, b, padding=1)[0]
    aux = torch.randn(cout, H, W, generator=g)
    for fac in (1.0, 4.0):
        amax_in = (x.abs().max() / (1.0 if fac == 1.0 else 3.3)).reshape(1).clone() if fac == 4.0 else x.abs().max().reshape(1).clone()
        for epi, want in ((0, F.leaky_relu(ref, 0.2)), (2, ref),
                          (1, F.conv2d(x[None].double(), w.double(), None, padding=1)[0] * torch.where(aux > 0, 1.0, 0.2).double())):
            out, amax_out = cg8p_alloc(cout, H, W, 'cpu'), torch.zeros(1)
            assert emu_lib.ae_conv_f16(ptr(to_cg8p(x)), ptr(wt), ptr(b), ptr(to_cg8p(aux)), ptr(out), H, W, 0, 0, 1, 1, cin, cout, epi, mt, pt, ks,
                                       ptr(amax_in), fac, ptr(wmax), ptr(amax_out), None) == 0
            got = from_cg8p(out, H, W)
            assert rel_err(got.double(), want) < 1e-6, (epi, fac, rel_err(got.double(), want), rel_err(f32.double(), ref))
            assert float(amax_out) == float(got.abs().max())
    amax_in = x.abs().max().reshape(1).clone()
    # stuffed output
    fH, fW = 14, 17
    out, amax_out = cg8p_alloc(cout, fH, fW, 'cpu'), torch.zeros(1)
    assert emu_lib.ae_conv_f16(ptr(to_cg8p(x)), ptr(wt), ptr(b), None, ptr(out), H, W, fH, fW, 1, 2, cin, cout, 0, mt, pt, ks,
                               ptr(amax_in), 1.0, ptr(wmax), ptr(amax_out), None) == 0
    S = from_cg8p(out, fH, fW)
    want = torch.zeros(cout, fH, fW, dtype=torch.float64)
    want[:, 0:2 * H:2, 0:2 * W:2] = F.leaky_relu(ref, 0.2)
    assert rel_err(S.double(), want) < 1e-6 and float(S[:, 1::2].abs().max()) == 0.0 and float(S[:, :, 1::2].abs().max()) == 0.0
    # strided input with the stuffed epilogue operand
    xf = torch.randn(cin, fH, fW, generator=g)
    auxf = torch.zeros(cout, fH, fW)
    auxf[:, 0:2 * H:2, 0:2 * W:2] = aux
    full = F.conv2d(xf[None].double(), w.double(), None, padding=1)[0]
    want = full[:, 0:2 * H:2, 0:2 * W:2] * torch.where(aux > 0, 1.0, 0.2).double()
    out, amax_out = cg8p_alloc(cout, H, W, 'cpu'), torch.zeros(1)
    assert emu_lib.ae_conv_f16(ptr(to_cg8p(xf)), ptr(wt), None, ptr(to_cg8p(auxf)), ptr(out), H, W, fH, fW, 2, 1, cin, cout, 1, mt, pt, ks,
                               ptr(xf.abs().max().reshape(1).clone()), 1.0, ptr(wmax), ptr(amax_out), None) == 0
    assert rel_err(from_cg8p(out, H, W).double(), want) < 1e-6
    # refusals: 8 input channels belong to the fp32-input kernel, a bound factor below 1 is not a bound
    assert emu_lib.ae_conv_f16(ptr(to_cg8p(x)), ptr(wt), ptr(b), None, ptr(out), H, W, 0, 0, 1, 1, 8, cout, 0, 0, 0, 0, ptr(amax_in), 1.0, ptr(wmax), ptr(amax_out), None) != 0
    assert emu_lib.ae_conv_f16(ptr(to_cg8p(x)), ptr(wt), ptr(b), None, ptr(out), H, W, 0, 0, 1, 1, cin, cout, 0, 0, 0, 0, ptr(amax_in), 0.5, ptr(wmax), ptr(amax_out), None) != 0


def test_engine_f16_and_fp32_arithmetic_agree(emu_lib, monkeypatch):
    """the step engine on its split-f16 convolutions (LEMO_AE_ARITH=f16, round 6) and on the fp32-input MFMA ones (default): same
    reconstruction, latent and parameters after 3 visible steps to fp32 rounding -- the weight gradients, Adam and every layout are shared"""
    from lemo_amd import infill
    from lemo_amd.infill import AE, finetune_and_infill
    w = _weights()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 4, 34, 21, generator=g)
    mask = torch.rand(34, 21, generator=g) > 0.3
    res = {}
    for arith in ('f16', 'fp32'):
        monkeypatch.setenv('LEMO_AE_ARITH', arith)
        infill._SESSIONS.clear()
        m = AE(_lib=emu_lib)
        r, z = finetune_and_infill(m, w, x, mask, steps=3, lr=1e-3, engine=True)
        res[arith] = (r, z, {k: p.detach().clone() for k, p in m.named_parameters()})
    infill._SESSIONS.clear()
    (ra, za, pa), (rb, zb, pb) = res['f16'], res['fp32']
    assert not torch.equal(ra, rb)                                        # two arithmetics, not one
    assert rel_err(ra, rb) < 2e-5 and rel_err(za, zb) < 2e-5
    for k in pa:
        assert float((pa[k] - pb[k]).abs().max()) < 2e-5, k


@pytest.mark.timeout(900)
def test_engine_equals_autograd_path(emu_lib):
    """the native step engine (packed parameter vector, fused stuffing, one weight-gradient launch, fused reduce + Adam) and the
    round-2 path (autograd function + gather tables + flat Adam) are two implementations of the same arithmetic with different
    K-summation splits: same reconstruction, latent and finetuned parameters after 3 visible steps, to fp32 rounding; and the
    engine's padded parameter entries stay exactly zero"""
    from lemo_amd import infill
    from lemo_amd.infill import AE, finetune_and_infill
    w = _weights()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(1, 4, 34, 21, generator=g)                       # odd sizes at every level
    mask = torch.rand(34, 21, generator=g) > 0.3
    infill._SESSIONS.clear()
    a, b = AE(_lib=emu_lib), AE(_lib=emu_lib)
    ra, za = finetune_and_infill(a, w, x, mask, steps=3, lr=1e-3, engine=True)
    ses = next(iter(infill._SESSIONS.values()))
    assert isinstance(ses, infill._EngineSession)
    rb, zb = finetune_and_infill(b, w, x, mask, steps=3, lr=1e-3, engine=False)
    assert ra.shape == rb.shape and za.shape == zb.shape
    assert rel_err(ra, rb) < 2e-5 and rel_err(za, zb) < 2e-5
    moved = 0.0
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert float((p.detach() - q.detach()).abs().max()) < 2e-5, k                  # lr 1e-3: an Adam step moves every entry by ~1e-3
        moved = max(moved, float((p.detach() - w[k]).abs().max()))
    assert moved > 1e-3
    # theta's padded entries: layer 0 has 4 real input channels of 8 (forward pack [tap][cin/8][cout][8], cout 32)
    n0 = 9 * 1 * 32 * 8
    th0 = ses.ws[:n0].reshape(9, 1, 32, 8)
    assert float(th0[..., 4:].abs().max()) == 0.0 and float(th0[..., :4].abs().max()) > 0.0
    infill._SESSIONS.clear()


def test_engine_refuses_bad_use(emu_lib):
    """error behaviour of the lemo_ae_* entry points: shapes the engine does not take, a workspace that is too small, steps before
    a clip is loaded (LEMO_ERR_STATE = 10003), null arguments (LEMO_ERR_ARG = 10002)"""
    import ctypes as C
    from lemo_amd import _hip
    assert emu_lib.ae_ws_floats(1, 40) == 0 and emu_lib.ae_ws_floats(4096, 4096) == 0
    n = int(emu_lib.ae_ws_floats(18, 22))
    assert n > 0 and emu_lib.ae_n_param() == sum(v.numel() for v in _weights().values())
    ws = torch.zeros(n)
    assert not emu_lib.ae_create(C.byref(_hip.AeDesc(18, 22, 3e-6, ptr(ws), n - 64)))          # workspace too small
    assert not emu_lib.ae_create(C.byref(_hip.AeDesc(18, 22, 0.0, ptr(ws), n)))               # lr must be positive
    h = emu_lib.ae_create(C.byref(_hip.AeDesc(18, 22, 3e-6, ptr(ws), n)))
    assert h
    rec = torch.empty(18, 22)
    assert emu_lib.ae_step(h, 1, 0, None) == 10003 and emu_lib.ae_forward(h, ptr(rec), None, None) == 10003
    assert emu_lib.ae_load(h, None, None, None, None) == 10002 and emu_lib.ae_step(h, -1, 0, None) == 10002
    emu_lib.ae_destroy(h)


def test_clip_pipeline_many_equals_one_by_one(emu_lib, monkeypatch):
    """AmassClipPipeline.fit_clips (clips grouped AE_CLIPS at a time through one AE engine, then decoded and fitted one by one)
    == fit_clip clip by clip, bit for bit, on the host-emulated kernels: 3 clips with AE_CLIPS = 2 (a full group and the tail)"""
    import __graft_entry__ as ge
    from lemo_amd import infill, pipeline as P, synthetic
    from lemo_amd.fitting import AmassTemporalFitter
    from lemo_amd.infill import AE
    p = ge.small_problem()
    B = p['B']
    fit = AmassTemporalFitter(p['model'], p['vposer_w'], p['enc_w'], p['ids'], p['Xmean'], p['Xstd'], B, torch.device('cpu'), full_vertices=True, lib=emu_lib)
    pipe = P.AmassClipPipeline(fit, AE(_lib=emu_lib), _weights())
    real_decode = P.decode_markers                    # (the reduced problem fits 5 markers: keep the first 5 of the decoded 67)

    def decode5(*a, **k):
        lbl, mk = real_decode(*a, **k)
        return lbl, mk[:, :5].contiguous()
    monkeypatch.setattr(P, 'decode_markers', decode5)
    g = torch.Generator().manual_seed(12)
    items = []
    for i in range(3):
        clip = torch.randn(1, 4, 208, B, generator=g) * 0.5
        init = synthetic.make_synthetic_sequence(20 + i, B=B)['init_params']
        items.append((clip, torch.tensor([0.3 * i], dtype=torch.float64), init, 1))
    infill._SESSIONS.clear()
    solo = []
    for c, piv, init, gd in items:
        o = pipe.fit_clip(c, piv, init, gender=gd, steps=2, finetune_steps=2, use_graph=False)
        solo.append({k: v.clone() for k, v in o.items()})
    monkeypatch.setattr(infill, 'AE_CLIPS', 2)
    many = pipe.fit_clips(items, steps=2, finetune_steps=2, use_graph=False)
    # (round 5: the AE's launch shapes follow the clips in flight -- a clip finetuned in a 2-clip engine equals its solo run to rounding,
    # and so does everything downstream of its reconstruction; the contact labels are a threshold of it)
    for a, b in zip(solo, many):
        for k in ('p72', 'markers_rec', 'clip_img_rec'):
            assert float((a[k] - b[k]).abs().max()) <= 1e-4 * float(a[k].abs().max()) + 1e-6, k
        assert float((a['contact_lbl_rec'] != b['contact_lbl_rec']).float().mean()) <= 0.01
    many2 = pipe.fit_clips(items, steps=2, finetune_steps=2, use_graph=False)
    for a, b in zip(many, many2):
        for k in ('p72', 'markers_rec', 'contact_lbl_rec', 'clip_img_rec'):
            assert torch.equal(a[k], b[k]), k
    infill._SESSIONS.clear()
