"""Fused encoder layer pairs (conv variant 5, csrc/conv_pair_kernels.hip) on the host-emulated build of the unmodified kernel
source: two 64 -> 64 3x3 layers per launch on 10 x 14 tiles, intermediate in LDS -- forward (lrelu(conv + bias) twice, the
intermediate activation also written out) and backward-data (conv x lrelu'(saved activation) twice) against torch in float64."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from lemo_amd._hip import ptr
from lemo_amd.priors import (cg8p_alloc, from_cg8p, to_cg8p, pack_conv3x3_split_f16, pack_conv3x3_bwd_split_f16)


def _packs(w):
    pf, fi = pack_conv3x3_split_f16(w.numpy())
    pb, bi = pack_conv3x3_bwd_split_f16(w.numpy())
    return torch.from_numpy(pf.view(np.int16)), fi, torch.from_numpy(pb.view(np.int16)), bi


def _border_is_zero(buf, H, W):
    b = buf.reshape(-1, H + 2, W + 2, 8)
    return float(b[:, 0].abs().max()) == 0.0 and float(b[:, -1].abs().max()) == 0.0 and float(b[:, :, 0].abs().max()) == 0.0 and \
        float(b[:, :, -1].abs().max()) == 0.0


@pytest.mark.timeout(1800)
@pytest.mark.parametrize('kernel', ['conv3x3_pair_f16'])
@pytest.mark.parametrize('H,W', [(10, 14), (7, 9), (23, 31), (36, 57)])
def test_pair_forward_and_backward_vs_float64(emu_lib, H, W, kernel):
    """one tile exactly / less than one tile / ragged edges in both directions / several tiles per XCD run -- for the 8-wave kernel on
    10 x 14 tiles (variant 5; its four-wave twin, variant 6, moved to csrc/attic in round 6)"""
    pair = getattr(emu_lib, kernel)
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(64, H, W, generator=g)
    w1, w2 = torch.randn(64, 64, 3, 3, generator=g) * 0.06, torch.randn(64, 64, 3, 3, generator=g) * 0.06
    b1, b2 = torch.randn(64, generator=g) * 0.3, torch.randn(64, generator=g) * 0.3
    p1, i1, pb1, ib1 = _packs(w1)
    p2, i2, pb2, ib2 = _packs(w2)
    a1_64 = F.leaky_relu(F.conv2d(x[None].double(), w1.double(), b1.double(), padding=1), 0.2)
    a2_64 = F.leaky_relu(F.conv2d(a1_64, w2.double(), b2.double(), padding=1), 0.2)[0]
    a1_32 = F.leaky_relu(F.conv2d(x[None], w1, b1, padding=1), 0.2)
    a2_32 = F.leaky_relu(F.conv2d(a1_32, w2, b2, padding=1), 0.2)[0]
    xin, mid, out = to_cg8p(x), cg8p_alloc(64, H, W, 'cpu'), cg8p_alloc(64, H, W, 'cpu')
    assert pair(ptr(xin), ptr(p1), i1, ptr(b1), None, ptr(mid), ptr(p2), i2, ptr(b2), None, ptr(out), H, W, 0, None, None) == 0
    e_mid, e_out = rel_err(from_cg8p(mid, H, W).double(), a1_64[0]), rel_err(from_cg8p(out, H, W).double(), a2_64)
    f_mid, f_out = rel_err(a1_32[0].double(), a1_64[0]), rel_err(a2_32.double(), a2_64)
    assert e_mid < 2e-6 and e_mid < 4 * f_mid, (e_mid, f_mid)
    assert e_out < 2e-6 and e_out < 4 * f_out, (e_out, f_out)
    assert _border_is_zero(mid, H, W) and _border_is_zero(out, H, W)
    # the pair equals two single-layer launches of the same arithmetic family up to the intermediate's tile scale (fp32-sized)
    s_mid, s_out = cg8p_alloc(64, H, W, 'cpu'), cg8p_alloc(64, H, W, 'cpu')
    if emu_lib.conv3x3_split_supported(H, W, 64, 64):
        from lemo_amd.priors import pack_conv3x3
        wt1, wt2 = torch.from_numpy(pack_conv3x3(w1.numpy())), torch.from_numpy(pack_conv3x3(w2.numpy()))
        assert emu_lib.conv3x3_mfma_split_f16(ptr(xin), ptr(p1), i1, ptr(wt1), ptr(b1), None, ptr(s_mid), H, W, 64, 64, 0, None) == 0
        assert emu_lib.conv3x3_mfma_split_f16(ptr(s_mid), ptr(p2), i2, ptr(wt2), ptr(b2), None, ptr(s_out), H, W, 64, 64, 0, None) == 0
        assert rel_err(out, s_out) < 2e-6
    # ---- backward-data pair: d(pre 2) -> d(pre 1) = convT(., w2) * lrelu'(a1) -> d(pre 0) = convT(., w1) * lrelu'(a0)
    d2 = torch.randn(64, H, W, generator=g) * 1e-6
    a0 = torch.randn(64, H, W, generator=g)
    a1 = a1_32[0]
    xr = x.clone().double().requires_grad_(True)
    pre1 = F.conv2d(xr[None], w1.double(), b1.double(), padding=1)
    y1 = F.leaky_relu(pre1, 0.2)
    pre2 = F.conv2d(y1, w2.double(), b2.double(), padding=1)
    pre2.backward(d2[None].double())
    # torch's own chain uses lrelu'(pre1); the kernel takes the sign from the saved activation a1 = lrelu(pre1): same sign
    ref_d0 = xr.grad * torch.where(a0 > 0, 1.0, 0.2).double()
    d2b, a1b, a0b, d0b = to_cg8p(d2), to_cg8p(a1), to_cg8p(a0), cg8p_alloc(64, H, W, 'cpu')
    assert pair(ptr(d2b), ptr(pb2), ib2, None, ptr(a1b), None, ptr(pb1), ib1, None, ptr(a0b), ptr(d0b), H, W, 1, None, None) == 0
    assert rel_err(from_cg8p(d0b, H, W).double(), ref_d0) < 2e-6
    assert _border_is_zero(d0b, H, W)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('kernel', ['conv3x3_pair_f16'])
def test_pair_range_homogeneity_and_zero_input(emu_lib, kernel):
    """per-workgroup power-of-two scales: magnitudes falling by 8 orders across the image keep fp32-sized errors row by row, the
    result is exactly homogeneous under power-of-two scalings (zero bias), and an all-zero input gives lrelu(conv(lrelu(b1)) + b2)"""
    H, W = 36, 29
    g = torch.Generator().manual_seed(5)
    x = torch.randn(64, H, W, generator=g) * (10.0 ** (-8.0 * torch.arange(H) / H))[None, :, None]
    w1, w2 = torch.randn(64, 64, 3, 3, generator=g) * 0.06, torch.randn(64, 64, 3, 3, generator=g) * 0.06
    zb = torch.zeros(64)
    p1, i1, _, _ = _packs(w1)
    p2, i2, _, _ = _packs(w2)
    ref = F.leaky_relu(F.conv2d(F.leaky_relu(F.conv2d(x[None].double(), w1.double(), None, padding=1), 0.2), w2.double(), None, padding=1), 0.2)[0]

    def run(xx, ba, bb):
        mid, out = cg8p_alloc(64, H, W, 'cpu'), cg8p_alloc(64, H, W, 'cpu')
        assert getattr(emu_lib, kernel)(ptr(to_cg8p(xx)), ptr(p1), i1, ptr(ba), None, ptr(mid), ptr(p2), i2, ptr(bb), None, ptr(out), H, W, 0, None, None) == 0
        return mid, out
    mid, out = run(x, zb, zb)
    got = from_cg8p(out, H, W).double()
    for y in range(H):
        loc = ref[:, max(0, y - 14):y + 15].abs().max()
        assert float((got[:, y] - ref[:, y]).abs().max() / loc) < 4e-6, y
    for k in (-30, 20):
        mk, ok = run(x * 2.0 ** k, zb, zb)
        assert torch.equal(ok, out * 2.0 ** k) and torch.equal(mk, mid * 2.0 ** k), k
    b1, b2 = torch.randn(64, generator=g), torch.randn(64, generator=g)
    _, oz = run(torch.zeros(64, H, W), b1, b2)
    refz = F.leaky_relu(F.conv2d(F.leaky_relu(b1, 0.2)[None, :, None, None].expand(1, 64, H, W), w2, b2, padding=1), 0.2)[0]
    assert rel_err(from_cg8p(oz, H, W), refz) < 2e-6


@pytest.mark.timeout(900)
@pytest.mark.parametrize('kernel', ['conv3x3_pair_f16'])
@pytest.mark.parametrize('case', ['second_phase_zero', 'second_phase_tiny', 'first_phase_zero'])
def test_pair_and_single_layer_with_a_vanishing_staging_phase(emu_lib, case, kernel):
    """the kernels stage channels {0-15, 32-47} and {16-31, 48-63} in two phases with their own power-of-two
    scales and rescale the accumulators by the ratio in between.  A second phase that is exactly zero (scale clamped to 2^126) or
    2^90 below the first made that ratio overflow: all-NaN tiles with rc == 0.  The later phase's scale is now bounded by the
    earlier one's (conv_f16.hpp::f16_scale_after); every output stays finite and fp32-accurate."""
    H, W = 16, 20
    g = torch.Generator().manual_seed(77)
    x = torch.randn(64, H, W, generator=g)
    second = torch.zeros(64, dtype=torch.bool)
    second[16:32] = True                    # the second staging phase: channel groups 2-3 and 6-7
    second[48:64] = True
    if case == 'second_phase_zero':
        x[second] = 0.0
    elif case == 'second_phase_tiny':
        x[second] *= 2.0 ** -95
    else:
        x[~second] = 0.0
    w1, w2 = torch.randn(64, 64, 3, 3, generator=g) * 0.06, torch.randn(64, 64, 3, 3, generator=g) * 0.06
    b1, b2 = torch.randn(64, generator=g) * 0.3, torch.randn(64, generator=g) * 0.3
    p1, i1, _, _ = _packs(w1)
    p2, i2, _, _ = _packs(w2)
    a1 = F.leaky_relu(F.conv2d(x[None].double(), w1.double(), b1.double(), padding=1), 0.2)
    a2 = F.leaky_relu(F.conv2d(a1, w2.double(), b2.double(), padding=1), 0.2)[0]
    xin, mid, out = to_cg8p(x), cg8p_alloc(64, H, W, 'cpu'), cg8p_alloc(64, H, W, 'cpu')
    assert getattr(emu_lib, kernel)(ptr(xin), ptr(p1), i1, ptr(b1), None, ptr(mid), ptr(p2), i2, ptr(b2), None, ptr(out), H, W, 0, None, None) == 0
    assert bool(torch.isfinite(mid).all()) and bool(torch.isfinite(out).all())
    assert rel_err(from_cg8p(mid, H, W).double(), a1[0]) < 2e-6 and rel_err(from_cg8p(out, H, W).double(), a2) < 2e-6
    if emu_lib.conv3x3_split_supported(H, W, 64, 64):          # the single-layer kernel the pair's staging was taken from
        from lemo_amd.priors import pack_conv3x3
        wt1 = torch.from_numpy(pack_conv3x3(w1.numpy()))
        s_mid = cg8p_alloc(64, H, W, 'cpu')
        assert emu_lib.conv3x3_mfma_split_f16(ptr(xin), ptr(p1), i1, ptr(wt1), ptr(b1), None, ptr(s_mid), H, W, 64, 64, 0, None) == 0
        assert bool(torch.isfinite(s_mid).all()) and rel_err(from_cg8p(s_mid, H, W).double(), a1[0]) < 2e-6


def test_pair_rejects_bad_arguments(emu_lib):
    x = cg8p_alloc(64, 12, 20, 'cpu')
    w = torch.zeros(4 * 9 * 2 * 2 * 64 * 8, dtype=torch.int16)
    b = torch.zeros(64)
    assert emu_lib.conv3x3_pair_f16(ptr(x), ptr(w), 1.0, ptr(b), None, None, ptr(w), 1.0, ptr(b), None, ptr(x), 12, 20, 0, None, None) != 0   # forward needs `mid`
    assert emu_lib.conv3x3_pair_f16(ptr(x), ptr(w), 1.0, None, None, None, ptr(w), 1.0, None, None, ptr(x), 12, 20, 1, None, None) != 0      # backward needs aux
    assert emu_lib.conv3x3_pair_f16(ptr(x), ptr(w), 0.0, ptr(b), None, ptr(x), ptr(w), 1.0, ptr(b), None, ptr(x), 12, 20, 0, None, None) != 0
    assert emu_lib.conv3x3_pair_f16(ptr(x), ptr(w), 1.0, ptr(b), None, ptr(x), ptr(w), 1.0, ptr(b), None, ptr(x), 12, 20, 2, None, None) != 0
