"""Winograd F(2x2, 3x3) encoder layer (conv variant 10, csrc/conv_wino_kernels.hip) on the host-emulated build of the unmodified
kernel source: one 64 -> 64 3x3 layer per launch, 32 tiles of 2 x 2 outputs per workgroup -- forward (lrelu(conv + bias)) and
backward-data (conv of the flipped, transposed weights x lrelu'(saved activation)) against torch in float64."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from lemo_amd._hip import ptr
from lemo_amd.priors import (cg8p_alloc, from_cg8p, to_cg8p, pack_conv3x3, pack_conv3x3_bwd, pack_conv3x3_wino_f16,
                             pack_conv3x3_bwd_wino_f16)


def _packs(w):
    uf, fi = pack_conv3x3_wino_f16(w.numpy())
    ub, bi = pack_conv3x3_bwd_wino_f16(w.numpy())
    return (torch.from_numpy(uf.view(np.int16)), fi, torch.from_numpy(pack_conv3x3(w.numpy())),
            torch.from_numpy(ub.view(np.int16)), bi, torch.from_numpy(pack_conv3x3_bwd(w.numpy())))


def _border_is_zero(buf, H, W):
    b = buf.reshape(-1, H + 2, W + 2, 8)
    return float(b[:, 0].abs().max()) == 0.0 and float(b[:, -1].abs().max()) == 0.0 and float(b[:, :, 0].abs().max()) == 0.0 and \
        float(b[:, :, -1].abs().max()) == 0.0


def test_wino_pack_is_the_transform_of_the_filter():
    """U = G g G^T per (cout, cin), two fp16 pieces of U * 2^k in MFMA A-fragment order: unpacking reproduces U to 2^-22 of its maximum"""
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).numpy()
    pack, winv = pack_conv3x3_wino_f16(w)
    assert pack.shape == (16, 4, 2, 2, 64, 8) and pack.dtype == np.uint16
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    U = np.einsum('ij,ocjk,lk->ocil', G, w.astype(np.float64), G).reshape(64, 64, 16)
    f = pack.view(np.float16).astype(np.float64)               # [pos][ks][mt][piece][lane][e]
    rec = (f[:, :, :, 0] + f[:, :, :, 1]) * winv               # [pos][ks][mt][lane][e]
    rec = rec.reshape(16, 4, 2, 2, 32, 8)                      # [pos][ks][mt][h][i][e]
    rec = rec.transpose(2, 4, 1, 3, 5, 0).reshape(64, 64, 16)  # [mt][i] -> cout ; [ks][h][e] -> cin
    assert np.abs(rec - U).max() <= 2.0 ** -21 * np.abs(U).max()


@pytest.mark.timeout(1800)
@pytest.mark.parametrize('H,W', [(8, 16), (7, 9), (23, 31), (36, 58), (2, 1)])
def test_wino_forward_and_backward_vs_float64(emu_lib, H, W):
    """exactly one workgroup of whole tiles / odd H and odd W (direct last row, dropped last column) / ragged, several workgroups /
    even sizes over several XCD runs / the smallest image the kernel takes"""
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(64, H, W, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.06
    b = torch.randn(64, generator=g) * 0.3
    uf, fi, wt, ub, bi, wtb = _packs(w)
    assert emu_lib.conv3x3_wino_supported(H, W, 64, 64) == 1 and emu_lib.conv3x3_wino_supported(H, W, 32, 64) == 0
    y64 = F.leaky_relu(F.conv2d(x[None].double(), w.double(), b.double(), padding=1), 0.2)[0]
    y32 = F.leaky_relu(F.conv2d(x[None], w, b, padding=1), 0.2)[0]
    xin, out = to_cg8p(x), cg8p_alloc(64, H, W, 'cpu')
    assert emu_lib.conv3x3_wino_f16(ptr(xin), ptr(uf), fi, ptr(wt), ptr(b), None, ptr(out), H, W, 0, None, None) == 0
    e, f = rel_err(from_cg8p(out, H, W).double(), y64), rel_err(y32.double(), y64)
    assert e < 2e-6 and e < 4 * f + 2e-7, (e, f)
    assert _border_is_zero(out, H, W)
    # backward-data: d(pre 1) -> convT(., w) * lrelu'(a0)
    d1 = torch.randn(64, H, W, generator=g) * 1e-6
    a0 = torch.randn(64, H, W, generator=g)
    ref = F.conv_transpose2d(d1[None].double(), w.double(), padding=1)[0] * torch.where(a0 > 0, 1.0, 0.2).double()
    d1b, a0b, d0b = to_cg8p(d1), to_cg8p(a0), cg8p_alloc(64, H, W, 'cpu')
    assert emu_lib.conv3x3_wino_f16(ptr(d1b), ptr(ub), bi, ptr(wtb), None, ptr(a0b), ptr(d0b), H, W, 1, None, None) == 0
    assert rel_err(from_cg8p(d0b, H, W).double(), ref) < 2e-6
    assert _border_is_zero(d0b, H, W)
    # arguments
    assert emu_lib.conv3x3_wino_f16(ptr(xin), ptr(uf), fi, ptr(wt), None, None, ptr(out), H, W, 0, None, None) != 0      # epi 0 needs a bias
    assert emu_lib.conv3x3_wino_f16(ptr(xin), ptr(uf), 0.0, ptr(wt), ptr(b), None, ptr(out), H, W, 0, None, None) != 0    # scale > 0


@pytest.mark.timeout(900)
def test_wino_range_homogeneity_and_zero_input(emu_lib):
    """per-workgroup power-of-two scale: magnitudes falling by 8 orders down the image keep fp32-sized errors row by row, the result is
    exactly homogeneous under power-of-two scalings (zero bias), and an all-zero input gives lrelu(bias) everywhere"""
    H, W = 36, 29
    g = torch.Generator().manual_seed(5)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.06
    uf, fi, wt, _, _, _ = _packs(w)
    x = torch.randn(64, H, W, generator=g) * torch.logspace(0, -8, H).view(1, H, 1)
    b0 = torch.zeros(64)
    xin, out = to_cg8p(x), cg8p_alloc(64, H, W, 'cpu')
    assert emu_lib.conv3x3_wino_f16(ptr(xin), ptr(uf), fi, ptr(wt), ptr(b0), None, ptr(out), H, W, 0, None, None) == 0
    got = from_cg8p(out, H, W).double()
    ref = F.leaky_relu(F.conv2d(x[None].double(), w.double(), padding=1), 0.2)[0]
    # a workgroup's 32 tiles span at most two tile rows = 4 image rows (+ the halo): its scale is set by rows a factor <= 10^1.2 apart
    for y in range(0, H, 4):
        blk = slice(y, min(y + 4, H))
        assert float((got[:, blk] - ref[:, blk]).abs().max() / ref[:, blk].abs().max()) < 4e-6, y
    out2 = cg8p_alloc(64, H, W, 'cpu')
    assert emu_lib.conv3x3_wino_f16(ptr(to_cg8p(x * 2.0 ** -7)), ptr(uf), fi, ptr(wt), ptr(b0), None, ptr(out2), H, W, 0, None, None) == 0
    assert torch.equal(out2 * 2.0 ** 7, out)
    b = torch.randn(64, generator=g)
    outz = cg8p_alloc(64, H, W, 'cpu')
    assert emu_lib.conv3x3_wino_f16(ptr(cg8p_alloc(64, H, W, 'cpu')), ptr(uf), fi, ptr(wt), ptr(b), None, ptr(outz), H, W, 0, None, None) == 0
    assert torch.equal(from_cg8p(outz, H, W), F.leaky_relu(b, 0.2).view(64, 1, 1).expand(64, H, W))


@pytest.mark.timeout(1800)
@pytest.mark.parametrize('B', [14, 33])
def test_engine_on_variant_10_vs_variant_9_and_oracle(emu_lib, B):
    """the whole AMASS engine with every 64 -> 64 layer of the encoder as a Winograd launch (conv variant 10: enc_head3, 7 forward and
    7 backward-data Winograd launches, enc_tail3) against the same engine on variant 9 (fused pairs) and against the oracle: head
    launch bit-identical (act[1..3]), everything behind it to a convolution's rounding, six losses <= 1e-5, gradients, two Adam steps"""
    import __graft_entry__ as ge
    from lemo_amd.fitting import AmassTemporalFitter
    prob = ge.small_problem(B=B)
    ofit, markers = ge.oracle_for(prob)
    total, parts, _, _ = ofit.losses()
    total.backward()
    fits = {}
    for v in (9, 10):
        fit = AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'], prob['B'], 'cpu',
                                  full_vertices=True, lib=emu_lib, conv_variant=v)
        assert fit.conv_variant == v
        fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
        fit.forward(); fit.backward()
        fits[v] = fit
    a, b = fits[9], fits[10]
    assert torch.equal(a.ws['x0'], b.ws['x0']) and all(torch.equal(a.act[l], b.act[l]) for l in (1, 2, 3))
    for l in range(4, 11):
        assert rel_err(b.act[l], a.act[l]) < 3e-6 * (l - 2), l
    assert rel_err(b.ws['dx0'], a.ws['dx0']) < 1e-5
    L = b.losses()
    for k in ('marker', 'vposer', 'shape', 'hand', 'contact', 'smooth'):
        assert abs(L[k] - float(parts[k])) <= 1e-5 * abs(float(parts[k])), (k, L[k], float(parts[k]))
    g = b.grads_with_priors()
    for k, ref in (('transl', ofit.transl.grad), ('rot6d', ofit.rot6d.grad), ('other', ofit.other.grad)):
        assert rel_err(g[k], ref) < 2e-4, k
    a.step(2, use_graph=False); b.step(2, use_graph=False)
    assert float((a.params75() - b.params75()).abs().max()) < 2e-5
