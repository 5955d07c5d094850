"""The encoder's fused head and tail (conv variant 7, csrc/conv_head_kernels.hip) on the host-emulated build of the unmodified source:
the whole AMASS engine with the marker image + layer 0 + layer 1 in one launch and layer 1's backward-data + layer 0's adjoint in one
launch, against the same engine on variant 5 (marker_c1 + single-layer launches) -- x0 and act[1] bit for bit (layer 0 keeps
marker_c1_kernel's FMA order), act[2] / dx0 / losses / gradients to the split-f16 layer's rounding -- and against the oracle; the tail
kernel alone (C-ABI lemo_enc_tail) against float64 on ragged tile edges."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from lemo_amd._hip import ptr
from lemo_amd.priors import cg8p_alloc, from_cg8p, to_cg8p, pack_conv3x3_bwd_split_f16


@pytest.mark.timeout(1800)
@pytest.mark.parametrize('B', [14, 33])
def test_engine_with_fused_head_and_tail_vs_variant_5_and_oracle(emu_lib, B):
    """B = 14: one column of tiles (W = 29 -> 3 tiles of 14 columns, ragged); B = 33: W = 48 (4 tiles), H = 20 -> 2 tile rows"""
    import __graft_entry__ as ge
    from lemo_amd.fitting import AmassTemporalFitter
    prob = ge.small_problem(B=B)
    ofit, markers = ge.oracle_for(prob)
    total, parts, _, _ = ofit.losses()
    total.backward()
    fits = {}
    for v in (5, 7, 8, 9):
        fit = AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'], prob['B'], 'cpu',
                                  full_vertices=True, lib=emu_lib, conv_variant=v)
        assert fit.conv_variant == v
        fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
        fit.forward(); fit.backward()
        fits[v] = fit
    c = fits[8]                                                      # variant 8: layer 2 inside the head launch as well
    assert torch.equal(c.ws['x0'], fits[5].ws['x0']) and torch.equal(c.act[1], fits[5].act[1])
    assert rel_err(c.act[2], fits[5].act[2]) < 2e-6 and rel_err(c.act[3], fits[5].act[3]) < 2e-6 and rel_err(c.act[10], fits[5].act[10]) < 1e-5
    assert rel_err(c.ws['dx0'], fits[5].ws['dx0']) < 1e-5
    t = fits[9]                                                      # variant 9: layer 2's backward inside the tail launch as well
    assert torch.equal(t.act[3], c.act[3]) and torch.equal(t.act[10], c.act[10])      # same forward as variant 8
    assert rel_err(t.ws['dx0'], fits[5].ws['dx0']) < 1e-5 and rel_err(t.ws['dx0'], c.ws['dx0']) < 2e-6
    a, b = fits[5], fits[7]
    assert torch.equal(a.ws['x0'], b.ws['x0']) and torch.equal(a.ws['canon'], b.ws['canon'])
    assert torch.equal(a.act[1], b.act[1])                           # layer 0: the same FMAs in the same order
    assert rel_err(b.act[2], a.act[2]) < 2e-6 and rel_err(b.act[10], a.act[10]) < 1e-5
    assert rel_err(b.ws['dx0'], a.ws['dx0']) < 1e-5
    L = b.losses()
    for k in ('marker', 'vposer', 'shape', 'hand', 'contact', 'smooth'):
        assert abs(L[k] - float(parts[k])) <= 1e-5 * abs(float(parts[k])), (k, L[k], float(parts[k]))
    g = b.grads_with_priors()
    for k, ref in (('transl', ofit.transl.grad), ('rot6d', ofit.rot6d.grad), ('other', ofit.other.grad)):
        assert rel_err(g[k], ref) < 2e-4, k
    a.step(2, use_graph=False); b.step(2, use_graph=False)
    assert float((a.params75() - b.params75()).abs().max()) < 2e-5


@pytest.mark.timeout(900)
@pytest.mark.parametrize('H,W', [(10, 14), (7, 9), (23, 31)])
def test_tail_kernel_vs_float64(emu_lib, H, W):
    g = torch.Generator().manual_seed(H * 50 + W)
    w1 = torch.randn(32, 32, 3, 3, generator=g) * 0.08
    w0 = torch.randn(32, 1, 3, 3, generator=g) * 0.3
    d2 = torch.randn(32, H, W, generator=g) * 1e-5
    a1 = torch.randn(32, H, W, generator=g)
    pb, ib = pack_conv3x3_bwd_split_f16(w1.numpy())
    pb = torch.from_numpy(pb.view(np.int16))
    d1 = F.conv_transpose2d(d2[None].double(), w1.double(), padding=1)[0] * torch.where(a1 > 0, 1.0, 0.2).double()
    ref = F.conv_transpose2d(d1[None], w0.double(), padding=1)[0, 0]
    dx0 = torch.zeros(H * W)
    assert emu_lib.enc_tail(ptr(to_cg8p(d2)), ptr(pb), ib, ptr(to_cg8p(a1)), ptr(w0.reshape(32, 9).contiguous()), ptr(dx0), H, W, None) == 0
    assert rel_err(dx0.view(H, W).double(), ref) < 2e-6
    assert emu_lib.enc_tail(None, ptr(pb), ib, ptr(to_cg8p(a1)), ptr(w0.reshape(32, 9).contiguous()), ptr(dx0), H, W, None) != 0


@pytest.mark.timeout(900)
@pytest.mark.parametrize('H,W', [(10, 14), (7, 9), (23, 31)])
def test_tail3_kernel_vs_float64(emu_lib, H, W):
    """variant 9's tail (layer 2, 1, 0 backwards in one launch) alone, ragged tile edges, against float64"""
    g = torch.Generator().manual_seed(H * 70 + W)
    w2 = torch.randn(64, 32, 3, 3, generator=g) * 0.06
    w1 = torch.randn(32, 32, 3, 3, generator=g) * 0.08
    w0 = torch.randn(32, 1, 3, 3, generator=g) * 0.3
    d3 = torch.randn(64, H, W, generator=g) * 1e-5
    a2 = torch.randn(32, H, W, generator=g)
    a1 = torch.randn(32, H, W, generator=g)
    p2, i2 = pack_conv3x3_bwd_split_f16(w2.numpy())
    p1, i1 = pack_conv3x3_bwd_split_f16(w1.numpy())
    p2, p1 = torch.from_numpy(p2.view(np.int16)), torch.from_numpy(p1.view(np.int16))
    d2 = F.conv_transpose2d(d3[None].double(), w2.double(), padding=1)[0] * torch.where(a2 > 0, 1.0, 0.2).double()
    d1 = F.conv_transpose2d(d2[None], w1.double(), padding=1)[0] * torch.where(a1 > 0, 1.0, 0.2).double()
    ref = F.conv_transpose2d(d1[None], w0.double(), padding=1)[0, 0]
    dx0 = torch.zeros(H * W)
    w0f = w0.reshape(32, 9).contiguous()
    assert emu_lib.enc_tail3(ptr(to_cg8p(d3)), ptr(p2), i2, ptr(to_cg8p(a2)), ptr(p1), i1, ptr(to_cg8p(a1)), ptr(w0f), ptr(dx0), H, W, None) == 0
    assert rel_err(dx0.view(H, W).double(), ref) < 3e-6
    assert emu_lib.enc_tail3(ptr(to_cg8p(d3)), None, i2, ptr(to_cg8p(a2)), ptr(p1), i1, ptr(to_cg8p(a1)), ptr(w0f), ptr(dx0), H, W, None) != 0
