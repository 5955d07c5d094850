"""The decision-conditioned parity machinery (tests/kink_attribution.py) on the host-emulated kernels: the small AMASS problem through the
native engine, the fp32 CPU oracle and the float64 oracle -- every frame's gradient, conditioned on the engine's own kink decisions, inside
the computed bound; a decision flipped BY HAND is found, attributed to its family and attributed to the frames in its reach."""
import numpy as np
import pytest
import torch

import kink_attribution as KA


def _problem(emu_lib, B=14, full=True):
    import __graft_entry__ as ge
    from lemo_amd.fitting import AmassTemporalFitter
    from oracle.f64 import amass_fit_oracle_f64
    prob = ge.small_problem(B=B)
    o32, markers = ge.oracle_for(prob)
    ej = list(range(21))
    o64 = amass_fit_oracle_f64(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'],
                               prob['seq']['init_params'], markers, prob['seq']['contact_lbl'], extra_joint_ids=ej)
    fit = AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'], prob['B'], 'cpu',
                              full_vertices=full, lib=emu_lib)
    fit.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
    return fit, o32, o64


@pytest.mark.timeout(900)
def test_conditioned_gradient_parity_small_problem(emu_lib):
    fit, o32, o64 = _problem(emu_lib)
    out = KA.analyse(fit, o32, o64, label='small problem (emulated kernels)')
    KA.check(out, 'small problem')
    # conditioning can only remove error: the conditioned worst frame is never worse than the unconditioned one by more than rounding
    assert float(out['cond_gpu'].max()) <= float(out['unc_gpu'].max()) + KA.ROUND


@pytest.mark.timeout(900)
def test_a_flipped_decision_is_found_and_explained(emu_lib):
    """float64 pinned to decisions with ONE VPoser unit and ONE L1 sign flipped by hand moves exactly the frames in their reach; the diff
    lists both with their family and frame; pinning the unflipped decisions restores the gradient"""
    from oracle.f64 import default_f64
    fit, o32, o64 = _problem(emu_lib)
    with default_f64():
        v64, a64, h64 = KA.own_forward(o64)
    d64 = KA.decisions_of(v64, a64, o64, h64)
    G0, _ = KA.conditioned_grads(o64, d64)
    with default_f64():
        Gplain = torch.autograd.grad(o64.losses()[0], (o64.transl, o64.rot6d, o64.other))
    for g, k in zip(Gplain, ('transl', 'rot6d', 'other')):           # float64's own decisions pinned = the plain objective, bit for bit almost
        assert float((g - G0[k]).abs().max()) <= 1e-12 * float(g.abs().max())
    flipped = dict(d64, vposer=[h.clone() for h in d64['vposer']], l1_sign=d64['l1_sign'].clone())
    flipped['vposer'][1][5, 100] = ~flipped['vposer'][1][5, 100]
    flipped['l1_sign'][9, 2, 1] *= -1
    diffs = KA.diff_decisions(flipped, d64, o64, v64, a64)
    assert sorted((d[0], d[2][0]) for d in diffs) == [('l1', 9), ('vposer_fc2', 5)]
    moved_by = lambda dec: torch.stack([(KA.conditioned_grads(o64, dec)[0][k] - G0[k]).abs().max(1).values / G0[k].abs().max() for k in G0]).max(0).values
    # an L1 sign reaches its own frame only; a VPoser unit moves its frame's pose (value AND slope: an arbitrary unit is not near its kink),
    # which the temporal encoder spreads over +-10 frames -- its own frame moves most
    m_l1 = moved_by(dict(d64, l1_sign=flipped['l1_sign']))
    assert (m_l1 > 0).nonzero().flatten().tolist() == [9]
    m_vp = moved_by(dict(d64, vposer=flipped['vposer']))
    assert int(m_vp.argmax()) in (4, 5, 6) and float(m_vp[5]) > 1e-6          # (the smoothness term differences neighbouring frames)
