"""The all-vertex forward off the critical path (lemo_fit_desc.verts_side, round 6) on the host-emulated library: the engine whose loss
path forwards the set U and whose every iteration ALSO regresses all vertices by a separate launch must (1) return the vertices of the in-line
all-vertex engine bit for bit -- of the LAST forward's parameters, although the Adam launch has rewritten `transl` since: the launch reads the
copy the set-U forward left; the tile-looping launch (125 workgroups) runs the same arithmetic per tile -- and (2) walk the same parameter
trajectory as the set-U engine, bit for bit.  (No second stream on the emulator: what is tested is the arithmetic and the bookkeeping, the
overlap is measured on the GPU.)"""
import pytest
import torch


@pytest.mark.timeout(1800)
def test_side_full_forward_equals_inline_full_forward(emu_lib):
    import __graft_entry__ as ge
    from lemo_amd.fitting import AmassTemporalFitter
    prob = ge.small_problem(B=12)
    _, markers = ge.oracle_for(prob)

    def make(full, side):
        f = AmassTemporalFitter(prob['model'], prob['vposer_w'], prob['enc_w'], prob['ids'], prob['Xmean'], prob['Xstd'], prob['B'], 'cpu',
                                full_vertices=full, lib=emu_lib, side_full_forward=side)
        f.load_sequence(prob['seq']['init_params'], markers, prob['seq']['contact_lbl'])
        return f

    inline, active, side = make(True, False), make(False, False), make(True, True)
    assert side.side_full and not side.full and side.full_output and not inline.side_full
    V = prob['V']
    # a bare forward: the all-vertex launch runs in line
    for f in (inline, active, side):
        f.forward(); f.backward()
    assert tuple(side.vertices().shape) == (prob['B'], V, 3)
    assert torch.equal(side.vertices(), inline.vertices())
    assert torch.equal(side.marker_vertices(), active.marker_vertices())         # the loss path IS the set-U path
    La, Ls = active.losses(), side.losses()
    assert all(La[k] == Ls[k] for k in La)
    ga, gs = active.grads_with_priors(), side.grads_with_priors()
    assert all(torch.equal(ga[k], gs[k]) for k in ga)
    # three iterations: same loss path, same trajectory as the set-U engine, bit for bit ...
    for f in (active, side):
        f.step(3, use_graph=False)
    assert torch.equal(side.params75(), active.params75())
    # ... and vertices() = all vertices at the parameters the LAST iteration's forward saw (`snap`: the Adam launch stores the pre-update
    # parameters there), bit-identical to an in-line all-vertex forward at those parameters -- although `transl` has been updated since
    B = prob['B']
    ref = make(True, False)
    ref.P['transl'].copy_(side.snap[:B * 3].view(B, 3))
    ref.P['rot6d'].copy_(side.snap[B * 3:B * 9].view(B, 6))
    ref.P['other'].copy_(side.snap[B * 9:].view(B, 56))
    ref.forward()
    assert not torch.equal(side.P['transl'], ref.P['transl'])                    # the update did move the translation
    assert torch.equal(side.vertices(), ref.vertices())
