"""the third-party import surfaces LEMO's hot path touches -- ``smplx.lbs.lbs`` /
``smplx.lbs.transform_mat`` (human_body_prior/body_model/body_model.py:29, temp_prox/camera.py:27) and the importable
``chamfer`` stub (temp_prox/dist_chamfer.py:27,43) -- served by lemo_amd.compat.  ``lbs`` runs on the (host-emulated)
HIP kernels and is checked against the oracle's lbs(), which is pinned to the reference's vendored lbs.py."""
import sys

import numpy as np
import pytest
import torch

from conftest import rel_err
from lemo_amd import synthetic
from oracle import lemo_oracle as O


def _lbs_args(V=640, nb=16, seed=3):
    m = synthetic.make_synthetic_smplx(seed=seed, V=V, F=1200)
    so = O.SmplxOracle(m, extra_joint_ids=[0])
    shapedirs = torch.cat([so.shapedirs, so.expr_dirs], dim=-1)[:, :, :nb].contiguous()      # [V,3,nb]
    return dict(v_template=so.v_template, shapedirs=shapedirs, posedirs=so.posedirs, J_regressor=so.J_regressor,
                parents=so.parents, lbs_weights=so.lbs_weights)


def test_lbs_function_forward_and_gradients_vs_oracle(emu_lib):
    from lemo_amd.compat.smplx.lbs import lbs
    a = _lbs_args()
    g = torch.Generator().manual_seed(11)
    B = 3
    betas = (torch.randn(B, 16, generator=g) * 0.5).requires_grad_(True)
    pose = (torch.randn(B, 165, generator=g) * 0.2).requires_grad_(True)
    v_ref, j_ref = O.lbs(betas, pose, **a)
    wv, wj = torch.randn(v_ref.shape, generator=g), torch.randn(j_ref.shape, generator=g)
    ((v_ref * wv).sum() + (j_ref * wj).sum()).backward()
    gb, gp = betas.grad.clone(), pose.grad.clone()
    betas.grad = pose.grad = None
    verts, joints = lbs(betas, pose, a['v_template'], a['shapedirs'], a['posedirs'], a['J_regressor'], a['parents'],
                        a['lbs_weights'], pose2rot=True, _lib=emu_lib)
    assert verts.shape == v_ref.shape and joints.shape == (B, 55, 3)
    assert rel_err(verts.detach(), v_ref.detach()) < 1e-4 and rel_err(joints.detach(), j_ref.detach()) < 1e-4
    ((verts * wv).sum() + (joints * wj).sum()).backward()
    assert rel_err(betas.grad, gb) < 1e-4 and rel_err(pose.grad, gp) < 1e-4
    # same tensors again: the prepared model is reused; a [1,V,3] template (as smplx passes it) is accepted
    from lemo_amd.compat.smplx import lbs as L
    n = len(L._CACHE)
    v2, _ = lbs(betas.detach(), pose.detach(), a['v_template'], a['shapedirs'], a['posedirs'], a['J_regressor'],
                a['parents'], a['lbs_weights'], _lib=emu_lib)
    assert len(L._CACHE) == n and torch.equal(v2, verts.detach())


def test_lbs_function_rejects_what_it_does_not_cover(emu_lib):
    from lemo_amd.compat.smplx.lbs import lbs
    a = _lbs_args(nb=10)
    betas, pose = torch.zeros(1, 10), torch.zeros(1, 165)
    with pytest.raises(NotImplementedError):
        lbs(betas, pose, **a, pose2rot=False, _lib=emu_lib)
    with pytest.raises(NotImplementedError):                       # a 24-joint SMPL skeleton
        lbs(betas, pose[:, :72], a['v_template'], a['shapedirs'], a['posedirs'][:207], a['J_regressor'][:24],
            a['parents'][:24], a['lbs_weights'][:, :24], _lib=emu_lib)


def test_transform_mat_matches_oracle():
    from lemo_amd.compat.smplx.lbs import transform_mat
    g = torch.Generator().manual_seed(0)
    R, t = torch.randn(5, 3, 3, generator=g), torch.randn(5, 3, 1, generator=g)
    assert torch.equal(transform_mat(R, t), O.transform_mat(R, t))


def test_install_registers_reference_import_names():
    import lemo_amd.compat as compat
    saved = {k: sys.modules.get(k) for k in ('smplx', 'smplx.lbs', 'chamfer')}
    try:
        for k in saved:
            sys.modules.pop(k, None)
        compat.install()
        import chamfer
        import smplx
        from smplx.lbs import lbs, transform_mat            # noqa: F401  (the reference's own import lines)
        assert smplx.create is compat.smplx.create and callable(lbs)
        with pytest.raises(NotImplementedError):
            chamfer.forward(None, None, None, None, None, None)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_load_vposer_reads_snapshot_and_ini(tmp_path):
    """human_body_prior/tools/model_loader.py:25-72: newest snapshots/*.pt + the experiment's *.ini settings"""
    import os
    import time
    from lemo_amd.vposer import VPoser, load_vposer, make_vposer_weights
    d = tmp_path / 'vposer_v1_0'
    (d / 'snapshots').mkdir(parents=True)
    src = VPoser()
    src.load_state_dict({**src.state_dict(), **{k: torch.from_numpy(v) for k, v in make_vposer_weights(4).items()}})
    torch.save({k: torch.zeros_like(v) for k, v in src.state_dict().items()}, d / 'snapshots' / 'TR00_E001.pt')
    time.sleep(0.02)
    torch.save(src.state_dict(), d / 'snapshots' / 'TR00_E096.pt')
    os.utime(d / 'snapshots' / 'TR00_E001.pt', (1, 1))
    (d / 'TR00_vposer_defaults.ini').write_text('[All]\nnum_neurons = 512\nlatentD = 32\ndata_shape = [1, 21, 3]\nkl_coef = 5e-3\n')
    vp, ps = load_vposer(str(d), vp_model='snapshot')
    assert isinstance(vp, VPoser) and not vp.training
    assert ps.num_neurons == 512 and ps.latentD == 32 and list(ps.data_shape) == [1, 21, 3] and ps.best_model_fname.endswith('TR00_E096.pt')
    assert torch.equal(vp.state_dict()['bodyprior_dec_fc2.weight'], src.state_dict()['bodyprior_dec_fc2.weight'])
    (d / 'TR00_vposer_defaults.ini').write_text('[All]\nnum_neurons = 256\nlatentD = 32\ndata_shape = [1, 21, 3]\n')
    with pytest.raises(NotImplementedError):
        load_vposer(str(d))
    with pytest.raises(ValueError):
        load_vposer(str(tmp_path / 'missing'))


def test_lbs_cache_is_keyed_on_tensor_identity(emu_lib):
    """a cached model must not be served for different tensors that happen to share address / version / shape"""
    from lemo_amd.compat.smplx import lbs as L
    L._CACHE.clear()
    m = synthetic.make_synthetic_smplx(seed=3, V=640, F=1200)
    so = O.SmplxOracle(m)
    sd = torch.cat([so.shapedirs, so.expr_dirs], -1)
    args = (so.v_template, sd, so.posedirs, so.J_regressor, so.parents, so.lbs_weights)
    a = L._model_for(*args, emu_lib)
    assert L._model_for(*args, emu_lib) is a
    key = next(iter(L._CACHE))
    other = tuple(t.clone() for t in args)
    L._CACHE[key] = (a, other)                         # same key, different tensor objects: must rebuild
    assert L._model_for(*args, emu_lib) is not a
