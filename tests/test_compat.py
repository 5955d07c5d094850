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


def _licensed_layout(m, rng):
    """the synthetic model re-expressed in the LICENSED file's layout (SMPLX_<GENDER>.npz / .pkl of smplx 0.1.26): 400-wide
    ``shapedirs`` (300 shape + 100 expression directions: betas read [:10], expression [300:310]), uint32 ``kintree_table`` with
    2^32 - 1 at the root, uint32 faces, and the keys the loader must ignore"""
    sd = np.asarray(m['shapedirs'], np.float32)
    wide = (rng.standard_normal(sd.shape[:2] + (400,)) * 0.01).astype(np.float32)      # directions LEMO never reads: non-zero on purpose
    wide[:, :, :10] = sd[:, :, :10]
    wide[:, :, 300:310] = sd[:, :, 10:20]
    out = dict(m)
    out['shapedirs'] = wide
    out['kintree_table'] = np.asarray(m['kintree_table'], np.int64).astype(np.uint32)
    assert out['kintree_table'][0, 0] == 2 ** 32 - 1
    out['f'] = np.asarray(m['f']).astype(np.uint32)
    out['dynamic_lmk_faces_idx'] = rng.integers(0, m['f'].shape[0], size=(79, 17)).astype(np.int64)
    out['dynamic_lmk_bary_coords'] = rng.random((79, 17, 3)).astype(np.float32)
    out['vt'] = rng.random((11313, 2)).astype(np.float32)
    out['ft'] = rng.integers(0, 11313, size=m['f'].shape).astype(np.uint32)
    out['joint2num'] = np.array({'Pelvis': 0}, dtype=object)
    out['part2num'] = np.array({'Global': 0}, dtype=object)
    out['hands_coeffsl'] = rng.standard_normal((8, 45)).astype(np.float32)
    out['hands_coeffsr'] = rng.standard_normal((8, 45)).astype(np.float32)
    return out


def test_smplx_create_from_files_in_the_licensed_layout(emu_lib, tmp_path):
    """``smplx.create(<dir>, model_type='smplx', gender=, ext='npz', num_pca_comps=12, batch_size=B, **extra)`` exactly as
    opt_amass_temp.py:73-87 and temp_prox/main_slide.py:160-179 call it (VERDICT r05 missing #4: every other test hands create()
    a dict).  The licensed assets are not available; a synthetic model written in their layout is: <dir>/smplx/SMPLX_MALE.npz
    (np.load route, object arrays inside) and SMPLX_FEMALE.pkl (pickle route with a scipy-sparse J_regressor, as the .pkl ships).
    Vertices / joints / full_pose must equal the dict route's bit for bit (the directions LEMO does not read are random, not zero)."""
    import pickle
    import scipy.sparse as sp
    import lemo_amd.compat as compat
    m = synthetic.make_synthetic_smplx(seed=5)                       # V = 10475, F = 20908: the default extra-joint vertex ids are SMPL-X's
    rng = np.random.default_rng(17)
    lic = _licensed_layout(m, rng)
    (tmp_path / 'smplx').mkdir()
    np.savez(tmp_path / 'smplx' / 'SMPLX_MALE.npz', **lic)
    pk = dict(lic)
    pk['J_regressor'] = sp.csc_matrix(np.asarray(lic['J_regressor'], np.float64))
    with open(tmp_path / 'smplx' / 'SMPLX_FEMALE.pkl', 'wb') as f:
        pickle.dump(pk, f, protocol=2)
    saved = {k: sys.modules.get(k) for k in ('smplx', 'smplx.lbs', 'chamfer')}
    try:
        for k in saved:
            sys.modules.pop(k, None)
        compat.install()
        import smplx
        B = 3
        # opt_amass_temp.py:73-80, verbatim keyword set (+ the test-only library handle)
        male = smplx.create(str(tmp_path), model_type='smplx', gender='male', ext='npz', num_pca_comps=12,
                            create_global_orient=True, create_body_pose=True, create_betas=True, create_left_hand_pose=True,
                            create_right_hand_pose=True, create_expression=True, create_jaw_pose=True, create_leye_pose=True,
                            create_reye_pose=True, create_transl=True, batch_size=B, _lib=emu_lib)
        # main_slide.py:160-179: model_path as a keyword, the WHOLE parsed-arguments dict behind it (unknown keys must be swallowed)
        args = dict(model_folder=str(tmp_path), model_type='smplx', use_vposer=True, ext='pkl', num_pca_comps=12, batch_size=B,
                    focal_length_x=1060.53, data_folder='x', output_folder='y', rho_contact=0.05, use_cuda=True, optim_type='adam')
        female = smplx.create(gender='female', model_path=args.get('model_folder'), joint_mapper=None, create_global_orient=True,
                              create_body_pose=not args.get('use_vposer'), create_betas=True, create_left_hand_pose=True,
                              create_right_hand_pose=True, create_expression=True, create_jaw_pose=True, create_leye_pose=True,
                              create_reye_pose=True, create_transl=True, dtype=torch.float32, _lib=emu_lib, **args)
        with pytest.raises(FileNotFoundError):
            smplx.create(str(tmp_path), model_type='smplx', gender='neutral', ext='npz', batch_size=B, _lib=emu_lib)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    from lemo_amd.body_model import create
    ref = create(m, model_type='smplx', gender='male', num_pca_comps=12, batch_size=B, _lib=emu_lib)
    assert male.get_num_verts() == 10475 and tuple(male.faces_tensor.shape) == (20908, 3)
    with pytest.raises(ValueError):                                  # SMPL-X's extra-joint ids on a smaller model: refused, not read out of bounds
        create(synthetic.make_synthetic_smplx(seed=5, V=640, F=1200), batch_size=B, _lib=emu_lib)
    assert not hasattr(female, 'body_pose') and hasattr(male, 'body_pose')          # create_body_pose = not use_vposer
    assert {n for n, _ in male.named_parameters()} == {'betas', 'global_orient', 'body_pose', 'left_hand_pose', 'right_hand_pose',
                                                      'jaw_pose', 'leye_pose', 'reye_pose', 'expression', 'transl'}
    g = torch.Generator().manual_seed(2)
    p = dict(betas=torch.randn(B, 10, generator=g) * 0.5, global_orient=torch.randn(B, 3, generator=g) * 0.3,
             body_pose=torch.randn(B, 63, generator=g) * 0.2, left_hand_pose=torch.randn(B, 12, generator=g) * 0.1,
             right_hand_pose=torch.randn(B, 12, generator=g) * 0.1, expression=torch.randn(B, 10, generator=g) * 0.5,
             jaw_pose=torch.randn(B, 3, generator=g) * 0.05, transl=torch.randn(B, 3, generator=g))
    want = ref(return_verts=True, return_full_pose=True, **p)
    assert male(return_verts=True, return_full_pose=False, **p).full_pose is None          # smplx: only on request
    for mod in (male, female):
        got = mod(return_verts=True, return_full_pose=True, **p)
        assert torch.equal(got.vertices, want.vertices), float((got.vertices - want.vertices).abs().max())
        assert torch.equal(got.joints, want.joints), (mod.gender, float((got.joints - want.joints).abs().max()))
        assert torch.equal(got.full_pose, want.full_pose) and got.joints.shape[1] == 127
    # and it is the oracle's SMPL-X (pinned to the reference's vendored lbs.py), expression directions included
    so = O.SmplxOracle(m)
    ov, oj = so.forward(p['betas'], p['global_orient'], p['body_pose'], p['left_hand_pose'], p['right_hand_pose'], transl=p['transl'],
                        expression=p['expression'], jaw_pose=p['jaw_pose'])[:2]
    assert rel_err(want.vertices.detach(), ov.detach()) < 1e-4 and rel_err(want.joints.detach(), oj.detach()) < 1e-4
