"""The synthetic stand-ins for the licensed model files (lemo_amd/synthetic.py)."""
def test_coherent_model_has_the_licensed_models_shape_and_locality():
    """make_synthetic_smplx(coherent=True): same keys / shapes / non-zero bounds as the i.i.d. model (SURVEY
    Appendix B), skinning rows sum to 1 with 1 .. 4 non-zeros, a joint is regressed from <= 32 vertices of its own part, every joint
    owns vertices -- and consecutive indices share joints: a 512-vertex chunk touches far fewer than 55 joints, a 42-vertex block
    of the forward kernel a handful.  The default model is unchanged (the golden fixtures pin it)."""
    import numpy as np
    from lemo_amd import synthetic
    a, b = synthetic.make_synthetic_smplx(0), synthetic.make_synthetic_smplx(0, coherent=True)
    assert set(a) == set(b) and all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)
    for k in ('posedirs', 'shapedirs', 'f', 'kintree_table', 'hands_componentsl'):
        assert np.array_equal(a[k], b[k]), k
    W = b['weights']
    nz = W != 0
    assert nz.sum(1).min() >= 1 and nz.sum(1).max() <= 4 and np.abs(W.sum(1) - 1).max() < 1e-6 and W.min() >= 0
    assert nz.any(0).all()                                                  # every joint moves some vertex
    jr = b['J_regressor']
    assert (jr != 0).sum(1).max() <= 32 and np.abs(jr.sum(1) - 1).max() < 1e-6 and jr.min() >= 0
    dom = W.argmax(1)
    for j in range(55):
        assert (dom[np.nonzero(jr[j])[0]] == j).all(), j                    # regressed from its own part
    per512 = [int(nz[i:i + 512].any(0).sum()) for i in range(0, W.shape[0], 512)]
    per42 = [int(nz[i:i + 42].any(0).sum()) for i in range(0, W.shape[0], 42)]
    assert max(per512) <= 24 and np.mean(per512) < 12 and np.mean(per42) < 7, (per512, np.mean(per42))
    assert [int((a['weights'] != 0)[i:i + 512].any(0).sum()) for i in (0, 5120)] == [55, 55]   # the i.i.d. model: all of them
    small = synthetic.make_synthetic_smplx(5, V=640, F=1200, coherent=True)    # the reduced test size works too
    assert (small['weights'] != 0).any(0).all() and np.abs(small['weights'].sum(1) - 1).max() < 1e-6
