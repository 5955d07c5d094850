"""CPU restatement of LEMO's marker-image decode / encode around the fitting loop.

TEST INFRASTRUCTURE ONLY: imported by tests/, tests/golden/make_golden.py and nothing else.  The product path
(lemo_amd/markers.py -> liblemo_hip.so) never touches this module.

Pinned against the reference itself: tests/golden/make_golden.py imports /root/reference/utils/utils.py in the
build container and checks both functions below against ``reconstruct_global_body`` / ``get_local_markers_4chan``
on seeded inputs (tests/golden/oracle_vs_reference.txt), then emits tests/golden/markers_decode.npz.

Reference: utils/utils.py:184-203 (reconstruct_global_body), :209-265 (get_local_markers_4chan),
utils/Quaternions.py:71-118, 138-140, 396-407 (product, inverse, between, from_angle_axis),
utils/Pivots.py:79-89 (from_quaternions), scipy.ndimage.gaussian_filter1d(sigma=20, mode='nearest').
Everything runs in float64 like the numpy reference.  Inputs are not modified (the reference works in place).
"""
from __future__ import annotations

import numpy as np

FILTER_SIGMA = 20            # utils/utils.py:235 direction_filterwidth
FILTER_TRUNCATE = 4.0        # scipy default -> radius 80


def _qmul(q, r):
    """Quaternions.__mul__ for quaternion operands (utils/Quaternions.py:93-108): q * r."""
    q0, q1, q2, q3 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    r0, r1, r2, r3 = r[..., 0], r[..., 1], r[..., 2], r[..., 3]
    return np.stack([r0 * q0 - r1 * q1 - r2 * q2 - r3 * q3,
                     r0 * q1 + r1 * q0 - r2 * q3 + r3 * q2,
                     r0 * q2 + r1 * q3 + r2 * q0 - r3 * q1,
                     r0 * q3 - r1 * q2 + r2 * q1 + r3 * q0], axis=-1)


def _qinv(q):
    return q * np.array([1.0, -1.0, -1.0, -1.0])


def _qrot(q, v):
    """Quaternions * vectors (utils/Quaternions.py:110-113): imag(q * ((0, v) * -q))."""
    vs = np.concatenate([np.zeros(v.shape[:-1] + (1,)), v], axis=-1)
    return _qmul(q, _qmul(vs, _qinv(q)))[..., 1:]


def _q_y(angle):
    """Quaternions.from_angle_axis(angle, [0, 1, 0]) (utils/Quaternions.py:402-407)."""
    axis = np.array([0.0, 1.0, 0.0]) / (1.0 + 1e-10)
    a = np.asarray(angle, np.float64)
    return np.concatenate([np.cos(a / 2.0)[..., None], axis * np.sin(a / 2.0)[..., None]], axis=-1)


def _pivot(q):
    """Pivots.from_quaternions(q).ps (utils/Pivots.py:79-89): angle of q * (0,0,1) in the xz plane."""
    d = _qrot(q, np.broadcast_to(np.array([0.0, 0.0, 1.0]), q.shape[:-1] + (3,)))
    return np.arctan2(d[..., 0], d[..., 2])


def gaussian_filter1d_nearest(x, sigma=FILTER_SIGMA, truncate=FILTER_TRUNCATE):
    """scipy.ndimage.gaussian_filter1d(x, sigma, axis=0, mode='nearest'): normalised taps, clamped indices."""
    radius = int(truncate * float(sigma) + 0.5)
    k = np.arange(-radius, radius + 1)
    w = np.exp(-0.5 * (k / float(sigma)) ** 2)
    w /= w.sum()
    T = x.shape[0]
    out = np.zeros_like(x, dtype=np.float64)
    for i, kk in enumerate(k):
        idx = np.clip(np.arange(T) + kk, 0, T - 1)
        out += w[i] * x[idx]
    return out


def reconstruct_global_body(body_joints_input, rot_0_pivot):
    """[T, 1+J+1, 3] (reference slot, J local joints, trajectory (dx, dz, dr)) -> [T, J, 3] global positions.
    utils/utils.py:184-203.  Every rotation is about +y, so the running quaternion is a running angle."""
    b = np.array(body_joints_input, dtype=np.float64, copy=True)
    traj = b[:, -1]
    root_r, root_x, root_z = traj[:, 2], traj[:, 0], traj[:, 1]
    b = b[:, :-1]
    b[:, :, [1, 2]] = b[:, :, [2, 1]]
    rotation = np.array([[1.0, 0.0, 0.0, 0.0]])
    translation = np.zeros((1, 3))
    for i in range(len(b)):
        if i == 0:
            rotation = _qmul(_q_y(-np.asarray(rot_0_pivot, np.float64).reshape(-1)[:1]), rotation)
        b[i] = _qrot(rotation, b[i])
        b[i, :, 0] += translation[0, 0]
        b[i, :, 2] += translation[0, 2]
        rotation = _qmul(_q_y(-root_r[i:i + 1]), rotation)
        translation = translation + _qrot(rotation, np.array([[root_x[i], 0.0, root_z[i]]]))
    b[:, :, [1, 2]] = b[:, :, [2, 1]]
    return b[:, 1:, :]


def get_local_markers_4chan(cur_body, contact_lbls):
    """[T, 1+67, 3] global pelvis + markers, [T, 4] contact labels -> ([4, T-1, 3*68+4], rot_0_pivot [1]).
    utils/utils.py:209-265."""
    c = np.array(cur_body, dtype=np.float64, copy=True)
    c[:, :, [1, 2]] = c[:, :, [2, 1]]
    c[:, :, 1] = c[:, :, 1] - c[:, :, 1].min()
    reference = c[:, 0] * np.array([1.0, 0.0, 1.0])
    c = np.concatenate([reference[:, None], c], axis=1)
    velocity = (c[1:, 0:1] - c[:-1, 0:1]).copy()
    c[:, :, 0] = c[:, :, 0] - c[:, 0:1, 0]
    c[:, :, 2] = c[:, :, 2] - c[:, 0:1, 2]
    sdr_l, sdr_r, hip_l, hip_r = 28, 58, 29, 59
    across = (c[:, sdr_r] - c[:, sdr_l]) + (c[:, hip_r] - c[:, hip_l])
    across = across / np.sqrt((across ** 2).sum(axis=-1))[..., None]
    forward = np.cross(across, np.array([[0.0, 1.0, 0.0]]))
    forward = gaussian_filter1d_nearest(forward)
    forward = forward / np.sqrt((forward ** 2).sum(axis=-1))[..., None]
    target = np.array([[0.0, 0.0, 1.0]]).repeat(len(forward), axis=0)
    a = np.cross(forward, target)                                       # Quaternions.between (:396-399)
    w = np.sqrt((forward ** 2).sum(-1) * (target ** 2).sum(-1)) + (forward * target).sum(-1)
    rot = np.concatenate([w[..., None], a], axis=-1)
    rot = rot / np.sqrt((rot ** 2).sum(-1))[..., None]
    rot = rot[:, None]                                                  # [T, 1, 4]
    c = _qrot(np.broadcast_to(rot, c.shape[:-1] + (4,)), c)
    velocity = _qrot(rot[1:], velocity)
    rvelocity = _pivot(_qmul(rot[1:], _qinv(rot[:-1])))                 # [T-1, 1]
    rot_0_pivot = _pivot(rot[0])                                        # [1]
    c[:, :, [1, 2]] = c[:, :, [2, 1]]
    c = c[:-1, 1:, :]
    c = c.reshape(len(c), -1)
    local = np.concatenate([c, np.asarray(contact_lbls, np.float64)[:-1]], axis=-1)[None]
    T, d = local.shape[1], local.shape[-1]
    gx = np.repeat(velocity[:, :, 0], d).reshape(1, T, d)
    gy = np.repeat(velocity[:, :, 2], d).reshape(1, T, d)
    gr = np.repeat(rvelocity, d).reshape(1, T, d)
    return np.concatenate([local, gx, gy, gr], axis=0), rot_0_pivot
