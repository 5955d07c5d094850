"""Teacher-forced one-step parity (shared by the emulator tests and the -m gpu tests).

``tests/golden/teacher_*.npz`` hold optimiser states the REFERENCE's own loops went through (tests/golden/make_teacher.py).
A test loads state k into an engine (C ABI ``lemo_fit_load_state`` / ``lemo_prox_load_state``), runs ONE iteration and
compares with the reference's state k + 1.  Three statements per step, from strict to derived:

1. *Optimiser arithmetic, exact*: from the loaded (p, m, v, k) and the gradient the ENGINE computed, torch's own Adam
   formulas in fp32 (evaluated here on the CPU) must give the engine's (p', m', v') to rounding -- lr level, bias corrections,
   step counter, eps placement.  No gradient noise enters: the bound is a few ulp.
2. *Gradient*: engine vs the reference's fp32 gradient of the same state, frame by frame, against the computed bound of
   tests/test_gpu_gates.py (ROUND + c x S[frame]; S = kink exposure computed in float64 at that state and stored in the
   fixture) or, where no S exists (stage 1, PROX), against a multiple of the reference's own distance from float64.
3. *Next state vs the reference's*: |p' - p'_ref| per entry <= lr x (1e-5 + c1 x E / (sqrt(v_hat') + eps)) with E the gradient
   bound of (2) for that entry's frame and group -- Adam's own formula applied to the gradient bound, nothing else.  Entries
   whose bound exceeds 0.05 x lr are "noise-level" (|g| ~ its own rounding noise; Adam's 1 / sqrt(v) makes their update an O(lr)
   coin toss in ANY fp32 implementation, the reference's included); they are counted and reported, and still bounded by
   2 x lr (an update cannot exceed lr / (1 - beta1^t) x ...).
"""
import numpy as np

B1, B2, EPS = 0.9, 0.999, 1e-8


def adam_reference(p, m, v, g, k, lr):
    """torch.optim.Adam (defaults, no amsgrad / weight decay) for step k -> k + 1 in float32, on numpy arrays, operation for
    operation as torch 2.x's CPU kernels evaluate it (probed bit for bit, tests/test_teacher_emu.py):
    exp_avg.lerp_(g, 1 - b1) = fma(g - m, 0.1f, m); exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2) = fma(0.001f g, g, v 0.999f);
    denom = sqrt(v) / sqrt(bias2) + eps; p += ((-lr / bias1) m) / denom (bias corrections in Python doubles, rounded once).
    float64 holds an fp32 product exactly, so fma(a, b, c) = float32(float64(a) * float64(b) + float64(c))."""
    f, d = np.float32, np.float64
    p, m, v, g = (np.asarray(a, f) for a in (p, m, v, g))
    t = k + 1
    m2 = ((g - m).astype(f).astype(d) * d(f(0.1)) + m.astype(d)).astype(f)
    v2 = ((f(0.001) * g).astype(f).astype(d) * g.astype(d) + (v * f(0.999)).astype(f).astype(d)).astype(f)
    bc1, bc2 = 1 - B1 ** t, 1 - B2 ** t
    denom = ((np.sqrt(v2) / f(np.sqrt(bc2))).astype(f) + f(EPS)).astype(f)
    p2 = (p + ((f(-(lr / bc1)) * m2).astype(f) / denom).astype(f)).astype(f)
    return p2, m2, v2


def update_bound(E, v_next_ref, k, lr):
    """per-entry bound on |p' - p'_ref| when the two gradients differ by at most E (same shape as v): first-order in E
    through m' = b1 m + (1 - b1) g and v' = b2 v + (1 - b2) g^2; the v-path is bounded by the same term (|d sqrt(v)| <=
    (1 - b2) |g| E / sqrt(v) and |m_hat| <= ~sqrt(v_hat) / ...), so a factor 2 covers both."""
    t = k + 1
    bc1, bc2 = 1 - B1 ** t, 1 - B2 ** t
    vhat = np.sqrt(np.asarray(v_next_ref, np.float64) / bc2) + EPS
    return lr * (1e-5 + 2.0 * (1 - B1) / bc1 * np.asarray(E, np.float64) / vhat)


def check_adam_arithmetic(tag, before, g_engine, after_engine, k, lr):
    """statement 1.  `before` / `after_engine`: dicts with p, m, v [B, D] arrays.  exp_avg / exp_avg_sq must be torch's BIT FOR
    BIT (same fused multiply-adds), the parameter within one ulp of p (+ one ulp of the update: the division and the final add
    are the only operations whose rounding may differ).  Returns the number of parameter entries that are not bit-identical."""
    p2, m2, v2 = adam_reference(before['p'], before['m'], before['v'], g_engine, k, lr)
    for name, want, got in (('m', m2, after_engine['m']), ('v', v2, after_engine['v'])):
        bad = want != np.asarray(got, np.float32)
        assert not bad.any(), (tag, name, int(bad.sum()), np.argwhere(bad)[:4].tolist())
    got = np.asarray(after_engine['p'], np.float64)
    dev = np.abs(p2.astype(np.float64) - got)
    tol = 1.2e-7 * np.maximum(np.abs(p2), np.abs(np.asarray(before['p'], np.float32))).astype(np.float64) + 2.4e-7 * lr / (1 - B1 ** (k + 1))
    bad = dev > tol
    assert not bad.any(), (tag, 'p', int(bad.sum()), float(dev.max()), np.argwhere(bad)[:4].tolist())
    return int((dev > 0).sum())


def check_next_state(tag, p_engine, p_ref, E, v_next_ref, k, lr, report=None):
    """statement 3.  Returns (max |dp| / lr over regular entries, number of noise-level entries)."""
    bound = update_bound(E, v_next_ref, k, lr)
    noise = bound > 0.05 * lr
    dp = np.abs(np.asarray(p_engine, np.float64) - np.asarray(p_ref, np.float64))
    bad = (dp > bound) & ~noise
    assert not bad.any(), (tag, int(bad.sum()), np.argwhere(bad)[:6].tolist(), [float(x) for x in dp[bad][:6]], [float(x) for x in bound[bad][:6]])
    assert not (dp[noise] > 2.2 * lr / (1 - B1 ** (k + 1)) * 1.0).any(), (tag, 'noise-level entry moved by more than Adam can move it')
    reg = float((dp[~noise] / lr).max()) if (~noise).any() else 0.0
    if report is not None:
        report.append(f'{tag}: next-state max |dp|/lr {reg:.2e} over {int((~noise).sum())} regular entries '
                      f'(bound median {float(np.median(bound[~noise]) / lr) if (~noise).any() else 0:.1e} lr), {int(noise.sum())} noise-level entries'
                      f'{" (max |dp|/lr %.2e)" % float((dp[noise] / lr).max()) if noise.any() else ""}')
    return reg, int(noise.sum())
