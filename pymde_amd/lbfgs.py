"""Host logic of the projected L-BFGS method: strong-Wolfe line search and the two-loop
recursion in coefficient form.

Restates the algorithm of the reference's ``pymde/lbfgs.py`` (a fork of torch.optim.LBFGS)
[ref: lbfgs.py:16-41 cubic interpolation, :44-253 strong Wolfe, :461-507 direction] as pure
scalar code: every n*d-sized vector stays on the GPU (``csrc/mde_vec.hip``) and the host only
sees the handful of inner products per step, so this module has no torch / GPU dependency
and is unit-tested on the CPU.

Direction in coefficient form.  With pairs (s_i, y_i), i = 0..m-1 oldest first,
rho_i = 1/(y_i.s_i), H = (y.s)/(y.y) of the newest pair, the two-loop recursion
(lbfgs.py:490-507)
    q = -g;  for i = m-1..0: a_i = rho_i s_i.q;  q -= a_i y_i
    r = H q; for i = 0..m-1: b_i = rho_i y_i.r;  r += (a_i - b_i) s_i
only ever forms vectors in span{g, s_j, y_j}; tracking the coefficients needs just the Gram
entries s_i.y_j, y_i.y_j and s_i.g, y_i.g, which the device computes in one fused pass.
"""
import math

import numpy as np


class LineSearchError(Exception):
    """Raised when the objective or gradient is not finite at the start of a line search."""


def cubic_interpolate(x1, f1, g1, x2, f2, g2, bounds=None):
    """Minimiser of the cubic through (x1, f1, g1), (x2, f2, g2), clipped to ``bounds``
    (default: the interval between the points); bisection if the cubic has no minimiser."""
    if bounds is not None:
        lo, hi = bounds
    else:
        lo, hi = (x1, x2) if x1 <= x2 else (x2, x1)
    d1 = g1 + g2 - 3.0 * (f1 - f2) / (x1 - x2)
    disc = d1 * d1 - g1 * g2
    if disc >= 0:
        d2 = math.sqrt(disc)
        if x1 <= x2:
            pos = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2.0 * d2))
        else:
            pos = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2.0 * d2))
        return min(max(pos, lo), hi)
    return (lo + hi) / 2.0


class _Point(object):
    __slots__ = ("t", "f", "gtd")

    def __init__(self, t, f, gtd):
        self.t, self.f, self.gtd = t, f, gtd


def _bad(x):
    return math.isnan(x) or math.isinf(x)


def strong_wolfe(phi, t, f0, gtd0, d_norm, c1=1e-4, c2=0.9, tolerance_change=1e-9, max_ls=25):
    """Strong-Wolfe line search along a fixed direction.

    ``phi(t) -> (f, gtd, grad_finite)`` evaluates the objective at the (retracted) trial point
    x + t d, its directional derivative g.d and whether the gradient is finite.  ``f0``,
    ``gtd0`` are the values at t = 0, ``d_norm = max|d|``.

    Returns ``(f_t, t, n_evals)``.  Semantics follow lbfgs.py:44-253: initial halving while
    the trial is not finite (<= 10 times, then LineSearchError), bracketing with cubic
    extrapolation, zoom with the 10 %-of-bracket safeguard, and the 0.8-backtracking fallback
    that ends at t = 0 when even Armijo cannot be met.
    """
    f_new = gtd_new = None
    finite = True
    for _ in range(10):
        f_new, gtd_new, finite = phi(t)
        if _bad(f_new) or not finite:
            t *= 0.5
        else:
            break
    if math.isnan(f_new):
        raise LineSearchError("Function evaluation returned NaN.")
    if math.isinf(f_new):
        raise LineSearchError("Function evaluation returned inf.")
    if not finite:
        raise LineSearchError("Gradient evaluation returned NaN or inf.")
    n_evals = 1

    prev = _Point(0.0, f0, gtd0)
    bracket = None
    done = False
    ls_iter = 0
    while ls_iter < max_ls:
        cur = _Point(t, f_new, gtd_new)
        if f_new > (f0 + c1 * t * gtd0) or (ls_iter > 1 and f_new >= prev.f):
            bracket = [prev, cur]
            break
        if abs(gtd_new) <= -c2 * gtd0:
            bracket = [cur]
            done = True
            break
        if gtd_new >= 0:
            bracket = [prev, cur]
            break
        # extrapolate
        min_step = t + 0.01 * (t - prev.t)
        max_step = t * 10
        t_next = cubic_interpolate(prev.t, prev.f, prev.gtd, t, f_new, gtd_new,
                                   bounds=(min_step, max_step))
        prev = cur
        t = t_next
        f_new, gtd_new, _ = phi(t)
        n_evals += 1
        ls_iter += 1

    if ls_iter == max_ls:
        bracket = [_Point(0.0, f0, gtd0), _Point(t, f_new, gtd_new)]

    # zoom
    insufficient = False
    low, high = (0, 1) if bracket[0].f <= bracket[-1].f else (1, 0)
    while not done and ls_iter < max_ls:
        a, b = bracket
        if abs(b.t - a.t) * d_norm < tolerance_change:
            break
        t = cubic_interpolate(a.t, a.f, a.gtd, b.t, b.f, b.gtd)
        hi_t, lo_t = max(a.t, b.t), min(a.t, b.t)
        eps = 0.1 * (hi_t - lo_t)
        if min(hi_t - t, t - lo_t) < eps:
            if insufficient or t >= hi_t or t <= lo_t:
                t = hi_t - eps if abs(t - hi_t) < abs(t - lo_t) else lo_t + eps
                insufficient = False
            else:
                insufficient = True
        else:
            insufficient = False

        f_new, gtd_new, _ = phi(t)
        n_evals += 1
        ls_iter += 1

        if math.isnan(f_new) or f_new > (f0 + c1 * t * gtd0) or f_new >= bracket[low].f:
            bracket[high] = _Point(t, f_new, gtd_new)
            low, high = (0, 1) if bracket[0].f <= bracket[1].f else (1, 0)
        else:
            if abs(gtd_new) <= -c2 * gtd0:
                done = True
            elif gtd_new * (bracket[high].t - bracket[low].t) >= 0:
                bracket[high] = bracket[low]
            bracket[low] = _Point(t, f_new, gtd_new)

    failed = math.isnan(f_new)
    if low < len(bracket):
        t, f_new = bracket[low].t, bracket[low].f
    else:
        t, failed = 1.0, True

    if failed:
        while t > 1e-8:
            t *= 0.8
            f_new, gtd_new, _ = phi(t)  # (the reference does not count the back-off evaluations)
            if math.isnan(f_new):
                continue
            if f_new < f0 + c1 * t * gtd0:
                break
    if math.isnan(f_new):
        t = 0.0
        f_new, gtd_new, _ = phi(t)
    return f_new, t, n_evals


class LbfgsMemory(object):
    """Host mirror of the device history: Gram matrices of the stored pairs (oldest first).

    The device object (``mde_lbfgs``) holds the vectors; ``stage`` hands back the inner
    products listed in include/mde_hip.h, which ``absorb`` folds in here.
    """

    def __init__(self, history_size):
        self.m = int(history_size)
        self.reset()

    def reset(self):
        self.count = 0
        self.SY = np.zeros((0, 0))  # SY[i, j] = s_i . y_j
        self.YY = np.zeros((0, 0))  # YY[i, j] = y_i . y_j
        self.H_diag = 1.0

    def absorb(self, dots, ys_threshold=1e-10):
        """Process the dots of a staged pair.  Returns ``(accepted, Sg, Yg)`` where Sg/Yg are
        s_j.g / y_j.g for the pairs stored AFTER the decision (oldest first)."""
        c = self.count
        ys, yy, sg_new, yg_new = (float(x) for x in dots[:4])
        per = np.asarray(dots[4:4 + 5 * c], dtype=np.float64).reshape(c, 5)
        s_old_ynew, y_old_ynew, snew_y_old = per[:, 0], per[:, 1], per[:, 2]
        Sg, Yg = per[:, 3].copy(), per[:, 4].copy()
        accepted = ys > ys_threshold  # lbfgs.py:472
        if accepted:
            SY = np.zeros((c + 1, c + 1))
            YY = np.zeros((c + 1, c + 1))
            SY[:c, :c], YY[:c, :c] = self.SY, self.YY
            SY[:c, c] = s_old_ynew
            SY[c, :c] = snew_y_old
            SY[c, c] = ys
            YY[:c, c] = y_old_ynew
            YY[c, :c] = y_old_ynew
            YY[c, c] = yy
            Sg = np.append(Sg, sg_new)
            Yg = np.append(Yg, yg_new)
            if c == self.m:  # drop the oldest pair (lbfgs.py:474-478)
                SY, YY, Sg, Yg = SY[1:, 1:], YY[1:, 1:], Sg[1:], Yg[1:]
            self.SY, self.YY = SY, YY
            self.count = SY.shape[0]
            self.H_diag = ys / yy  # lbfgs.py:486
        return accepted, Sg, Yg

    def direction_coefficients(self, Sg, Yg):
        """Coefficients (c_g, cs[m], cy[m]) of d = c_g g + sum_j cs_j s_j + cy_j y_j."""
        m = self.count
        H = self.H_diag
        if m == 0:
            return -H, np.zeros(0), np.zeros(0)
        SY, YY = self.SY, self.YY
        rho = 1.0 / np.diag(SY)
        cq = np.zeros(m)  # q = -g + sum_j cq_j y_j
        al = np.zeros(m)
        for i in range(m - 1, -1, -1):
            sq = -Sg[i] + float(np.dot(cq, SY[i, :]))
            al[i] = rho[i] * sq
            cq[i] = -al[i]
        cr = np.zeros(m)  # r = H q + sum_j cr_j s_j
        for i in range(m):
            yq = -Yg[i] + float(np.dot(cq, YY[i, :]))
            yr = H * yq + float(np.dot(cr, SY[:, i]))
            cr[i] = al[i] - rho[i] * yr
        return -H, cr, H * cq
