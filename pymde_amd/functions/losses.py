"""Losses: distortion functions derived from original deviations, f_k(d_k) = l(d_k, delta_k).

Same names, constructor arguments and semantics as ``pymde.functions.losses``
[ref: pymde/functions/losses.py:61-239]; a loss is 0 when d_k == delta_k and grows with
|d_k - delta_k|.  The formulas live in the HIP kernels (``csrc/mde_functions.h``).

    Quadratic (delta-d)^2 | WeightedQuadratic w (delta-d)^2 (w = 1/delta^2 by default) |
    Huber | Cubic |delta-d|^3 | Power |delta-d|^e | Absolute | Logistic |
    Fractional max(delta/d, d/delta) - 1 | SoftFractional
"""
import torch

from pymde_amd import util
from pymde_amd.functions.function import Function, HipSpec, KIND, read_scalars


class _Loss(Function):
    _kind = None
    _scalar_attrs = ()
    _weighted = False

    def __init__(self, deviations):
        super(_Loss, self).__init__()
        self.deviations = util.to_tensor(deviations)

    def _scalars(self):
        vals = read_scalars(self, self._scalar_attrs)
        return tuple(vals + [0.0] * (3 - len(vals)))

    def _hip_spec(self):
        a1 = self.weights if self._weighted else None
        return HipSpec(KIND[self._kind], self.deviations, a1, self._scalars())


def _default_weights(deviations, weights):
    if weights is None:
        weights = 1.0 / deviations.pow(2)
    return util.to_tensor(weights, device=deviations.device)


class Quadratic(_Loss):
    """l(d, delta) = (d - delta)^2"""
    _kind = "L_QUADRATIC"


class WeightedQuadratic(_Loss):
    """l(d, delta) = w (d - delta)^2, w = 1/delta^2 unless ``weights`` is given"""
    _kind = "L_WEIGHTED_QUADRATIC"
    _weighted = True

    def __init__(self, deviations, weights=None):
        super(WeightedQuadratic, self).__init__(deviations)
        self.weights = _default_weights(self.deviations, weights)


class _ClippedQuadratic(_Loss):
    _kind = "L_CLIPPED_QUADRATIC"
    _scalar_attrs = ("threshold",)

    def __init__(self, deviations, threshold):
        super(_ClippedQuadratic, self).__init__(deviations)
        self.threshold = threshold


class Huber(_Loss):
    """l = r^2 for r = |d - delta| < threshold, threshold (2 r - threshold) otherwise"""
    _kind = "L_HUBER"
    _scalar_attrs = ("threshold",)

    def __init__(self, deviations, threshold):
        super(Huber, self).__init__(deviations)
        self.threshold = threshold


class Cubic(_Loss):
    """l(d, delta) = |d - delta|^3"""
    _kind = "L_CUBIC"


class Power(_Loss):
    """l(d, delta) = |d - delta|^exponent"""
    _kind = "L_POWER"
    _scalar_attrs = ("exponent",)

    def __init__(self, deviations, exponent):
        super(Power, self).__init__(deviations)
        self.exponent = util.to_tensor(exponent, device=self.deviations.device)


class _WeightedPower(_Loss):
    _kind = "L_WEIGHTED_POWER"
    _scalar_attrs = ("exponent",)
    _weighted = True

    def __init__(self, deviations, exponent, weights=None):
        super(_WeightedPower, self).__init__(deviations)
        self.exponent = util.to_tensor(exponent, device=self.deviations.device)
        self.weights = _default_weights(self.deviations, weights)


class Absolute(_Loss):
    """l(d, delta) = |d - delta|"""
    _kind = "L_ABSOLUTE"


class Logistic(_Loss):
    """l(d, delta) = log(1 + exp(|d - delta|))"""
    _kind = "L_LOGISTIC"


class Fractional(_Loss):
    """l(d, delta) = max(delta/d, d/delta) - 1"""
    _kind = "L_FRACTIONAL"


class SoftFractional(_Loss):
    """l = (1/gamma) log((exp(gamma delta/d) + exp(gamma d/delta)) / (2 exp(gamma)))"""
    _kind = "L_SOFT_FRACTIONAL"
    _scalar_attrs = ("gamma",)

    def __init__(self, deviations, gamma=10.0):
        super(SoftFractional, self).__init__(deviations)
        if gamma <= 0.0:
            raise ValueError("gamma must be positive, received ", float(gamma))
        self.gamma = util.to_tensor(gamma, device=self.deviations.device)


class _Log1p(_Loss):
    """l(d, delta) = log(1 + (d - delta)^exponent)  [ref: losses.py:232-239; private upstream]"""
    _kind = "L_LOG1P"
    _scalar_attrs = ("exponent",)

    def __init__(self, deviations, exponent):
        super(_Log1p, self).__init__(deviations)
        self.exponent = util.to_tensor(exponent, device=self.deviations.device)
