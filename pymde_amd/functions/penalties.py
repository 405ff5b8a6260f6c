"""Penalties: distortion functions derived from weights, f_k(d_k) = w_k p(d_k).

Same names, constructor arguments and semantics as ``pymde.functions.penalties``
[ref: pymde/functions/penalties.py:112-400].  Positive weights attract, negative weights
repel; ``PushAndPull`` combines an attractive penalty (used where w_k >= 0) with a repulsive
one (w_k < 0).  The formulas live in the HIP kernels (``csrc/mde_functions.h``); a class here
only carries the parameters.

    Linear w d | Quadratic w d^2 | Cubic w d^3 | Power w d^e | Huber | Logistic | Sigmoid |
    Hinge | Log1p w log(1+d^e) | Log w log(1-exp(-d^e)) | InvPower |w|/d^e |
    LogRatio w log(d^e/(1+d^e)) | PushAndPull
"""
import torch

from pymde_amd import util
from pymde_amd.functions.function import Function, HipSpec, KIND, read_scalars


class _Penalty(Function):
    """Weights plus up to three scalars; subclasses name the kind and the scalar attributes."""

    _kind = None
    _scalar_attrs = ()

    def __init__(self, weights):
        super(_Penalty, self).__init__()
        self.weights = util.to_tensor(weights)

    def _scalars(self):
        vals = read_scalars(self, self._scalar_attrs)
        return tuple(vals + [0.0] * (3 - len(vals)))

    def _hip_spec(self):
        return HipSpec(KIND[self._kind], self.weights, None, self._scalars())


def _tensor_exponent(exponent, device=None):
    if not isinstance(exponent, torch.Tensor):
        exponent = torch.tensor(exponent, device=device)
    return exponent


def _check_threshold(threshold):
    if threshold < 0:
        raise ValueError("Threshold must be nonnegative, received ", threshold)


class Linear(_Penalty):
    """p(d) = d"""
    _kind = "LINEAR"


class Quadratic(_Penalty):
    """p(d) = d^2"""
    _kind = "QUADRATIC"


class Cubic(_Penalty):
    """p(d) = d^3"""
    _kind = "CUBIC"


class Power(_Penalty):
    """p(d) = d^exponent"""
    _kind = "POWER"
    _scalar_attrs = ("exponent",)

    def __init__(self, weights, exponent):
        super(Power, self).__init__(weights)
        self.exponent = _tensor_exponent(exponent, self.weights.device)


class Huber(_Penalty):
    """p(d) = d^2/2 for d < threshold, threshold (d - threshold/2) otherwise"""
    _kind = "HUBER"
    _scalar_attrs = ("threshold",)

    def __init__(self, weights, threshold=0.5):
        _check_threshold(threshold)
        super(Huber, self).__init__(weights)
        self.threshold = threshold


class Logistic(_Penalty):
    """p(d) = log(1 + exp(alpha (d - threshold)))"""
    _kind = "LOGISTIC"
    _scalar_attrs = ("threshold", "alpha")

    def __init__(self, weights, threshold=0.0, alpha=3.0):
        _check_threshold(threshold)
        super(Logistic, self).__init__(weights)
        self.threshold = threshold
        self.alpha = alpha


class Sigmoid(_Penalty):
    """p(d) = sigmoid(alpha (d - threshold))"""
    _kind = "SIGMOID"
    _scalar_attrs = ("threshold", "alpha")

    def __init__(self, weights, threshold, alpha=1.0):
        _check_threshold(threshold)
        super(Sigmoid, self).__init__(weights)
        self.threshold = threshold
        self.alpha = alpha


class Hinge(_Penalty):
    """f(d) = max(0, w (d - (threshold - sign(w) sigma)))"""
    _kind = "HINGE"
    _scalar_attrs = ("threshold", "sigma")

    def __init__(self, weights, threshold, sigma=None):
        _check_threshold(threshold)
        super(Hinge, self).__init__(weights)
        self.threshold = threshold
        self.sigma = threshold / 2 if sigma is None else sigma


class Log1p(_Penalty):
    """p(d) = log(1 + d^exponent)"""
    _kind = "LOG1P"
    _scalar_attrs = ("exponent",)

    def __init__(self, weights, exponent=1.5):
        super(Log1p, self).__init__(weights)
        self.exponent = _tensor_exponent(exponent, self.weights.device)


class Log(_Penalty):
    """p(d) = log(1 - exp(-d^exponent))"""
    _kind = "LOG"
    _scalar_attrs = ("exponent",)

    def __init__(self, weights, exponent=1.0):
        super(Log, self).__init__(weights)
        self.exponent = _tensor_exponent(exponent, self.weights.device)


class InvPower(_Penalty):
    """p(d) = 1/d^exponent, for nonpositive weights (|w| is used)"""
    _kind = "INVPOWER"
    _scalar_attrs = ("exponent",)

    def __init__(self, weights, exponent=1):
        weights = util.to_tensor(weights)
        if not bool((weights <= 0).all()):
            raise ValueError("Weights must be negative.")
        super(InvPower, self).__init__(weights)
        self.exponent = _tensor_exponent(exponent)


class LogRatio(_Penalty):
    """p(d) = log(d^exponent / (1 + d^exponent))"""
    _kind = "LOGRATIO"
    _scalar_attrs = ("exponent",)

    def __init__(self, weights, exponent=2):
        super(LogRatio, self).__init__(weights)
        self.exponent = _tensor_exponent(exponent)


class _DeadzoneQuadratic(_Penalty):
    _kind = "DEADZONE_QUADRATIC"
    _scalar_attrs = ("threshold",)

    def __init__(self, weights, threshold):
        super(_DeadzoneQuadratic, self).__init__(weights)
        self.threshold = threshold


class _DeadzoneCubic(_Penalty):
    _kind = "DEADZONE_CUBIC"
    _scalar_attrs = ("threshold",)

    def __init__(self, weights, threshold):
        super(_DeadzoneCubic, self).__init__(weights)
        self.threshold = threshold


class _ClippedQuadratic(_Penalty):
    _kind = "CLIPPED_QUADRATIC"
    _scalar_attrs = ("threshold",)

    def __init__(self, weights, threshold):
        super(_ClippedQuadratic, self).__init__(weights)
        self.threshold = threshold


class PushAndPull(Function):
    """Attractive penalty where w_k >= 0, repulsive penalty where w_k < 0
    [ref: penalties.py:372-400; zero weights are attractive, :390]."""

    def __init__(self, weights, attractive_penalty=Log1p, repulsive_penalty=LogRatio):
        super(PushAndPull, self).__init__()
        weights = util.to_tensor(weights)
        if weights.nelement() == 1:
            raise ValueError("`PushAndPull` requires at least two weights.")
        self.weights = weights
        self.pos_idx = weights >= 0
        self.attractive_penalty = attractive_penalty(weights[self.pos_idx])
        self.repulsive_penalty = repulsive_penalty(weights[~self.pos_idx])

    def _hip_spec(self):
        a, r = self.attractive_penalty, self.repulsive_penalty
        if not (isinstance(a, _Penalty) and isinstance(r, _Penalty)):
            return None
        return HipSpec(KIND[a._kind], self.weights, None, a._scalars(), KIND[r._kind],
                       r._scalars())

    def forward(self, distances):
        if self._hip_spec() is not None:
            return super(PushAndPull, self).forward(distances)
        # arbitrary penalty callables: the reference's masked evaluation (penalties.py:394-400)
        output = torch.zeros(distances.shape, dtype=distances.dtype, device=distances.device)
        pos = self.pos_idx.to(distances.device)
        output[pos] = self.attractive_penalty(distances[pos])
        output[~pos] = self.repulsive_penalty(distances[~pos])
        return output
