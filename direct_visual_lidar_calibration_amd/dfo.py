"""RESTATEMENT OF AN OUT-OF-SCOPE CALLER (include/dfo/nelder_mead.hpp), kept so that GPU runs can be driven probe for probe like the
reference's host code in tests and in the Python `calibrate` command; it is not a component of the hot path and claims
no coverage of SURVEY section 8.

Host-side derivative-free optimiser with the semantics of the reference's ``dfo::NelderMead<N>``
(include/dfo/nelder_mead.hpp:11-113): simplex from ``x0 + init_step * e_i``, sort, variance-based
convergence test on the coordinates only, centroid is *evaluated* every iteration, reflection /
expansion / outside contraction / shrink with (alpha, gamma, rho) = (1, 2, 0.5).

Optimiser glue stays on the host in the reference too (it is the caller of the hot path, not the
hot path); the objective it drives is the GPU cost.
"""
from dataclasses import dataclass, field

import numpy as np


@dataclass
class NelderMeadParams:  # nelder_mead.hpp:11-23
    init_step: float = 0.1
    alpha: float = 1.0
    gamma: float = 2.0
    rho: float = 0.5
    sigma: float = 0.5
    max_iterations: int = 1024
    convergence_var_thresh: float = 1e-5


@dataclass
class OptimizationResult:  # optimizer.hpp:8-19
    converged: bool = False
    num_iterations: int = 0
    x: np.ndarray = field(default_factory=lambda: np.zeros(0))
    y: float = 0.0
    num_evaluations: int = 0


class NelderMead:
    def __init__(self, params=None):
        self.params = params or NelderMeadParams()
        self.callback = None

    def set_callback(self, f):
        self.callback = f

    @staticmethod
    def _is_converged(xs, thresh):  # nelder_mead.hpp:105-113
        m = np.mean(xs, axis=0)
        var = ((xs - m) ** 2).sum(axis=0)
        return var[1:].sum() < thresh

    def optimize(self, function, x0, batch_function=None):  # nelder_mead.hpp:32-102
        """``batch_function(list of vertices) -> list of values`` (optional) is used where the reference evaluates vertices that
        do not depend on each other -- the n + 1 vertices of the initial simplex (nelder_mead.hpp:37-45) and the n vertices of
        a shrink step (:88-92) -- so that a GPU objective can have them all in flight at once (nidreg_submit / nidreg_wait);
        it must return what ``function`` would, vertex by vertex, in order."""
        p = self.params
        x0 = np.asarray(x0, dtype=np.float64)
        n = x0.shape[0]
        result = OptimizationResult()
        evals = 0

        def fn(v):
            nonlocal evals
            evals += 1
            return float(function(v))

        def fn_many(vs):
            nonlocal evals
            if batch_function is None:
                return [fn(v) for v in vs]
            evals += len(vs)
            return [float(y) for y in batch_function(vs)]

        verts = [x0]
        for i in range(n):
            xi = x0.copy()
            xi[i] += p.init_step
            verts.append(xi)
        x = np.array([np.concatenate([[y], v]) for y, v in zip(fn_many(verts), verts)])

        for it in range(p.max_iterations):
            result.num_iterations = it
            x = x[np.argsort(x[:, 0], kind="stable")]
            if self._is_converged(x, p.convergence_var_thresh):
                result.converged = True
                break
            xo = x[:-1].mean(axis=0)
            xo[0] = fn(xo[1:])
            xr = xo + p.alpha * (xo - x[-1])
            xr[0] = fn(xr[1:])
            if x[0, 0] <= xr[0] and xr[0] < x[n - 1, 0]:
                x[-1] = xr
            elif xr[0] < x[0, 0]:
                xe = xo + p.gamma * (xo - x[-1])
                xe[0] = fn(xe[1:])
                x[-1] = xe if xe[0] < xr[0] else xr
            else:
                xc = xo + p.rho * (xo - x[-1])
                xc[0] = fn(xc[1:])
                if xc[0] < x[-1, 0]:
                    x[-1] = xc
                else:
                    for j in range(1, x.shape[0]):
                        x[j] = x[0] + p.rho * (x[j] - x[0])
                    for j, y in enumerate(fn_many([x[j, 1:].copy() for j in range(1, x.shape[0])]), start=1):
                        x[j, 0] = y
            if self.callback:
                self.callback(x[0, 1:].copy())

        result.x = x[0, 1:].copy()
        result.y = float(x[0, 0])
        result.num_evaluations = evals
        return result
