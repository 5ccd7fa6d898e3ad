"""RESTATEMENT OF AN OUT-OF-SCOPE CALLER (src/vlcal/calib/visual_camera_calibration.cpp), kept so that GPU runs can be driven probe for probe like the
reference's host code in tests and in the Python `calibrate` command; it is not a component of the hot path and claims
no coverage of SURVEY section 8.

Host-side calibration driver with the structure of the reference's ``VisualCameraCalibration``
(src/vlcal/calib/visual_camera_calibration.cpp:35-68 outer loop, :70-139 Nelder-Mead inner solve,
:190-238 BFGS inner solve).  This is the *caller* of the hot path; the reference keeps it on the
host (Ceres / dfo) and so do we.  Ceres itself is not available offline, so the BFGS + Wolfe line
search below follows Ceres' documented ``GradientProblemSolver`` defaults (LBFGS/BFGS direction,
Wolfe line search with cubic interpolation, function_tolerance 1e-6, gradient_tolerance 1e-10,
parameter_tolerance 1e-8, max 50 iterations, sufficient decrease 1e-4, curvature 0.9) on the
``T * exp(delta)`` manifold (``Sophus::Manifold<SE3>``).

The cost objects are pluggable factories, so the same driver runs on the GPU engine (product) and on
the CPU oracle (parity tests compare the two final poses; BASELINE.json tolerance 1e-3 m / 1e-3 rad).
"""
import math
from dataclasses import dataclass

import numpy as np

from . import se3
from .dfo import NelderMead, NelderMeadParams


@dataclass
class VisualCameraCalibrationParams:  # visual_camera_calibration.hpp:10-41
    max_outer_iterations: int = 10
    max_inner_iterations: int = 256
    delta_trans_thresh: float = 0.1
    delta_rot_thresh: float = 0.5 * math.pi / 180.0
    disable_z_buffer_culling: bool = False
    nid_bins: int = 16
    registration_type: str = "nid_bfgs"  # or "nid_nelder_mead"
    nelder_mead_init_step: float = 1e-3
    nelder_mead_convergence_criteria: float = 1e-8
    bfgs_max_iterations: int = 50


def _cubic_min(a, fa, ga, b, fb, gb):
    """Minimiser of the cubic interpolating (a, fa, ga), (b, fb, gb); falls back to bisection."""
    d1 = ga + gb - 3.0 * (fa - fb) / (a - b)
    rad = d1 * d1 - ga * gb
    if rad < 0 or a == b:
        return 0.5 * (a + b)
    d2 = math.copysign(math.sqrt(rad), b - a)
    den = gb - ga + 2.0 * d2
    if den == 0:
        return 0.5 * (a + b)
    t = b - (b - a) * (gb + d2 - d1) / den
    lo, hi = min(a, b), max(a, b)
    if not (lo + 0.05 * (hi - lo) <= t <= hi - 0.05 * (hi - lo)):
        return 0.5 * (a + b)
    return t


def bfgs_minimize(cost, x0, max_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, callback=None):
    """Minimise ``cost(x7, want_grad) -> (ok, f, grad7)`` over SE(3) with BFGS + strong-Wolfe line
    search on the right-perturbation manifold.  ``ok == False`` (trust gate / non-finite NID) is an
    invalid step: the line search contracts, exactly what Ceres does with a functor returning false."""
    x = np.asarray(x0, dtype=np.float64).copy()
    evals = 0

    def f_and_g(xx):
        nonlocal evals
        evals += 1
        ok, c, g = cost(xx, True)
        if not ok or not np.isfinite(c):
            return False, math.inf, None
        return True, c, se3.plus_jacobian(xx).T @ np.asarray(g)

    ok, f, g = f_and_g(x)
    if not ok:
        raise RuntimeError("initial pose is not a valid evaluation point")
    H = np.eye(6)
    summary = dict(iterations=0, evaluations=0, initial_cost=f, termination="max_iterations")
    for it in range(max_iterations):
        if np.abs(g).max() <= gradient_tolerance:
            summary["termination"] = "gradient_tolerance"
            break
        d = -H @ g
        dg = float(d @ g)
        if dg >= 0:  # not a descent direction: reset
            H = np.eye(6)
            d = -g
            dg = float(d @ g)
        # ---- strong Wolfe line search along x * exp(alpha d)
        c1, c2 = 1e-4, 0.9
        alpha = min(1.0, 1.0 / max(np.abs(g).max(), 1e-12)) if it == 0 else 1.0
        a_lo, f_lo, dg_lo = 0.0, f, dg
        a_hi = f_hi = dg_hi = None
        best = None
        for _ in range(20):
            xt = se3.plus(x, alpha * d)
            okt, ft, gt = f_and_g(xt)
            if not okt:
                a_hi, f_hi, dg_hi = alpha, math.inf, None
                alpha = 0.5 * (a_lo + alpha)
                continue
            dgt = float(d @ gt)
            if ft > f + c1 * alpha * dg or (a_hi is None and ft >= f_lo and a_lo > 0):
                a_hi, f_hi, dg_hi = alpha, ft, dgt
            else:
                if abs(dgt) <= -c2 * dg:
                    best = (alpha, xt, ft, gt)
                    break
                if a_hi is not None and dgt * (a_hi - a_lo) >= 0:
                    a_hi, f_hi, dg_hi = a_lo, f_lo, dg_lo
                a_lo, f_lo, dg_lo = alpha, ft, dgt
                best = (alpha, xt, ft, gt)
                if a_hi is None:
                    alpha = min(alpha * 10.0, alpha + 10.0)  # expansion (max_step_expansion 10)
                    continue
            if a_hi is not None:
                if math.isfinite(f_hi) and dg_hi is not None:
                    alpha = _cubic_min(a_lo, f_lo, dg_lo, a_hi, f_hi, dg_hi)
                else:
                    alpha = 0.5 * (a_lo + a_hi)
                if abs(a_hi - a_lo) < 1e-12:
                    break
        if best is None:
            summary["termination"] = "line_search_failed"
            break
        alpha, x_new, f_new, g_new = best
        s = alpha * d
        y = g_new - g
        sy = float(s @ y)
        if sy > 1e-12 * np.linalg.norm(s) * np.linalg.norm(y):
            rho = 1.0 / sy
            I = np.eye(6)
            if it == 0:
                H = (sy / float(y @ y)) * I  # initial inverse-Hessian scaling
            H = (I - rho * np.outer(s, y)) @ H @ (I - rho * np.outer(y, s)) + rho * np.outer(s, s)
        df = abs(f - f_new)
        x, f_prev, f, g = x_new, f, f_new, g_new
        summary["iterations"] = it + 1
        if callback:
            callback(x)
        if df <= function_tolerance * abs(f_prev):
            summary["termination"] = "function_tolerance"
            break
        if np.linalg.norm(s) <= parameter_tolerance * (np.linalg.norm(x) + parameter_tolerance):
            summary["termination"] = "parameter_tolerance"
            break
    summary["evaluations"] = evals
    summary["final_cost"] = f
    return x, summary


class VisualCameraCalibration:
    """``VisualCameraCalibration`` (visual_camera_calibration.hpp:43-58).

    ``dataset``: list of ``(image_u8 (H,W), points (N,4), intensities (N,))``.
    ``nid_cost_factory(image_f64, points, intensities, bins)`` -> callable ``(x7, want_grad) ->
    (ok, cost, grad7)`` (``NIDCost.__call__``);  ``multi_factory(init_x7, [costs])`` combines pairs
    with MultiNIDCost semantics (default: trust gate + plain sum on the host).
    ``nearest_cost_factory(image_u8, points, intensities, bins)`` -> object with ``calculate(T4x4)``.
    ``cull(points, intensities, T4x4) -> index array`` is the per-outer-iteration view culling
    (view_culling.cpp:21-32); ``None`` disables culling (the CLI's --disable_culling)."""

    def __init__(self, dataset, params=None, nid_cost_factory=None, nearest_cost_factory=None, cull=None, multi_factory=None, trust_gate=None, callback=None,
                 fused_nid_factory=None, fused_nearest_factory=None):
        self.params = params or VisualCameraCalibrationParams()
        self.dataset = dataset
        self.nid_cost_factory = nid_cost_factory
        self.nearest_cost_factory = nearest_cost_factory
        self.cull = cull
        self.multi_factory = multi_factory
        self.trust_gate = trust_gate
        self.callback = callback
        # optional device-resident route: fused_*_factory(pair_index, T4x4, bins) builds the culled cost
        # object straight from a GPU-resident cloud (nid.NIDCost.from_cloud), replacing cull + factory
        self.fused_nid_factory = fused_nid_factory
        self.fused_nearest_factory = fused_nearest_factory
        self.log = []

    # visual_camera_calibration.cpp:35-68
    def calibrate(self, init_T_camera_lidar):
        x = np.asarray(init_T_camera_lidar, dtype=np.float64).copy()
        for outer in range(self.params.max_outer_iterations):
            if self.params.registration_type == "nid_bfgs":
                new_x, info = self.estimate_pose_bfgs(x)
            else:
                new_x, info = self.estimate_pose_nelder_mead(x)
            delta_t, delta_r = se3.delta_trans_rot(new_x, x)
            x = new_x
            converged = delta_t < self.params.delta_trans_thresh and delta_r < self.params.delta_rot_thresh
            self.log.append(dict(outer=outer, delta_t=delta_t, delta_r=delta_r, converged=converged, **info))
            if converged:
                break
        return x

    def _culled(self, x):
        T = se3.to_matrix(x)
        out = []
        for image, points, intensities in self.dataset:
            if self.cull is not None:
                idx = self.cull(points, intensities, T)
                out.append((image, np.ascontiguousarray(points[idx]), np.ascontiguousarray(intensities[idx])))
            else:
                out.append((image, points, intensities))
        return out

    # visual_camera_calibration.cpp:190-238
    def estimate_pose_bfgs(self, init_x):
        costs = []
        if self.fused_nid_factory is not None:
            T = se3.to_matrix(init_x)
            costs = [self.fused_nid_factory(k, T, self.params.nid_bins) for k in range(len(self.dataset))]
        else:
            for image, points, intensities in self._culled(init_x):
                img64 = image.astype(np.float64) * (1.0 / 255.0)  # convertTo(CV_64FC1, 1/255) (:204)
                costs.append(self.nid_cost_factory(img64, points, intensities, self.params.nid_bins))
        if self.multi_factory is not None:
            multi = self.multi_factory(init_x, costs)
        else:
            multi = HostMultiNIDCost(init_x, costs, self.trust_gate)
        x, summary = bfgs_minimize(multi, init_x, max_iterations=self.params.bfgs_max_iterations, callback=self.callback)
        for c in costs:
            if hasattr(c, "close"):
                c.close()
        return x, dict(inner="bfgs", **summary)

    # visual_camera_calibration.cpp:70-139
    def estimate_pose_nelder_mead(self, init_x):
        calcs = []
        if self.fused_nearest_factory is not None:
            T = se3.to_matrix(init_x)
            calcs = [self.fused_nearest_factory(k, T, self.params.nid_bins) for k in range(len(self.dataset))]
        else:
            for image, points, intensities in self._culled(init_x):
                calcs.append(self.nearest_cost_factory(image, points, intensities, self.params.nid_bins))
        T0 = se3.to_matrix(init_x)
        best = [math.inf]

        def note(total, T):
            if total < best[0]:
                best[0] = total
                if self.callback:
                    self.callback(se3.from_matrix(T))
            return total

        def f(x6):
            T = T0 @ se3.pose3_expmap(x6)  # init * Pose3::Expmap(x) (:104)
            total = 0.0
            for c in calcs:
                total += c.calculate(T)
            return note(total, T)

        def f_many(xs):
            # vertices that do not depend on each other (initial simplex, shrink step): every (vertex, pair) evaluation queued
            # before the first is collected (nidreg_submit_iso / nidreg_wait; at most 8 in flight per handle), then the same
            # sums, comparisons and callbacks as f, vertex by vertex in order
            Ts = [T0 @ se3.pose3_expmap(x6) for x6 in xs]
            out = []
            for lo in range(0, len(Ts), 8):
                tickets = [[c.submit(T) for c in calcs] for T in Ts[lo : lo + 8]]
                for T, row in zip(Ts[lo : lo + 8], tickets):
                    total = 0.0
                    for c, t in zip(calcs, row):
                        total += c.wait(t)
                    out.append(note(total, T))
            return out

        p = NelderMeadParams(init_step=self.params.nelder_mead_init_step, convergence_var_thresh=self.params.nelder_mead_convergence_criteria,
                             max_iterations=self.params.max_inner_iterations)
        batched = len(calcs) > 0 and all(hasattr(c, "submit") and hasattr(c, "wait") for c in calcs)
        result = NelderMead(p).optimize(f, np.zeros(6), batch_function=f_many if batched else None)
        T = T0 @ se3.pose3_expmap(result.x)
        for c in calcs:
            if hasattr(c, "close"):
                c.close()
        return se3.from_matrix(T), dict(inner="nelder_mead", iterations=result.num_iterations, evaluations=result.num_evaluations, final_cost=result.y)


class HostMultiNIDCost:
    """MultiNIDCost semantics over arbitrary cost callables (visual_camera_calibration.cpp:147-173):
    trust gate (0.2 m / 2 deg), plain sum, false if any pair failed."""

    def __init__(self, init_x, costs, trust_gate=None):
        self.init = np.asarray(init_x, dtype=np.float64).copy()
        self.costs = costs
        self.trust_gate = trust_gate or default_trust_gate

    def __call__(self, x, want_grad=True):
        if not self.trust_gate(self.init, x):
            return False, float("nan"), None
        total, grad, ok_all = 0.0, np.zeros(7), True
        for c in self.costs:
            ok, v, g = c(x, want_grad)
            ok_all = ok_all and ok
            total += v
            if want_grad and g is not None:
                grad += g
        return ok_all, total, (grad if want_grad else None)


def default_trust_gate(init_x, x):
    dt, dr = se3.delta_trans_rot(init_x, x)
    return not (dt > 0.2 or dr > 2.0 * math.pi / 180.0)
