"""Parity bars of the GPU tests in ONE place, and a record of the margins the kernels actually deliver.

Every oracle comparison of the ``-m gpu`` suite goes through ``check_cost`` / ``check_grad`` / ``check_hist``: they assert
the bar and remember the observed difference.  ``conftest.py`` writes the session's maxima (and the worst few cases per
quantity) to ``$NIDREG_MARGINS_OUT`` (default ``gpurun_out/parity_margins.json``); the copy committed as
``profiles/r04_parity_margins.json`` is what the bars below were set from: about 10x the largest difference observed over
the whole suite (VERDICT r3 "tighten the parity bars to what the kernels deliver").

Bars (fp64 SPLINE path against the CPU oracle; the reference defines none -- SURVEY.md section 7 planned these):
  * NID               abs <= COST_ATOL
  * 7-gradient        |d| <= GRAD_ATOL + GRAD_RTOL |ref|   per component
  * joint histogram   abs <= HIST_ATOL per bin at test sizes; at 10M / 50M points the bar scales with the bin population
                      (``hist_atol_for``): the fixed-point quantum is 2^-frac per TAP, a bin holding n taps is off by
                      O(sqrt(n)) quanta, and frac drops from 40 to 38 / 36 bits at 10M / 50M points.
"""
import os

import numpy as np

COST_ATOL = 1e-10
GRAD_RTOL = 1e-7
GRAD_ATOL = 1e-10
HIST_ATOL = 1e-9
HIST_IMAGE_ATOL = 1e-8

_records = {"cost": [], "grad": [], "hist": [], "hist_image": []}


def _where():
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]


def _note(kind, value, extra=None):
    rec = {"test": _where(), "value": float(value)}
    if extra:
        rec.update(extra)
    _records[kind].append(rec)


def check_cost(c, ref, atol=None, what=""):
    atol = COST_ATOL if atol is None else atol
    d = abs(float(c) - float(ref))
    _note("cost", d, {"what": what, "bar": atol})
    assert d <= atol, (what, c, ref, d)


def check_grad(g, ref, rtol=None, atol=None, what=""):
    rtol = GRAD_RTOL if rtol is None else rtol
    atol = GRAD_ATOL if atol is None else atol
    g = np.asarray(g, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    d = np.abs(g - ref)
    # how much of the bar the worst component uses, and the rtol that would have been needed at the planned atol of 1e-12
    used = float((d / (atol + rtol * np.abs(ref))).max())
    nz = np.abs(ref) > 0
    need = float((np.maximum(d - 1e-12, 0.0)[nz] / np.abs(ref)[nz]).max()) if np.any(nz) else 0.0
    _note("grad", used, {"what": what, "max_abs": float(d.max()), "rtol_needed_at_atol_1e-12": need, "ref_norm": float(np.linalg.norm(ref)), "bar": [rtol, atol]})
    assert np.all(d <= atol + rtol * np.abs(ref)), (what, g, ref, d)


def check_hist(joint, ref, atol=None, what="", kind="hist"):
    atol = (HIST_ATOL if kind == "hist" else HIST_IMAGE_ATOL) if atol is None else atol
    d = float(np.abs(np.asarray(joint) - np.asarray(ref)).max())
    _note(kind, d, {"what": what, "bar": atol, "ref_max": float(np.abs(ref).max())})
    assert d <= atol, (what, d, atol)


def hist_atol_for(ref_hist):
    """The joint-histogram bar at the full-size configurations (10M / 50M points): HIST_ATOL relative to a bin population of
    1000 -- what a bin of the 30k ... 300k-point test scenes holds at most -- scaled with the largest bin of THIS histogram."""
    return HIST_ATOL * max(1.0, float(np.abs(ref_hist).max()) / 1000.0)


def summary():
    out = {}
    for kind, recs in _records.items():
        if not recs:
            continue
        worst = sorted(recs, key=lambda r: -r["value"])[:5]
        out[kind] = {"checks": len(recs), "max": worst[0]["value"], "worst": worst}
    if _records["grad"]:
        out["grad"]["max_rtol_needed_at_atol_1e-12"] = max(r["rtol_needed_at_atol_1e-12"] for r in _records["grad"])
        out["grad"]["max_abs"] = max(r["max_abs"] for r in _records["grad"])
    return out
