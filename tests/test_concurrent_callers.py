"""Concurrent callers of nidreg_eval on one device -- the reference's OpenMP loop over the pairs of a MultiNIDCost
(visual_camera_calibration.cpp:161), one thread per pair, every thread calling its own NIDCost at the same pose -- give the
same results as evaluating each pair alone (cost bit for bit, gradient to rounding).  An evaluation that has its device to
itself runs with the progress issue priority, the others without: the results must not depend on it."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(pairs):
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_concurrent_check.py"), str(pairs)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("pairs", [2, 3, 5])
def test_concurrent_callers_match_serial_evaluation(pairs):
    d = _run(pairs)
    assert d["cost_identical_grad_equal"] and d["mixed_poses_ok"], d

