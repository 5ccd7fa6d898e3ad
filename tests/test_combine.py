"""NIDREG_COMBINE=1: concurrent callers of nidreg_eval on one device (the reference's OpenMP loop over the pairs of a
MultiNIDCost) are collected and evaluated as one grid -- with the same results as evaluating each pair alone (cost bit for bit, gradient to rounding)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(combine, pairs):
    env = dict(os.environ)
    env["NIDREG_COMBINE"] = combine
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_combine_check.py"), str(pairs)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("pairs", [2, 5])
def test_concurrent_callers_combined_into_one_grid(pairs):
    d = _run("1", pairs)
    assert d["cost_identical_grad_equal"] and d["mixed_poses_ok"], d


@pytest.mark.gpu
def test_threads_without_the_combiner():
    d = _run("0", 3)
    assert d["cost_identical_grad_equal"] and d["mixed_poses_ok"], d
