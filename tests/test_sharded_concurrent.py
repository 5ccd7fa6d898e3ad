"""Several sharded pairs evaluated concurrently from host threads -- the reference's OpenMP loop over the pairs of a
multi-bag dataset (visual_camera_calibration.cpp:161-164) with NIDREG_DEVICES set -- at the default number of hardware
queues and with a raised one.  The library serialises the sets of a process per device and launches co-located shards
phase by phase, so that no in-kernel wait can sit behind the kernel it waits for; results equal the plain handles'."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(pairs, shards, bins, hw_queues, workers=False):
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    if hw_queues:
        env["GPU_MAX_HW_QUEUES"] = str(hw_queues)
    env["NIDREG_SHARD_TIMEOUT_MS"] = "2000"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_sharded_concurrent_check.py"), str(pairs), str(shards), str(bins)] + (["workers"] if workers else []), capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("pairs,shards,bins", [(2, 2, 256), (6, 2, 256), (6, 3, 16)])
@pytest.mark.parametrize("hw_queues", [0, 16])
def test_concurrent_sharded_handles(pairs, shards, bins, hw_queues):
    d = _run(pairs, shards, bins, hw_queues)
    assert d["serial_ok"] and d["threads_ok"] and d["hist_ok"] and not d["errors"], d


@pytest.mark.gpu
@pytest.mark.parametrize("pairs,shards,bins", [(1, 2, 256), (2, 3, 256), (1, 3, 16)])
def test_shards_driven_by_their_own_host_threads(pairs, shards, bins):
    """Shards on devices of their own are driven by one host thread each (the caller for shard 0), all four phases queued at once;
    co-located shards are normally launched phase by phase by the caller.  One GPU can exercise the multi-device launch pattern
    when every stream has a hardware queue of its own (GPU_MAX_HW_QUEUES=24, NIDREG_SHARD_COLOCATED_WORKERS=1): same cost bits,
    histogram and gradient as plain handles, cost-only evaluations included, also with two sets evaluated by two caller threads."""
    d = _run(pairs, shards, bins, 24, workers=True)
    assert d["workers"] and d["serial_ok"] and d["threads_ok"] and d["hist_ok"] and d["cost_only_ok"] and not d["errors"], d
