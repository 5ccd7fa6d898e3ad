import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")

# Co-located shards (the same GPU listed several times in desc.device_ids -- the only way a 1-GPU box can run the
# in-library sharding protocol) wait for each other INSIDE kernels; two of their streams must therefore never share one of
# the device's in-order hardware queues.  ROCm multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) queues, and a test
# process owns more streams than that.  Must be set before the HIP runtime initialises; real multi-GPU sets (one shard
# per device) do not need it.
import os  # noqa: E402

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def pytest_sessionfinish(session, exitstatus):
    """Margins the kernels delivered against the oracle over this session (tests/parity.py): written when any check ran."""
    import json

    import parity

    s = parity.summary()
    if not s:
        return
    out = os.environ.get("NIDREG_MARGINS_OUT", os.path.join(ROOT, "gpurun_out", "parity_margins.json"))
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        s["exitstatus"] = int(exitstatus)
        s["bars"] = {"cost_atol": parity.COST_ATOL, "grad_rtol": parity.GRAD_RTOL, "grad_atol": parity.GRAD_ATOL, "hist_atol": parity.HIST_ATOL}
        with open(out, "w") as f:
            json.dump(s, f, indent=1)
    except OSError:
        pass
