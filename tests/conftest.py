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
