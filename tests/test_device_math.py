"""Scalar device math of csrc/nid_device.hpp checked on the CPU: the header's camera models, fast
reciprocal / rsqrt / atan2 and the hand-derived projection Jacobians are `__host__ __device__`, so
tests/cxx/test_device_math.cpp runs them on the host against the oracle's camera functors
instantiated with Jet<7> (the reference's route to the same derivatives)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_math_on_host():
    src = os.path.join(ROOT, "tests", "cxx", "test_device_math.cpp")
    exe = os.path.join(ROOT, "tests", "cxx", "test_device_math.bin")
    subprocess.check_call(
        ["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-O2", "-ffp-contract=off", "-Wno-unused-value", "-Wno-unused-result", src, "-o", exe]
    )
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fast_atan2 max abs err" in out.stdout
