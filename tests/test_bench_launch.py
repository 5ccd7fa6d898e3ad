"""bench.py's multi-GPU launch path.  CPU: `python bench.py --gpus 2` (no launcher, no WORLD_SIZE) re-launches itself as
two torch.distributed.run ranks, each of which then fails loudly for want of a GPU (the NID core has no CPU path).
GPU (one device): the same command with the test hooks (both ranks on GPU 0, gloo instead of RCCL) produces ONE JSON
line with n_gpus = 2, the weak-scaling value, and every multi_gpu case -- configs[3] pairs, configs[2] / [4] sharded
over the ranks with the all-reduce of the fixed-point histogram, and the single-process route of the C ABI."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra)
    return env


def test_plain_python_launch_spawns_one_rank_per_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("covered by the GPU test")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--points", "1000"], capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode != 0
    # both ranks were started by the elastic launcher and refused to run without a GPU (the launcher terminates the second
    # rank as soon as the first has failed: it may not have got as far as printing its own refusal)
    assert r.stderr.count("bench.py needs an MI355X") >= 1, r.stderr[-2000:]
    assert "local_rank: 1" in r.stderr and "local_rank: 0" in r.stderr, r.stderr[-2000:]


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_produce_the_full_line():
    r = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--blocks", "3", "--points", "300000", "--extra-points-scale", "0.02", "--no-cpu-baseline"],
        capture_output=True, text=True, env=_env(NIDREG_BENCH_ONE_GPU="1", NIDREG_BENCH_BACKEND="gloo"), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert r.stdout.strip() == lines[0]  # stdout is the ONE JSON line: banners of libraries (RCCL prints one on stdout) go to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["ranks_seen"] == 2
    mg = d["multi_gpu"]
    assert mg["ranks_seen"] == 2
    for key in ("pairs_configs3", "shard_configs2", "shard_configs4"):
        assert "error" not in mg[key], mg[key]
        assert mg[key]["value"] > 0 and mg[key]["ranks_seen"] == 2
    sp = mg["single_process_sharded"]
    assert "error" not in sp, sp
    assert sp["configs2"]["devices"] == [0, 0] and sp["configs2"]["value"] > 0 and sp["configs4"]["value"] > 0


@pytest.mark.gpu
def test_a_crash_in_the_single_process_leg_does_not_cost_the_line():
    """multi_gpu.single_process_sharded (one process driving every GPU: cross-device stores and in-kernel waits, never run on two
    physical GPUs) runs in a CHILD process of rank 0: when it dies -- here by abort(), as a GPU memory fault would kill it -- the leg
    records the exit code and the one JSON line still comes out, with the headline and the other legs in it."""
    r = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--blocks", "3", "--points", "300000", "--extra-points-scale", "0.02", "--no-cpu-baseline"],
        capture_output=True, text=True, env=_env(NIDREG_BENCH_ONE_GPU="1", NIDREG_BENCH_BACKEND="gloo", NIDREG_BENCH_TEST_CRASH_CHILD="1"), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    mg = d["multi_gpu"]
    assert "error" not in mg["shard_configs2"], mg["shard_configs2"]
    assert "child process exit code" in mg["single_process_sharded"].get("error", ""), mg["single_process_sharded"]


@pytest.mark.gpu
def test_one_gpu_bench_stdout_is_exactly_one_json_line():
    """`python bench.py` at N = 1 creates a one-rank RCCL communicator inside the library (other_entry_points.inlib_rccl_world1):
    RCCL's version banner must not reach stdout, which carries the JSON line and nothing else."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--blocks", "3", "--points", "300000", "--no-cpu-baseline", "--no-config-legs"],
                       capture_output=True, text=True, env=_env(), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("\n") == 1 and r.stdout.startswith("{"), r.stdout[-2000:]
    d = json.loads(r.stdout)
    rc = d["other_entry_points"]["inlib_rccl_world1"]
    assert "error" not in rc and rc["cost_equals_plain_handle"] and rc["evals_per_s"] > 0


@pytest.mark.gpu
def test_watchdog_prints_the_headline_line_when_a_leg_hangs():
    """A multi_gpu leg that never returns (a collective that hangs) must not cost the driver its JSON line: the watchdog prints the
    headline with the leg marked and every rank leaves."""
    r = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--blocks", "3", "--points", "300000", "--extra-points-scale", "0.02", "--no-cpu-baseline"],
        capture_output=True, text=True, env=_env(NIDREG_BENCH_ONE_GPU="1", NIDREG_BENCH_BACKEND="gloo", NIDREG_BENCH_TEST_HANG_LEG="shard_configs2", NIDREG_BENCH_LEG_WATCHDOG_S="12"), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    mg = d["multi_gpu"]
    assert "error" not in mg["pairs_configs3"] and mg["shard_configs2"]["error"].startswith("watchdog")


@pytest.mark.gpu
def test_one_gpu_bench_carries_shard_proxy_and_time_to_solution():
    """The N = 1 line's strong-scaling inputs (one GPU's eighth of configs[1] / [2] / [4], the exchange cost of co-located shards, the
    RCCL chain's floor, the projection that follows from them) and the end-to-end records (dT of GPU vs CPU `calibrate`), at reduced sizes."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--blocks", "3", "--points", "300000", "--extra-points-scale", "0.02", "--cpu-sample", "100000"],
                       capture_output=True, text=True, env=_env(NIDREG_BENCH_FORCE_LEGS="1"), timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("\n") == 1 and r.stdout.startswith("{"), r.stdout[-2000:]
    d = json.loads(r.stdout)
    sp = d["shard_proxy"]
    for key in ("c2", "c3", "c5"):
        assert "error" not in sp[key] and sp[key]["ms"] > 0 and sp[key]["parts"] == 8, sp[key]
        assert sp["projected_8gpu"][key]["no_exchange_bound"] > 0
    assert "exchange_error" not in sp and sp["exchange_us_3_colocated"] > 0 and sp["rccl_world1_overhead_us"] > 0, sp
    tt = d["time_to_solution"]
    assert "error" not in tt, tt
    for key in ("configs0", "configs1"):
        dt, dr = tt[key]["dT"]
        assert dt <= 1e-3 and dr <= 1e-3 and tt[key]["evals"] > 0 and tt[key]["cpu_wall_s"] > 0, tt[key]
    assert d["timing"]["distinct_poses_in_window"] >= 12 and d["roofline"]["kernel_ms_source"].startswith("HIP events")
