#!/usr/bin/env python3
"""bench.py -- NID cost+Jacobian evaluations per second on MI355X (BASELINE.json metric).

A "step" is ONE synchronous evaluation of the hot path -- NIDCost::operator()<Jet<double,7>>, i.e.
cost + 7-gradient -- over one LiDAR-camera pair at a distinct pose (7 doubles in, 8 doubles out,
host sync), with the cloud and image already resident in HBM.  Workload at N=1 = BASELINE.json
configs[1]: 1 pair, 10M-point Ouster-style synthetic cloud + 1920x1080 pinhole, 256 x 256 bins.
The K steps are issued back to back through the C ABI (nidreg_eval_batch: each evaluation completes,
host sync included, before the next starts -- an optimiser's inner loop); the K-step block is
repeated (--blocks, default 25) and `value` / `ms_per_step` are those of the MEDIAN block, so the
timed window is >= 100 ms and one scheduler hiccup cannot move the number (`timing` holds min / max).

N>1: one process per GPU.  Launched by `python -m torch.distributed.run ... bench.py --gpus N` (the
driver's form), or plainly as `python bench.py --gpus N`, which re-launches itself that way.
  value (weak scaling)  one independent pair of the SAME per-GPU workload per GPU (configs[3] style: no
                        data-path collective), pair-evaluations/s summed over the ranks;
  multi_gpu.*           in the same JSON line: configs[3] itself (5M-pt fisheye pair per GPU), and the
                        strong-scaling cases -- configs[2] (10M-pt equirectangular) and configs[4] (50M-pt 4K
                        pinhole) with ONE pair's points sharded over the ranks and an RCCL all-reduce of the
                        int64 fixed-point histogram (+ the 7-gradient) per evaluation -- plus the single-process
                        route of the C ABI (desc.device_ids: direct GPU-to-GPU exchange inside nidreg_eval).

Prints ONE JSON line on rank 0 with `roofline` (per evaluation; per kernel; VALU issue) and, at
N=1, `cpu_baseline` (the oracle timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SHADER_GHZ = 2.4  # MI355X_MICROARCH.md: peak engine clock
NUM_SIMDS = 256 * 4


def algorithmic_bytes(n_points, width, height, bins):
    """16 B per point (one float32 x,y,z,intensity PLY record) counted once, the
    8-bit image once, the scalar histogram once, params/outputs."""
    return 16 * n_points + width * height + 8 * (bins * bins + 2 * bins) + 64


def _same_workload(t, n_points, width, height, bins, precision, camera):
    w = t.get("workload", {})
    if (w.get("points"), w.get("width"), w.get("height"), w.get("bins"), w.get("precision")) != (n_points, width, height, bins, precision):
        return False
    # (summaries of rounds 1-4 carry no camera: they are the headline's; two config cameras share 10M points / 2048 x 2048)
    return camera is None or w.get("camera", "pinhole_1080p") == camera


def _matching_pmc(n_points, width, height, bins, precision, build=None, camera=None):
    """The newest committed rocprofv3 PMC summary (profiles/*_traffic.json) of this workload; with `build`, only
    one stamped with the same kernel-source hash (a summary of another kernel build is not evidence)."""
    import glob

    best = None
    # (profiles/ holds the current round's passes, profiles/archive/ those of earlier rounds -- earlier kernel builds)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "archive", "*_traffic.json"))) + sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            with open(path) as f:
                t = json.load(f)
        except (OSError, ValueError):
            continue
        if not _same_workload(t, n_points, width, height, bins, precision, camera):
            continue
        if build is not None and t.get("kernel_build") != build:
            continue
        best = t
        best["_file"] = os.path.basename(path)
    return best


def _matching_kernel_stats(n_points, width, height, bins, precision, build, camera=None):
    """The newest committed rocprofv3 kernel-stats summary (profiles/*_kernel_stats.json, tools/kernel_stats_json.py) of this
    workload measured on THIS kernel build: average kernel durations without HIP-event markers inside them."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "archive", "*_kernel_stats.json"))) + sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats.json"))):
        try:
            with open(path) as f:
                t = json.load(f)
        except (OSError, ValueError):
            continue
        if not _same_workload(t, n_points, width, height, bins, precision, camera) or t.get("kernel_build") != build:
            continue
        best = t
        best["_file"] = os.path.basename(path)
    return best


def pmc_traffic(kernel, n_points, width, height, bins, precision, build=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of the same workload, FETCH_SIZE doubled
    for the 16 B/lane point stream as MI355X_MICROARCH.md prescribes).  None when no pass matches."""
    t = _matching_pmc(n_points, width, height, bins, precision, build)
    if t is None or kernel not in t.get("kernels", {}):
        return None
    return t["kernels"][kernel]["hbm_bytes_corrected"]


def baseline_config_label(args):
    """Which BASELINE.json config the chosen workload is (the default run is configs[1])."""
    key = (args.points, args.camera, args.bins)
    table = {
        (100000, "pinhole_vga", 16): "BASELINE configs[0] on the GPU",
        (10000000, "pinhole_1080p", 256): "BASELINE configs[1]",
        (10000000, "equirect_2k", 256): "BASELINE configs[2], un-sharded" if args.mode == "pairs" else "BASELINE configs[2]",
        (10000000, "omnidir_2k", 256): "BASELINE configs[2] omnidir variant, un-sharded" if args.mode == "pairs" else "BASELINE configs[2] omnidir variant",
        (5000000, "fisheye_1080p", 256): "BASELINE configs[3], one pair per GPU",
        (50000000, "pinhole_4k", 256): "BASELINE configs[4]",
    }
    return table.get(key, "custom workload")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_one_process_per_gpu(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1) and hand their output through."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def single_process_sharded_leg(world, one_gpu, ps, steps2, blocks2, bins, precision):
    """The single-process route of the C ABI (what an unchanged one-process calibrate uses): ONE process drives every GPU of the
    job -- desc.device_ids, cross-device stores and in-kernel waits.  Runs in a CHILD process of rank 0 (`bench.py
    --single-process-leg <json>`): the route has never met two physical GPUs, and a GPU memory fault there aborts the process that
    caused it -- which must not be the one that owes the driver its JSON line."""
    import numpy as np

    from direct_visual_lidar_calibration_amd import nid, synth

    rng = np.random.default_rng(4321)
    out = {}
    if os.environ.get("NIDREG_BENCH_TEST_CRASH_CHILD"):  # test hook (tests/test_bench_launch.py): die the way a GPU memory fault kills a process
        os.abort()
    # (the first time this route meets two physical GPUs should say WHICH exchange path fails, if one does: every ordered
    # pair of shards ping-pongs once at creation, 200 ms timeout, report on stderr)
    os.environ.setdefault("NIDREG_SHARD_SELFTEST", "1")
    devs = [0] * world if one_gpu else list(range(world))
    for camera, n_points, key, seed in (("equirect_2k", int(10_000_000 * ps), "configs2", 20250523 + 3), ("pinhole_4k", int(50_000_000 * ps), "configs4", 20250523 + 5)):
        s = synth.make_scene(camera, num_points=n_points, seed=seed, device="cuda:0")
        pr = nid.create_camera(s.model, s.intrinsics, s.distortion)
        t1 = time.perf_counter()
        c = nid.NIDCost(pr, s.image_f64, s.points, s.intensities, bins, precision=precision, devices=devs)
        setup = time.perf_counter() - t1
        ps_ = np.ascontiguousarray([synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(steps2)])
        c.eval_batch(ps_[:3])
        secs = []
        for _ in range(blocks2):
            t1 = time.perf_counter()
            c.eval_batch(ps_)
            secs.append(time.perf_counter() - t1)
        med = float(np.median(secs)) / steps2
        out[key] = {"value": round(1.0 / med, 2), "unit": "evals/s", "ms_per_step": round(1e3 * med, 5), "points": n_points, "devices": c.shard_devices(), "setup_s": round(setup, 3)}
        c.close()
        del s
    out["route"] = "one process, desc.device_ids: the cloud cut along the histogram column, every GPU stores its columns of the integer histogram into every other GPU's replica inside nidreg_eval (one exchange), host sums the gradient partials"
    return out


POSE_POOL = 4096  # distinct evaluation poses a timed window draws from (>= 50 evaluations at DISTINCT poses)


def timed_blocks(run_block, steps, blocks, sync, first_block=0):
    """`blocks` K-step blocks, each bracketed by sync(); block b is handed its index so that it evaluates ITS OWN K poses
    (no block replays another's until the pool of POSE_POOL distinct poses is used up); returns the per-block seconds."""
    out = []
    for b in range(blocks):
        sync()
        t0 = time.perf_counter()
        run_block(first_block + b)
        sync()
        out.append(time.perf_counter() - t0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=25, help="repetitions of the K-step timed block (median reported)")
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--camera", default="pinhole_1080p")
    ap.add_argument("--bins", type=int, default=256)
    ap.add_argument("--precision", default=os.environ.get("NIDREG_BENCH_PRECISION", "fp64"))
    ap.add_argument("--mode", default="pairs", choices=["pairs", "shard"], help="what `value` measures at N>1 (the other cases are reported under multi_gpu)")
    ap.add_argument("--columns-per-group", type=int, default=0)
    ap.add_argument("--target-blocks", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="points of the workload the CPU oracle is timed on (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="N>1: skip the multi_gpu.* cases (configs[2], [3], [4], single-process route)")
    ap.add_argument("--no-config-legs", action="store_true", help="N=1: skip the other BASELINE configs and the view-culled form of the workload")
    ap.add_argument("--extra-points-scale", type=float, default=1.0, help="scale the point counts of the multi_gpu.* cases (tests)")
    ap.add_argument("--single-process-leg", default=None, help=argparse.SUPPRESS)  # internal: the child process of multi_gpu.single_process_sharded
    args = ap.parse_args()

    if args.single_process_leg:
        if os.environ.get("NIDREG_BENCH_ONE_GPU"):
            os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        kw = json.loads(args.single_process_leg)
        sys.stdout.flush()
        real = os.dup(1)
        os.dup2(2, 1)  # (library / runtime chatter goes to stderr: stdout carries the leg's one JSON object)
        res = single_process_sharded_leg(**kw)
        os.write(real, (json.dumps(res) + "\n").encode())
        return

    if os.environ.get("NIDREG_BENCH_ONE_GPU"):
        # test hook: co-located shards wait for each other inside kernels and must not share an in-order hardware queue
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_one_process_per_gpu(args.gpus))
    # stdout carries exactly ONE line, the JSON line of rank 0: everything else that writes to file descriptor 1 while the bench
    # runs -- RCCL prints a five-line version banner there when the first communicator of a process is created (torch's "nccl"
    # backend at N > 1, the in-library route's world-1 communicator at N = 1) -- goes to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the NID core)")
    # test hooks (not used by the driver): run several ranks on ONE GPU over gloo to exercise the
    # multi-process code path where only a single device is available
    backend = os.environ.get("NIDREG_BENCH_BACKEND", "nccl")
    one_gpu = bool(os.environ.get("NIDREG_BENCH_ONE_GPU"))
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    cpu_group = None
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        cpu_group = dist.new_group(backend="gloo")  # host-side barriers that leave the GPUs alone
        t = torch.ones(1, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t)  # every rank of the communicator answers: ranks_seen
        ranks_seen = int(t.item())

    from direct_visual_lidar_calibration_amd import _lib, nid, synth

    tuning = dict(columns_per_group=args.columns_per_group, target_blocks=args.target_blocks)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor(seconds, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    def measure(cost, poses, steps, warmup, blocks, batch=True):
        """W warm-up steps, then `blocks` timed K-step blocks (barrier + device sync on both sides, MAX over ranks)."""
        for k in range(warmup):
            ok, c, g = cost(poses[k % len(poses)])
            assert ok, "evaluation rejected during warm-up"
        pool = np.ascontiguousarray(poses)
        # block b evaluates ITS OWN K poses: built outside the timed region (2 probe blocks, then at most 400)
        cache = [np.ascontiguousarray(pool[[(warmup + b * steps + k) % len(pool) for k in range(steps)]]) for b in range(402)]
        if batch and hasattr(cost, "eval_batch"):
            def run_block(b):
                cost.eval_batch(cache[b])
        else:
            def run_block(b):
                for x in cache[b]:
                    cost(x)
        # enough blocks for a timed window of >= 150 ms (the same count on every rank: decided from the MAX over ranks)
        probe = max_over_ranks(timed_blocks(run_block, steps, 2, sync))
        blocks = int(min(400, max(blocks, np.ceil(0.15 / max(min(probe), 1e-6)))))
        secs = max_over_ranks(timed_blocks(run_block, steps, blocks, sync, first_block=2))
        med = float(np.median(secs))
        return {"ms_per_step": 1e3 * med / steps, "block_s": secs, "median_s": med, "distinct_poses": int(min(len(pool), steps * blocks))}

    def timing_summary(m, steps):
        return {"blocks": len(m["block_s"]), "steps_per_block": steps, "window_ms": round(1e3 * sum(m["block_s"]), 2),
                "ms_per_step_median": round(m["ms_per_step"], 5), "ms_per_step_min": round(1e3 * min(m["block_s"]) / steps, 5),
                "ms_per_step_max": round(1e3 * max(m["block_s"]) / steps, 5), "distinct_poses_in_window": m.get("distinct_poses")}

    rng = np.random.default_rng(1234)  # same pose sequence on every rank

    def shard_proxy_ms(proj_, scene_, bins_, parts=8, reps=9):
        """ONE GPU's share of a point-sharded evaluation, timed on this GPU: the handle of the first 1/parts index slice of the cloud
        with the pair's fixed-point unit (desc.scale_points = the whole cloud: exactly what rank 0 of the RCCL route builds),
        synchronous cost+Jacobian evaluations at distinct poses.  An input of the strong-scaling model, not a measurement of it."""
        n_ = scene_.points.shape[0]
        hi_ = n_ // parts
        c_ = nid.NIDCost(proj_, scene_.image_f64, scene_.points[:hi_], scene_.intensities[:hi_], bins_, device=local_rank, precision=args.precision, scale_points=n_)
        ps_ = np.ascontiguousarray([synth.random_pose_near(scene_.T_camera_lidar_true, rng) for _ in range(20 * reps + 4)])
        c_.eval_batch(ps_[:4])
        tl = []
        for r_ in range(reps):
            blk = np.ascontiguousarray(ps_[4 + 20 * r_ : 24 + 20 * r_])
            t1 = time.perf_counter()
            c_.eval_batch(blk)
            tl.append((time.perf_counter() - t1) / len(blk))
        c_.close()
        return {"pts_per_gpu": int(hi_), "parts": parts, "ms": round(1e3 * float(np.median(tl)), 5)}

    # ------------------------------------------------------------------ headline leg
    t0 = time.time()
    if args.mode == "pairs":
        seed = 20250523 + 2 + rank  # config id 2, a different pair per rank
        scene = synth.make_scene(args.camera, num_points=args.points, seed=seed, device=f"cuda:{local_rank}")
        pts, ints = scene.points, scene.intensities
    else:
        scene = synth.make_scene(args.camera, num_points=args.points, seed=20250523 + 2, device=f"cuda:{local_rank}")
        lo = args.points * rank // world
        hi = args.points * (rank + 1) // world
        pts, ints = scene.points[lo:hi], scene.intensities[lo:hi]
    t_gen = time.time() - t0
    proj = nid.create_camera(scene.model, scene.intrinsics, scene.distortion)
    poses = [synth.random_pose_near(scene.T_camera_lidar_true, rng) for _ in range(max(POSE_POOL, args.steps + args.warmup))]

    t0 = time.time()
    if args.mode == "shard" and world > 1:
        from direct_visual_lidar_calibration_amd import parallel

        cost = parallel.ShardedNIDCost(proj, scene.image_f64, pts, ints, args.bins, device=local_rank, precision=args.precision, **tuning)
    else:
        cost = nid.NIDCost(proj, scene.image_f64, pts, ints, args.bins, device=local_rank, precision=args.precision, **tuning)
    torch.cuda.synchronize()
    t_setup = time.time() - t0

    m = measure(cost, poses, args.steps, args.warmup, args.blocks)
    ms_per_step = m["ms_per_step"]
    units_per_step = world if args.mode == "pairs" else 1
    value = units_per_step * 1e3 / ms_per_step

    inner = cost.inner if hasattr(cost, "inner") else cost
    plain = not hasattr(cost, "inner")
    info = inner.info() if hasattr(inner, "info") else {}
    shape = (scene.width, scene.height, scene.model)
    roof = extra = pipelined = culled = configs = shard_proxy = tts_out = cpu = multi = None

    def build_line():
        line = {
            "metric": "NID cost+Jacobian evals/sec on 10M-pt cloud",
            "value": round(value, 3),
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "weak" if args.mode == "pairs" else "strong",
            "vs_baseline": None,
            "dtype": "f64" if args.precision == "fp64" else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{'1 pair per GPU' if args.mode == 'pairs' else '1 pair point-sharded'}, {args.points}-pt Ouster-style cloud + "
                f"{shape[0]}x{shape[1]} {shape[2]}, {args.bins}x{args.bins} NID bins, cost+Jacobian ({baseline_config_label(args)})",
                "points": args.points,
                "image": [shape[0], shape[1]],
                "camera_model": shape[2],
                "camera": args.camera,
                "bins": args.bins,
                "mode": args.mode,
                "accumulate": "u64 fixed point",
                "layout": info,
                "setup_s": round(t_setup, 3),
                "datagen_s": round(t_gen, 3),
                "ranks_seen": ranks_seen,
            },
            "timing": timing_summary(m, args.steps),
            # construction folded in at the reference's usage (one cost object per pair per outer iteration, ~50 evaluations each)
            "amortised_50_evals_per_handle": {
                "first_handle_of_the_process": round(units_per_step * 50.0 / (t_setup + 50.0 * ms_per_step * 1e-3), 1),  # includes loading the code objects, growing the scratch arena
                "later_handles": round(units_per_step * 50.0 / (extra["setup_again_s"] + 50.0 * ms_per_step * 1e-3), 1) if extra and "setup_again_s" in extra else None,
            },
            "roofline": roof,
            "pipelined": pipelined,
            "culled": culled,
            "configs": configs,
            "shard_proxy": shard_proxy,
            "time_to_solution": tts_out,
            "cpu_baseline": cpu,
            "other_entry_points": extra,
            "multi_gpu": multi,
        }
        if cpu:
            line["speedup_vs_cpu_port"] = round(value / cpu["value"], 1)
        return line

    emitted = []

    def emit():
        if emitted:
            return
        emitted.append(True)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(build_line()) + "\n").encode())

    # ---- watchdog: a collective that never completes (a communicator that does not form, a peer that died) blocks where no exception
    # handler reaches.  Every optional leg arms a deadline; when one passes, rank 0 prints the headline line with what has been
    # measured so far (the leg marked as timed out) and every rank leaves the process.
    import threading

    watchdog = {"deadline": None, "leg": None}

    def _watch():
        while True:
            time.sleep(0.5)
            d_ = watchdog["deadline"]
            if d_ is not None and time.time() > d_:
                name_ = watchdog["leg"]
                if rank == 0:
                    try:
                        if isinstance(multi, dict):
                            multi[name_] = {"error": f"watchdog: the leg did not finish within {leg_watchdog_s:.0f} s (a collective or an exchange hung); line printed by the watchdog"}
                        emit()
                    finally:
                        os._exit(0)
                time.sleep(5.0)  # rank 0 prints first
                os._exit(0)

    leg_watchdog_s = float(os.environ.get("NIDREG_BENCH_LEG_WATCHDOG_S", "90"))
    threading.Thread(target=_watch, daemon=True).start()

    def arm(name_, seconds=None):
        watchdog["leg"] = name_
        watchdog["deadline"] = time.time() + (seconds if seconds is not None else leg_watchdog_s)

    def disarm():
        watchdog["deadline"] = None

    # ---- kernel timing with HIP events on the handle's own stream (extra, untimed evaluations)
    python_call_rate = None
    if rank == 0 and plain:
        t1 = time.perf_counter()
        for k in range(args.steps):
            cost(poses[k % len(poses)])
        python_call_rate = args.steps / (time.perf_counter() - t1)
        reps = max(10, min(args.steps, 30))
        # (a) the whole evaluation between two events; (b) the same with an event after every kernel (k_spline_hist /
        # k_entropy / k_spline_grad): the per-pass breakdown
        inner.set_timing(2)
        whole = []
        for k in range(reps):
            inner(poses[k % len(poses)])
            whole.append(inner.timing_ms()["total"])
        inner.set_timing(True)
        acc = {}
        for k in range(reps):
            inner(poses[k % len(poses)])
            tm = inner.timing_ms()
            for key, v in tm.items():
                acc.setdefault(key, []).append(v)
        inner.set_timing(False)
        kt = {key: float(np.mean(v)) for key, v in acc.items()}
        whole_ms = float(np.mean(whole))
        n_local = pts.shape[0]
        build = _lib.library_kernel_build()  # (what the loaded library was built from: a committed summary matches it or is not quoted)
        kstats = _matching_kernel_stats(n_local, scene.width, scene.height, args.bins, args.precision, build, camera=args.camera)
        ks = {k_: v["avg_ns"] * 1e-6 for k_, v in kstats["kernels"].items()} if kstats else {}
        eval_bytes = algorithmic_bytes(n_local, scene.width, scene.height, args.bins)
        eval_achieved = eval_bytes / (ms_per_step * 1e-3) / 1e9
        dom = "k_spline_hist" if kt["hist"] >= kt["grad"] else "k_spline_grad"
        dom_ms_events = max(kt["hist"], kt["grad"])
        # algorithmic bytes ONE launch of a streaming pass moves: 16 B/point + the 8-bit image + the B x B 64-bit histogram
        launch_bytes = 16 * n_local + scene.width * scene.height + 8 * args.bins * args.bins
        # `kernel_frac` is THIS run's own measurement (HIP events on the handle's stream around the launch); the rocprofv3 average of
        # the same kernel build, when a summary stamped with this build's hash is committed, rides beside it as kernel_frac_profiled
        dom_ms = dom_ms_events
        kernel_achieved = launch_bytes / (dom_ms * 1e-3) / 1e9
        pmc = _matching_pmc(n_local, scene.width, scene.height, args.bins, args.precision, build, camera=args.camera)
        pk = pmc.get("kernels", {}) if pmc else {}
        traffic = pk[dom]["hbm_bytes_corrected"] if dom in pk else None
        route = ["k_spline_hist", "k_entropy", "k_spline_grad"]
        eval_traffic = sum(pk[k_]["hbm_bytes_corrected"] for k_ in route) if all(k_ in pk for k_ in route) else None
        # VALU issue roof: wave-instructions the evaluation's kernels issue (PMC SQ_INSTS_VALU of this kernel build)
        # at one quad-cycle (4 clocks) each on 1024 SIMDs -- the roof that actually binds
        valu = None
        insts = {k_: pk[k_].get("valu_insts") for k_ in route if k_ in pk and k_ != "k_entropy"}
        if insts and all(v for v in insts.values()):
            floor_us = {k_: v * 4.0 / NUM_SIMDS / (SHADER_GHZ * 1e3) for k_, v in insts.items()}
            # VALU-busy fraction of a kernel: SQ_ACTIVE_INST_VALU (cycles a CU's VALUs are issuing, summed over the CUs) over the
            # CU-cycles of its duration (256 CUs x kernel time x 2.4 GHz)
            busy = {}
            for k_ in insts:
                act = pk[k_].get("valu_active_quad_cycles")
                t_ms = ks.get(k_) or {"k_spline_hist": kt.get("hist"), "k_spline_grad": kt.get("grad")}.get(k_)
                if act and t_ms:
                    busy[k_] = round(act / (256 * t_ms * 1e-3 * SHADER_GHZ * 1e9), 3)
            # ... and over the mean LIFETIME of a wave instead of the kernel's duration (SQ_WAVE_CYCLES / SQ_WAVES, quad-cycles like
            # SQ_ACTIVE_INST_VALU): what the SIMDs do while the point loops run -- the duration also holds launch, prologue, the
            # staggered end and the finalising workgroup (profiles/r06_experiments.md section 5)
            alive = {}
            for k_ in insts:
                c_ = pk[k_].get("counters", {})
                if c_.get("SQ_WAVE_CYCLES") and c_.get("SQ_WAVES") and c_.get("SQ_ACTIVE_INST_VALU"):
                    alive[k_] = round(c_["SQ_ACTIVE_INST_VALU"] / NUM_SIMDS / (c_["SQ_WAVE_CYCLES"] / c_["SQ_WAVES"]), 3)
            valu = {
                "busy_frac": busy or None,
                "busy_frac_while_waves_alive": alive or None,
                "insts_per_point": {k_: round(v * 64.0 / n_local, 1) for k_, v in insts.items()},
                "issue_floor_us": {k_: round(v, 2) for k_, v in floor_us.items()},
                "frac_of_evaluation": round(sum(floor_us.values()) / (ms_per_step * 1e3), 3),
                "model": "SQ_INSTS_VALU x 4 clk / (1024 SIMDs x 2.4 GHz)",
                "source": pmc["_file"],
            }
        roof = {
            "bound": "hbm",
            "binding_roof": "valu-issue (fp64)",
            "achieved": round(eval_achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(eval_achieved / HBM_PEAK_GBS, 4),  # algorithmic bytes of ONE evaluation / time of one evaluation
            "traffic": traffic,
            "eval_traffic": eval_traffic,  # counter bytes of every kernel of one evaluation (the records are streamed twice)
            "traffic_source": pmc["_file"] if pmc else f"no PMC summary of kernel build {build} committed",
            "kernel": dom,
            "kernel_achieved": round(kernel_achieved, 1),
            "kernel_frac": round(kernel_achieved / HBM_PEAK_GBS, 4),
            "kernel_ms_used": round(dom_ms, 4),
            "kernel_ms_source": "HIP events on the handle's stream around the launch, this run (the markers' few us included: see empty_event_interval_ms)",
            # the interval between two event records with NO work between them, same stream, same evaluations (the `memset` slot: the
            # histogram buffer is pre-cleared by the previous evaluation's kernels, nothing is launched there): what a marker pair costs
            # in this run -- the event figure of a long kernel exceeds its rocprofv3 duration by about this much
            "empty_event_interval_ms": round(kt.get("memset", 0.0), 4),
            "kernel_frac_profiled": round(launch_bytes / (ks[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dom in ks else None,
            "kernel_frac_profiled_source": (kstats["_file"] + " (rocprofv3 --kernel-trace --stats average of the bench command, measured on this kernel build)") if dom in ks else None,
            "achieved_side": "algorithmic bytes / time; the counter figures (`traffic`) are fabric side: L2 -> fabric requests, Infinity-Cache hits included (the 160 MB record array fits the 256 MiB cache)",
            "launch_bytes": launch_bytes,
            "eval_bytes": eval_bytes,
            "route": "three kernels per evaluation",
            "kernel_ms_events": {"whole_evaluation": round(whole_ms, 4), "three_kernel_route": {k_: round(v, 4) for k_, v in kt.items()}},
            "kernel_ms_rocprof": {k_: round(v, 4) for k_, v in ks.items()} or None,
            "kernel_build": build,
            "valu": valu,
        }

    # ---- the other two entry points of the path on the same resident workload (informational):
    # cost only (T = double instantiation, line-search probes) and the Nelder-Mead twin
    if rank == 0 and world == 1:
        def rate(fn, n=30):
            for k in range(3):
                fn(k)
            t1 = time.perf_counter()
            for k in range(n):
                fn(3 + k)
            return n / (time.perf_counter() - t1)

        from direct_visual_lidar_calibration_amd import se3 as _se3

        extra = {"cost_only_evals_per_s": round(rate(lambda k: cost(poses[k % len(poses)], want_grad=False)), 1)}
        t1 = time.perf_counter()
        cloud = nid.Cloud(pts, ints, device=local_rank)
        extra["cloud_upload_s"] = round(time.perf_counter() - t1, 4)
        max_fov = nid.estimate_camera_fov(proj, (scene.width, scene.height), device=local_rank)
        Tm = _se3.to_matrix(scene.T_camera_lidar_init)
        t1 = time.perf_counter()
        near = nid.CostCalculatorNID.from_cloud(proj, scene.image_u8, cloud, nid.NIDCostParams(args.bins), max_fov=max_fov, cull=(Tm, float(np.cos(max_fov)), True),
                                                precision=args.precision)
        extra["device_cull_build_s"] = round(time.perf_counter() - t1, 4)
        extra["culled_points"] = near.num_points
        mats = [_se3.to_matrix(p_) for p_ in poses]
        extra["nearest_evals_per_s"] = round(rate(lambda k: near.calculate(mats[k % len(mats)])), 1)
        near.close()
        cloud.close()
        # a second construction from host arrays: the scratch arena is warm, as on every outer iteration after the first
        t1 = time.perf_counter()
        again = nid.NIDCost(proj, scene.image_f64, pts, ints, args.bins, device=local_rank, precision=args.precision, **tuning)
        torch.cuda.synchronize()
        extra["setup_again_s"] = round(time.perf_counter() - t1, 4)
        again.close()
        if python_call_rate:
            extra["python_call_evals_per_s"] = round(python_call_rate, 1)
        # the in-library RCCL route (nidreg_shard_comm_init; what a C++ caller of the drop-in gets for one pair over several GPUs)
        # with a ONE-rank communicator on this GPU: the all-reduce of one rank moves nothing, so what this measures is the chain
        # histogram -> ncclAllReduce(int64) -> entropy -> gradient -> ncclAllReduce(f64 x 7) itself -- the floor the collectives put
        # under a sharded evaluation whatever the link speed.  Informational; NIDREG_BENCH_NO_INLIB_RCCL=1 skips it.
        if not os.environ.get("NIDREG_BENCH_NO_INLIB_RCCL"):
            try:
                from direct_visual_lidar_calibration_amd import parallel as _par

                one = _par.InLibShardedNIDCost(proj, scene.image_f64, pts, ints, args.bins, device=local_rank, precision=args.precision, total_points=pts.shape[0])
                okr, cr, _gr = one(poses[0])
                okp, cp_, _gp = cost(poses[0])
                extra["inlib_rccl_world1"] = {"evals_per_s": round(rate(lambda k: one(poses[k % len(poses)]), n=20), 1), "cost_equals_plain_handle": bool(okr and okp and cr == cp_)}
                one.close()
            except Exception as exc:
                extra["inlib_rccl_world1"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}

    # ---- the same workload with the evaluations queued ahead (nidreg_submit / nidreg_wait): what a caller with independent
    # poses in hand gets -- Nelder-Mead's initial simplex, multi-start, batches; `value` stays the synchronous rate
    if rank == 0 and world == 1 and plain:
        bp = np.ascontiguousarray([poses[k % len(poses)] for k in range(args.steps)])
        cost.eval_batch(bp[:8], pipelined=True)
        tp = []
        for _ in range(15):
            t1 = time.perf_counter()
            cost.eval_batch(bp, pipelined=True)
            tp.append((time.perf_counter() - t1) / len(bp))
        pipelined = {"evals_per_s": round(1.0 / float(np.median(tp)), 1), "ms_per_step": round(1e3 * float(np.median(tp)), 5), "in_flight": 7,
                     "frac": round(eval_bytes / float(np.median(tp)) / 1e9 / HBM_PEAK_GBS, 4) if roof else None}

    # ---- the view-culled form of the workload (what `calibrate` evaluates: visual_camera_calibration.cpp:201-206 culls before
    # every inner solve): the scene plus a copy pushed out of the view, culled and bucketed on the device
    if rank == 0 and world == 1 and plain and not args.no_config_legs:
        try:
            from direct_visual_lidar_calibration_amd import se3 as _se3c

            moved = (pts + np.array([6.0, 0.0, 0.0, 0.0])).astype(np.float32).astype(np.float64)
            cl = nid.Cloud(np.concatenate([pts, moved]), np.concatenate([ints, ints[::-1]]), device=local_rank)
            del moved
            cc = nid.NIDCost.from_cloud(proj, scene.image_f64, cl, args.bins, cull=(_se3c.to_matrix(scene.T_camera_lidar_init), 0.0, True), precision=args.precision)
            bp = np.ascontiguousarray([poses[k % len(poses)] for k in range(20)])
            cc.eval_batch(bp[:5])
            tcs = []
            for _ in range(9):
                t1 = time.perf_counter()
                cc.eval_batch(bp)
                tcs.append((time.perf_counter() - t1) / len(bp))
            kept = int(cc.info()["num_points"])
            us = 1e6 * float(np.median(tcs))
            culled = {"in_pts": int(2 * pts.shape[0]), "kept": kept, "us_per_eval": round(us, 2), "ns_per_pt": round(1e3 * us / kept, 3), "headline_ns_per_pt": round(1e6 * ms_per_step / pts.shape[0], 3),
                      "frac": round(algorithmic_bytes(kept, scene.width, scene.height, args.bins) / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "seg": [int(cc.info()["segmented"]), int(cc.info()["segmented_hist"])]}
            cc.close()
            cl.close()
        except Exception as exc:  # an informational leg must not cost the headline line
            culled = {"error": f"{type(exc).__name__}: {exc}"[:200]}

    # ---- CPU baseline: the oracle (faithful restatement, 1 core, Jet<7>) on a bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_sample > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib  # test infrastructure, used here only as the timed CPU baseline

        # >= 3 cost+Jacobian evaluations of the WHOLE cloud on one core when that takes <= ~25 s (10M points: ~4.8 s each);
        # larger clouds: every (N/ns)-th sweep-ordered point (keeps the spatial / intensity distribution), scaled, with the
        # per-point cost at a second sample size next to it
        full = pts.shape[0] <= 12_000_000 and args.cpu_sample >= min(2_000_000, pts.shape[0])  # a smaller --cpu-sample asks for a shorter CPU leg
        ns = pts.shape[0] if full else min(args.cpu_sample, pts.shape[0])
        sel = np.linspace(0, pts.shape[0] - 1, ns).astype(np.int64)
        sp, si = (pts, ints) if full else (np.ascontiguousarray(pts[sel]), np.ascontiguousarray(ints[sel]))
        img64 = scene.image_f64
        ts = []
        t_budget = time.time()
        for k in range(3 if full else 8):
            t1 = time.perf_counter()
            oracle_lib.nid_cost(scene.model, scene.intrinsics, scene.distortion, img64, sp, si, args.bins, poses[k % len(poses)], want_grad=True, threads=1)
            ts.append(time.perf_counter() - t1)
            if time.time() - t_budget > 25.0:
                break
        t_med = float(np.median(ts))
        scale = pts.shape[0] / ns
        try:
            with open("/proc/cpuinfo") as f:
                cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
        except OSError:
            cpu_model = "unknown"
        cpu = {
            "value": round(1.0 / (t_med * scale), 6),
            "unit": "evals/s",
            "cores": 1,
            "kind": "port",
            "sample": (f"{len(ts)} cost+Jacobian evals of the oracle (Jet<7>, serial loop like the reference) on all {ns} points, median {t_med:.3f} s" if full else
                       f"{len(ts)} cost+Jacobian evals of the oracle (Jet<7>, serial loop like the reference) on {ns} of the {pts.shape[0]} points, median {t_med:.3f} s, scaled linearly x{scale:.1f}")
            + f"; host cpus={os.cpu_count()}, {cpu_model}",
            "ns_per_point": round(1e9 * t_med / ns, 1),
            "cpu_model": cpu_model,
        }
        if full:  # the per-point cost at a fifth of the cloud, so that the (non-)linearity is on record
            sel5 = np.linspace(0, pts.shape[0] - 1, pts.shape[0] // 5).astype(np.int64)
            s5p, s5i = np.ascontiguousarray(pts[sel5]), np.ascontiguousarray(ints[sel5])
            t1 = time.perf_counter()
            oracle_lib.nid_cost(scene.model, scene.intrinsics, scene.distortion, img64, s5p, s5i, args.bins, poses[0], want_grad=True, threads=1)
            cpu["ns_per_point_at_one_fifth"] = round(1e9 * (time.perf_counter() - t1) / sel5.shape[0], 1)
            sel = np.linspace(0, pts.shape[0] - 1, min(2_000_000, pts.shape[0])).astype(np.int64)
            sp, si = np.ascontiguousarray(pts[sel]), np.ascontiguousarray(ints[sel])  # the informational legs below stay on a 2M sample
            scale = pts.shape[0] / sp.shape[0]
        # informational: the reference's OWN source (include/vlcal/costs/nid_cost.hpp, Jet<7>) when
        # oracle/_ref/libref.so travelled with the snapshot -- compiled against the stand-in Eigen of
        # oracle/shim/, so its speed is not real Eigen's; the headline CPU figure stays the port
        try:
            import ref_lib

            if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref.so")):
                tr = []
                for k in range(2):
                    t1 = time.perf_counter()
                    ref_lib.nid_cost(scene.model, scene.intrinsics, scene.distortion, img64, sp, si, args.bins, poses[k], want_grad=True)
                    tr.append(time.perf_counter() - t1)
                cpu["reference_sources_value"] = round(1.0 / (min(tr) * scale), 6)
                cpu["reference_sources_note"] = "vlcal::NIDCost::operator()<Jet<7>> compiled from the reference tree against stand-in third-party headers (oracle/_ref), 1 core"
        except Exception as exc:  # never let the informational leg break the bench line
            cpu["reference_sources_note"] = f"not timed: {exc}"
        # generous variant: same arithmetic, OpenMP over points on all host cores
        nthr = oracle_lib.num_threads()
        if nthr > 1:
            tgs = []
            for k in range(3):  # the first run warms the thread pool and the per-thread histograms
                t1 = time.perf_counter()
                oracle_lib.nid_cost(scene.model, scene.intrinsics, scene.distortion, img64, sp, si, args.bins, poses[k], want_grad=True, threads=nthr)
                tgs.append(time.perf_counter() - t1)
            tg = min(tgs[1:])
            cpu["generous_value"] = round(1.0 / (tg * scale), 6)
            cpu["generous_cores"] = nthr

    default_workload = (args.points, args.camera, args.bins) == (10_000_000, "pinhole_1080p", 256)
    if rank == 0 and world == 1 and plain and not args.no_config_legs and (default_workload or os.environ.get("NIDREG_BENCH_FORCE_LEGS")):
        shard_proxy = {"note": "ONE-GPU PROJECTION, NOT A MEASUREMENT: t_shard = this GPU's time on 1/8 of the cloud (index slice, the pair's fixed-point unit); "
                               "projected speed-up at 8 GPUs = t_full / (t_shard + exchange)"}
        try:
            shard_proxy["c2"] = dict(shard_proxy_ms(proj, scene, args.bins), camera=args.camera, full_ms=round(ms_per_step, 5))
        except Exception as exc:
            shard_proxy["c2"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    cost.close()
    del scene, pts, ints

    # ---- every other BASELINE.json config on this one GPU (one pair each; the sharded / per-GPU forms are the N > 1 legs):
    # ms per synchronous cost+Jacobian evaluation, algorithmic GB/s, fraction of the HBM roof
    if rank == 0 and world == 1 and not args.no_config_legs and (default_workload or os.environ.get("NIDREG_BENCH_FORCE_LEGS")):
        configs = {}
        t_legs = time.time()
        for key, camera, n_points, bins_ in (("c1", "pinhole_vga", 100_000, 16), ("c3", "equirect_2k", 10_000_000, 256), ("c3b", "omnidir_2k", 10_000_000, 256),
                                             ("c4", "fisheye_1080p", 5_000_000, 256), ("c5", "pinhole_4k", 50_000_000, 256)):
            if time.time() - t_legs > 90.0:
                configs[key] = {"error": "skipped: the config legs' time budget (90 s) was used up"}
                continue
            try:
                s_ = synth.make_scene(camera, num_points=int(n_points * args.extra_points_scale), seed=20250523 + 7, device=f"cuda:{local_rank}")
                pr_ = nid.create_camera(s_.model, s_.intrinsics, s_.distortion)
                c_ = nid.NIDCost(pr_, s_.image_f64, s_.points, s_.intensities, bins_, device=local_rank, precision=args.precision)
                reps = 12 if n_points <= 10_000_000 else 5
                ps_ = np.ascontiguousarray([synth.random_pose_near(s_.T_camera_lidar_true, rng) for _ in range(20 * reps + 4)])  # every evaluation at its own pose
                c_.eval_batch(ps_[:4])
                tl = []
                for r_ in range(reps):
                    blk = np.ascontiguousarray(ps_[4 + 20 * r_ : 24 + 20 * r_])
                    t1 = time.perf_counter()
                    c_.eval_batch(blk)
                    tl.append((time.perf_counter() - t1) / len(blk))
                msl = 1e3 * float(np.median(tl))
                ab = algorithmic_bytes(s_.points.shape[0], s_.width, s_.height, bins_)
                configs[key] = {"pts": int(s_.points.shape[0]), "cam": camera, "bins": bins_, "ms_per_step": round(msl, 5), "evals_per_s": round(1e3 / msl, 1), "gbs": round(ab / (msl * 1e-3) / 1e9, 1),
                                "frac": round(ab / (msl * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                # this camera model's kernels from the committed rocprofv3 passes of THIS kernel build (profiles/*_<camera>_*_kernel_stats.json,
                # *_traffic.json; tools/round_pass.sh stats= / pmc=): average kernel durations, VALU wave-instructions per point
                bld = _lib.library_kernel_build()
                ks_ = _matching_kernel_stats(int(s_.points.shape[0]), s_.width, s_.height, bins_, args.precision, bld, camera=camera)
                if ks_:
                    configs[key]["kernel_ms"] = {k_: round(v["avg_ns"] * 1e-6, 5) for k_, v in ks_["kernels"].items() if k_ in ("k_spline_hist", "k_entropy", "k_spline_grad")}
                    configs[key]["kernel_ms_source"] = "profiles/" + ks_["_file"]
                pm_ = _matching_pmc(int(s_.points.shape[0]), s_.width, s_.height, bins_, args.precision, bld, camera=camera)
                if pm_:
                    vi = {k_: round(v["valu_insts"] * 64.0 / s_.points.shape[0], 1) for k_, v in pm_.get("kernels", {}).items() if k_ in ("k_spline_hist", "k_spline_grad") and v.get("valu_insts")}
                    if vi:
                        configs[key]["valu_insts_per_point"] = vi
                    tr = {k_: v["hbm_bytes_corrected"] for k_, v in pm_.get("kernels", {}).items() if "hbm_bytes_corrected" in v}
                    if tr:
                        configs[key]["traffic_bytes"] = tr
                c_.close()
                if shard_proxy is not None and key in ("c3", "c5"):  # the sharded configs: one GPU's eighth of the same cloud
                    try:
                        shard_proxy[key] = dict(shard_proxy_ms(pr_, s_, bins_, reps=9 if key == "c3" else 5), camera=camera, full_ms=round(msl, 5))
                    except Exception as exc:
                        shard_proxy[key] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
                del s_, c_
                torch.cuda.empty_cache()
            except Exception as exc:
                configs[key] = {"error": f"{type(exc).__name__}: {exc}"[:200]}

    # ---- the exchange term of the strong-scaling model: the in-library protocol with the device listed 2 / 3 times on a cloud so small
    # that the kernels' work is negligible (a separate process: co-located shards need GPU_MAX_HW_QUEUES set before the runtime starts
    # and wait for each other inside kernels -- bounded there by the wall clock, here by a timeout), and the RCCL chain's floor
    if shard_proxy is not None:
        try:
            r_ = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_cost.py"), str(args.bins), "--tiny-only"], capture_output=True, text=True, timeout=120)
            row = json.loads(r_.stdout.strip().splitlines()[-1])["tiny"]
            base = row["plain_three_kernels"]["us_per_eval_cost_grad"]
            shard_proxy["exchange_us_2_colocated"] = round(row["shards_2"]["us_per_eval_cost_grad"] - base, 2)
            shard_proxy["exchange_us_3_colocated"] = round(row["shards_3"]["us_per_eval_cost_grad"] - base, 2)
            shard_proxy["colocated_us_per_eval"] = {k_: v["us_per_eval_cost_grad"] for k_, v in row.items()}
        except Exception as exc:
            shard_proxy["exchange_error"] = f"{type(exc).__name__}: {exc}"[:200]
        w1 = (extra or {}).get("inlib_rccl_world1", {})
        if w1.get("evals_per_s"):
            shard_proxy["rccl_world1_overhead_us"] = round(1e6 / w1["evals_per_s"] - 1e3 * ms_per_step, 2)
        proj8 = {}
        for key in ("c2", "c3", "c5"):
            e_ = shard_proxy.get(key, {})
            if "ms" not in e_:
                continue
            row8 = {}
            if "exchange_us_3_colocated" in shard_proxy:
                row8["single_process_route"] = round(e_["full_ms"] / (e_["ms"] + 1e-3 * shard_proxy["exchange_us_3_colocated"]), 2)
            if "rccl_world1_overhead_us" in shard_proxy:
                row8["rccl_route_floor"] = round(e_["full_ms"] / (e_["ms"] + 1e-3 * shard_proxy["rccl_world1_overhead_us"]), 2)
            row8["no_exchange_bound"] = round(e_["full_ms"] / e_["ms"], 2)
            proj8[key] = row8
        shard_proxy["projected_8gpu"] = proj8

    # ---- time to solution (north star: final T_lidar_camera within 1e-3 m / 1e-3 rad of the CPU path on identical inputs): the whole
    # `calibrate` outer loop on the GPU engine and, as a CPU-baseline leg, the same host driver on the oracle -- BASELINE configs[0]
    # (serial oracle) and configs[1] (the oracle's OpenMP split over points on every host core; bounded on both sides)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_config_legs and (default_workload or os.environ.get("NIDREG_BENCH_FORCE_LEGS")):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        tts_out = {}
        try:
            import oracle_lib as _ol
            import time_to_solution as tts  # test / bench infrastructure: drives the product path and, beside it, the oracle

            scale_ = args.extra_points_scale
            tts_out["configs0"] = tts.compare("configs0", "nid_bfgs", points=int(100_000 * scale_), threads=1, repeats=3, cpu_budget_s=60.0)
            tts_out["configs1"] = tts.compare("configs1", "nid_bfgs", points=int(10_000_000 * scale_), threads=_ol.num_threads(), repeats=2, device=f"cuda:{local_rank}", cpu_budget_s=150.0,
                                              max_outer_iterations=2, bfgs_max_iterations=12)
            torch.cuda.empty_cache()
        except Exception as exc:
            tts_out["error"] = f"{type(exc).__name__}: {exc}"[:300]

    # ------------------------------------------------------------------ N>1: the other multi-GPU cases, same JSON line
    if world > 1 and not args.no_extra_legs:
        from direct_visual_lidar_calibration_amd import parallel

        multi = {"ranks_seen": ranks_seen}
        ps = args.extra_points_scale
        steps2 = max(5, args.steps // 2)
        blocks2 = max(3, args.blocks // 5)

        t_legs = time.time()
        leg_budget_s = float(os.environ.get("NIDREG_BENCH_LEG_BUDGET_S", "300"))

        def leg(name, fn):
            # every optional case is bounded: a case that fails (an exchange that times out: NIDREG_SHARD_TIMEOUT_MS, one
            # evaluation) is recorded as an error, and once the cases together have used their wall-clock budget the rest is
            # skipped -- on every rank alike (the decision is rank 0's, broadcast) -- so that the headline line always comes out
            skip = torch.tensor([1 if time.time() - t_legs > leg_budget_s else 0], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
            if dist is not None:
                dist.broadcast(skip, src=0)
            if int(skip.item()):
                multi[name] = {"error": f"skipped: the multi_gpu cases' wall-clock budget ({leg_budget_s:.0f} s) was used up"}
                return
            t_leg = time.time()
            arm(name)
            try:
                if os.environ.get("NIDREG_BENCH_TEST_HANG_LEG") == name:  # test hook: a leg that never returns (tests/test_bench_launch.py)
                    time.sleep(1e6)
                multi[name] = fn()
            except Exception as exc:  # an optional case must not cost the headline line
                multi[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            disarm()
            if isinstance(multi[name], dict):
                multi[name]["leg_s"] = round(time.time() - t_leg, 1)
            if dist is not None:
                dist.barrier(group=cpu_group)

        def pairs_leg(camera, n_points, config_name):
            s = synth.make_scene(camera, num_points=n_points, seed=20250523 + 4 + rank, device=f"cuda:{local_rank}")
            pr = nid.create_camera(s.model, s.intrinsics, s.distortion)
            c = nid.NIDCost(pr, s.image_f64, s.points, s.intensities, args.bins, device=local_rank, precision=args.precision)
            ps_ = [synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(512)]
            mm = measure(c, ps_, steps2, 3, blocks2)
            c.close()
            return {"config": config_name, "scaling": "weak", "value": round(world * 1e3 / mm["ms_per_step"], 2), "unit": "pair-evals/s (all ranks)", "ms_per_step": round(mm["ms_per_step"], 5),
                    "points_per_gpu": n_points, "ranks_seen": ranks_seen}

        def shard_leg(camera, n_points, config_name, seed, inlib=False):
            s = synth.make_scene(camera, num_points=n_points, seed=seed, device=f"cuda:{local_rank}")
            pr = nid.create_camera(s.model, s.intrinsics, s.distortion)
            lo_, hi_ = parallel.shard_slice(n_points, rank, world)
            # inlib: the collectives run INSIDE libnidreg.so (nidreg_shard_comm_init: dlopen'ed RCCL on the handle's stream) -- what
            # a C++ caller of the drop-in gets; otherwise the split-phase ABI with torch.distributed's all-reduce between the phases
            cls = parallel.InLibShardedNIDCost if inlib else parallel.ShardedNIDCost
            c = cls(pr, s.image_f64, s.points[lo_:hi_], s.intensities[lo_:hi_], args.bins, device=local_rank, precision=args.precision, total_points=n_points)
            ps_ = [synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(512)]
            mm = measure(c, ps_, steps2, 3, blocks2, batch=False)
            c.close()
            return {"config": config_name, "scaling": "strong", "value": round(1e3 / mm["ms_per_step"], 2), "unit": "evals/s", "ms_per_step": round(mm["ms_per_step"], 5),
                    "points": n_points, "points_per_gpu": hi_ - lo_, "collective": "RCCL all-reduce: int64 fixed-point histogram + 7-double gradient" + (", issued by libnidreg.so itself (dlopen)" if inlib else ""),
                    "ranks_seen": ranks_seen}

        leg("pairs_configs3", lambda: pairs_leg("fisheye_1080p", int(5_000_000 * ps), "BASELINE configs[3]: one 5M-pt fisheye pair per GPU"))
        leg("shard_configs2", lambda: shard_leg("equirect_2k", int(10_000_000 * ps), "BASELINE configs[2]: 10M-pt equirectangular, points sharded", 20250523 + 3))
        leg("shard_configs4", lambda: shard_leg("pinhole_4k", int(50_000_000 * ps), "BASELINE configs[4]: 50M-pt 4K pinhole, points sharded", 20250523 + 5))
        # RCCL inside the library over N ranks: opt-in (NIDREG_BENCH_INLIB_RCCL=1).  The route is tested with one-rank communicators
        # only (tests/test_rccl_inlib.py: a one-GPU box offers nothing else); a communicator that never forms would hang this leg --
        # and with it the one JSON line -- where no exception handler can reach, so the default multi-GPU run does not risk it.
        if backend == "nccl" and os.environ.get("NIDREG_BENCH_INLIB_RCCL"):
            leg("shard_configs2_inlib", lambda: shard_leg("equirect_2k", int(10_000_000 * ps), "BASELINE configs[2]: 10M-pt equirectangular, points sharded, RCCL inside the library", 20250523 + 3, inlib=True))

        # the single-process route of the C ABI (what an unchanged one-process calibrate uses): a CHILD process of rank 0 drives every
        # GPU of the job itself (single_process_sharded_leg above), the ranks wait on the host
        def single_process_leg():
            if rank != 0:
                return {}
            kw = dict(world=world, one_gpu=one_gpu, ps=ps, steps2=steps2, blocks2=blocks2, bins=args.bins, precision=args.precision)
            env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                                      "TORCHELASTIC_RUN_ID", "NIDREG_BENCH_TEST_HANG_LEG")}
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--single-process-leg", json.dumps(kw)], env=env, capture_output=True, text=True, timeout=max(30.0, leg_watchdog_s - 5.0))
            except subprocess.TimeoutExpired:
                return {"error": f"the child process did not finish within {leg_watchdog_s - 5.0:.0f} s (killed)"}
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                return {"error": f"child process exit code {r.returncode}: " + (r.stderr or "").strip()[-240:]}
            return json.loads(lines[-1])

        leg("single_process_sharded", single_process_leg)

    watchdog["deadline"] = None
    if rank == 0:
        emit()
    if dist is not None:
        dist.barrier(group=cpu_group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
