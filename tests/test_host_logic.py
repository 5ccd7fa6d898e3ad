"""CPU tests of the host-side conventions the callers rely on (se3.py, dfo.py): Sophus SE3 storage
order and right-plus manifold, the 7x6 PlusJacobian Ceres multiplies onto the ambient gradient, GTSAM
Pose3::Expmap, TUM order of calib.json, and the restated Nelder-Mead."""
import numpy as np
import scipy.linalg

from direct_visual_lidar_calibration_amd import se3
from direct_visual_lidar_calibration_amd.dfo import NelderMead, NelderMeadParams


def rand_pose(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return np.concatenate([q, rng.normal(size=3)])


def test_matrix_roundtrip_and_inverse():
    rng = np.random.default_rng(0)
    for _ in range(20):
        x = rand_pose(rng)
        T = se3.to_matrix(x)
        assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-12) and abs(np.linalg.det(T[:3, :3]) - 1) < 1e-12
        y = se3.from_matrix(T)
        assert np.allclose(se3.to_matrix(y), T, atol=1e-12)
        assert np.allclose(se3.to_matrix(se3.inverse(x)), np.linalg.inv(T), atol=1e-12)
        z = rand_pose(rng)
        assert np.allclose(se3.to_matrix(se3.compose(x, z)), T @ se3.to_matrix(z), atol=1e-12)


def test_exp_matches_matrix_exponential():
    rng = np.random.default_rng(1)
    for scale in (1e-12, 1e-6, 0.3, 2.0):
        d = rng.normal(size=6) * scale
        ups, om = d[:3], d[3:]
        A = np.zeros((4, 4))
        A[:3, :3] = se3.hat(om)
        A[:3, 3] = ups
        assert np.allclose(se3.to_matrix(se3.se3_exp(d)), scipy.linalg.expm(A), atol=1e-10)
        # GTSAM order is [omega; v]
        assert np.allclose(se3.pose3_expmap(np.concatenate([om, ups])), scipy.linalg.expm(A), atol=1e-10)


def test_plus_jacobian_is_the_derivative_of_plus():
    rng = np.random.default_rng(2)
    x = rand_pose(rng)
    J = se3.plus_jacobian(x)
    h = 1e-7
    for k in range(6):
        e = np.zeros(6)
        e[k] = h
        # un-normalised difference of the 7 storage numbers
        fd = (se3.plus(x, e) - se3.plus(x, -e)) / (2 * h)
        assert np.allclose(fd, J[:, k], atol=1e-6)


def test_tum_order_and_delta():
    x = rand_pose(np.random.default_rng(3))
    tum = se3.to_tum(x)
    assert np.allclose(tum[:3], x[4:]) and np.allclose(tum[3:], x[:4])
    assert np.allclose(se3.from_tum(tum), x)
    y = se3.plus(x, np.array([0.03, -0.04, 0.0, 0.0, 0.0, 0.02]))
    dt, dr = se3.delta_trans_rot(x, y)
    assert abs(dr - 0.02) < 1e-9 and abs(dt - 0.05) < 1e-3


def test_nelder_mead_minimises_and_counts_like_the_reference():
    calls = []

    def f(x):
        calls.append(x.copy())
        return float((x[0] - 1.0) ** 2 + 10 * (x[1] + 0.5) ** 2)

    r = NelderMead(NelderMeadParams(init_step=0.1, convergence_var_thresh=1e-12, max_iterations=500)).optimize(f, np.zeros(2))
    assert r.converged and np.allclose(r.x, [1.0, -0.5], atol=1e-4)
    assert r.num_evaluations == len(calls)
    # simplex construction: x0, then x0 + init_step * e_i (nelder_mead.hpp:36-46)
    assert np.allclose(calls[0], [0, 0]) and np.allclose(calls[1], [0.1, 0]) and np.allclose(calls[2], [0, 0.1])
    # the centroid is EVALUATED every iteration (nelder_mead.hpp:58): >= 2 evaluations per iteration
    assert r.num_evaluations >= 3 + 2 * r.num_iterations


def test_nelder_mead_batch_function_changes_nothing_but_the_grouping():
    """dfo.NelderMead.optimize(batch_function=...) hands the vertices that do not depend on each other -- the n + 1 of the
    initial simplex (nelder_mead.hpp:37-45), the n of a shrink step (:88-92) -- to the objective in one call (a GPU objective
    then has them all in flight: nidreg_submit / nidreg_wait); every probe, its order, the result and the evaluation count
    must equal the vertex-by-vertex run.  A Rosenbrock-like valley forces shrink steps."""
    def f(x):
        return float(100.0 * (x[1] - x[0] ** 2) ** 2 + (1.0 - x[0]) ** 2 + 0.3 * np.sin(7 * x[2]) ** 2 + x[2] ** 2)

    seq_a, seq_b, batches = [], [], []

    def fa(x):
        seq_a.append(x.copy())
        return f(x)

    def fb(x):
        seq_b.append(x.copy())
        return f(x)

    def fb_many(xs):
        batches.append(len(xs))
        return [fb(x) for x in xs]

    params = NelderMeadParams(init_step=0.7, convergence_var_thresh=1e-14, max_iterations=400)
    ra = NelderMead(params).optimize(fa, np.array([-1.2, 1.0, 0.5]))
    rb = NelderMead(params).optimize(fb, np.array([-1.2, 1.0, 0.5]), batch_function=fb_many)
    assert ra.num_evaluations == rb.num_evaluations == len(seq_a) == len(seq_b)
    assert all(np.array_equal(a, b) for a, b in zip(seq_a, seq_b))
    assert np.array_equal(ra.x, rb.x) and ra.y == rb.y and ra.num_iterations == rb.num_iterations
    assert batches[0] == 4 and all(b == 3 for b in batches[1:]) and len(batches) >= 2  # the initial simplex, then shrink steps


def test_bench_helpers_algorithmic_bytes_labels_and_committed_traffic():
    """bench.py's bookkeeping (no GPU): SURVEY 8(d) byte formula, BASELINE config labels, and the PMC traffic
    figure it reports is the one committed under profiles/ for the default workload."""
    import argparse
    import glob
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    # 16 N + W H + 8 (B^2 + 2 B) + 64; config 2 = 162.6 MB
    assert bench.algorithmic_bytes(10_000_000, 1920, 1080, 256) == 16 * 10_000_000 + 1920 * 1080 + 8 * (256 * 256 + 2 * 256) + 64
    assert abs(bench.algorithmic_bytes(10_000_000, 1920, 1080, 256) / 1e6 - 162.6) < 0.05
    ns = argparse.Namespace(points=10_000_000, camera="pinhole_1080p", bins=256, mode="pairs")
    assert bench.baseline_config_label(ns) == "BASELINE configs[1]"
    ns.camera, ns.points = "fisheye_1080p", 5_000_000
    assert "configs[3]" in bench.baseline_config_label(ns)
    ns.points = 123
    assert bench.baseline_config_label(ns) == "custom workload"
    files = sorted(glob.glob(os.path.join(root, "profiles", "archive", "*_traffic.json"))) + sorted(glob.glob(os.path.join(root, "profiles", "*_traffic.json")))
    assert files, "a PMC traffic summary must be committed under profiles/"
    headline = [f for f in files if json.load(open(f)).get("workload", {}).get("width") == 1920 and json.load(open(f))["workload"].get("points") == 10_000_000]
    t = json.load(open(headline[-1]))  # (other camera models' summaries are committed next to it since round 5)
    got = bench.pmc_traffic("k_spline_grad", 10_000_000, 1920, 1080, 256, "fp64")
    assert got == t["kernels"]["k_spline_grad"]["hbm_bytes_corrected"]
    assert 0.95 < got / bench.algorithmic_bytes(10_000_000, 1920, 1080, 256) < 1.15  # no over-fetch
    assert bench.pmc_traffic("k_spline_grad", 7, 1, 1, 16, "fp64") is None


def test_column_group_partition_of_a_sharded_pair():
    """The cut of a pair's column groups over n GPUs (csrc/nidreg_shard.hip partition_groups, through a test hook of the library --
    host arithmetic, no GPU): contiguous, exhaustive, balanced to within one group, well defined for empty clouds and more
    GPUs than groups."""
    import ctypes

    from direct_visual_lidar_calibration_amd import _lib

    lib = _lib.load()
    lib.nidreg_debug_partition_groups.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]

    def cut(counts, n):
        g = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        out = (ctypes.c_int * (n + 1))()
        assert lib.nidreg_debug_partition_groups(g.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), len(counts), n, out) == 0
        return list(out), g

    rng = np.random.default_rng(3)
    for ng, n in ((256, 8), (256, 3), (16, 8), (16, 5), (7, 2), (4, 8), (1, 2)):
        for trial in range(5):
            counts = rng.integers(0, 2000, ng) if trial else np.full(ng, 39062)
            if trial == 4:
                counts[rng.integers(0, ng, ng // 2)] = 0  # culled-away columns
            c, g = cut(counts, n)
            assert c[0] == 0 and c[-1] == ng and all(a <= b for a, b in zip(c, c[1:]))  # contiguous, exhaustive, ordered
            total = int(g[-1])
            biggest = int(counts.max()) if len(counts) else 0
            for k in range(n):
                share = int(g[c[k + 1]] - g[c[k]])
                assert abs(share - total / n) <= biggest + 1, (ng, n, trial, c)  # balanced to within one group
    c, g = cut(np.zeros(256, dtype=np.int64), 8)  # nothing to balance: equal column ranges
    assert c == [0, 32, 64, 96, 128, 160, 192, 224, 256]
    c, g = cut(np.full(256, 39062), 8)
    assert c == [0, 32, 64, 96, 128, 160, 192, 224, 256]


def test_chunk_tables_fit_one_round_of_workgroups():
    """The chunk table of a pass (csrc/nidreg_plan.hip split_groups, through a test hook -- host arithmetic, no GPU).  Since round 4 a
    chunk is a contiguous range of records that may run across column-group boundaries (the workgroup flushes its tile at each
    one), and the table minimises the longest chunk under the cost model  cost = sum over segments of (overhead + records):
    every record in exactly one chunk, `group` = the group of the chunk's first record, never more chunks than the round
    holds, and no chunk costlier than the mean cost plus what one segment may add -- also for column populations that are
    not uniform (a pair of a multi-pair set, a view-culled cloud: the clouds `calibrate` really evaluates)."""
    import ctypes

    from direct_visual_lidar_calibration_amd import _lib

    lib = _lib.load()
    lib.nidreg_debug_chunk_table.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]

    def table(counts, target, overhead, pair=-1, max_segs=4):
        g = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        cap = 4 * (len(counts) + target) + 16
        rows = (ctypes.c_uint32 * (4 * cap))()
        n = lib.nidreg_debug_chunk_table(g.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), len(counts), target, overhead, max_segs, pair, rows, cap)
        assert 0 <= n <= cap
        return np.array(rows[: 4 * n], dtype=np.int64).reshape(n, 4), g

    def chunk_cost(start, count, g, overhead):
        """segments the chunk [start, start + count) is cut into by the group offsets g, and its cost"""
        lo = np.searchsorted(g, start, side="right") - 1
        hi = np.searchsorted(g, start + count, side="left")
        segs = sum(1 for k in range(lo, hi) if min(g[k + 1], start + count) > max(g[k], start))
        return segs, segs * overhead + count

    rng = np.random.default_rng(11)
    cases = [(np.full(256, 39062), 1024, 384), (np.full(256, 39062), 512, 1024)]  # cfg 2
    cases += [(rng.binomial(40000, 0.5, 256) + rng.integers(-600, 600, 256), t, ov) for t, ov in ((512, 384), (256, 1024), (1024, 384))]  # a pair of a multi-pair set
    culled = rng.integers(0, 60000, 256)
    culled[rng.integers(0, 256, 100)] = 0
    cases += [(culled, 1024, 384), (culled, 512, 1024), (culled, 64, 384), (np.array([5]), 1024, 384), (np.zeros(16, dtype=np.int64), 1024, 384), (rng.integers(0, 300, 16), 1024, 384),
              (rng.integers(0, 40, 256), 1024, 1024), (culled, 1024, 0)]
    for counts, target, overhead in cases:
        for pair, max_segs in ((-1, 4), (3, 4), (-1, 1)):
            rows, g = table(counts, target, overhead, pair, max_segs)
            n_rec = int(g[-1])
            nonempty = int(np.count_nonzero(counts))
            assert len(rows) <= max(target, -(-nonempty // max_segs)) and (len(rows) > 0) == (n_rec > 0)  # (a chunk holds at most max_segs groups)
            covered = np.zeros(n_rec, dtype=np.int32)
            costs = []
            slot = 0  # Chunk::pad = pair | (segments in the table's earlier chunks) << 8: one gradient partial per segment, in table order
            for k, (start, count, group, pad) in enumerate(rows):
                assert count > 0 and g[group] <= start < g[group + 1]  # the group of the chunk's first record
                if start > g[group]:
                    assert (start - g[group]) % 64 == 0  # cuts inside a group fall on whole waves
                covered[start : start + count] += 1
                assert pad == ((0 if pair < 0 else pair) | (slot << 8))
                segs, cost = chunk_cost(start, count, g, overhead)
                assert 1 <= segs <= max_segs  # max_segs = 1: a chunk inside one column group, as in rounds 1-3 (kernels whose tile spans several columns)
                slot += segs
                costs.append(cost)
            assert np.all(covered == 1)
            if len(rows) and max_segs > 1:
                nonempty = int(np.count_nonzero(counts))
                # the longest chunk: no worse than an even share of the total cost plus one segment's overhead and rounding
                # (a lower bound on any table of `target` chunks is (records + nonempty * overhead) / target)
                # (+5 %: a chunk that reaches its segment limit closes early and the others take up the slack)
                # (x 1.10: a table of one group per chunk is kept unless chunks across groups gain more than 10 % in the cost model --
                # the looped kernel instantiations are a few per cent slower per point, NIDREG_SEG_MIN_GAIN)
                bound = 1.10 * 1.05 * (n_rec + (nonempty + target) * overhead) / target + overhead + 128
                assert max(costs) <= bound, (max(costs), bound, len(rows), target)
    # the uniform cloud keeps the table it always had: four equal parts per column
    rows, _ = table(np.full(256, 39062), 1024, 384)
    assert len(rows) == 1024 and rows[:, 1].max() <= 9792
    rows, _ = table(np.full(256, 39062), 512, 1024)
    assert len(rows) == 512 and rows[:, 1].max() <= 19584
    # a view-culled cloud: the longest chunk stays within a few per cent of the mean, where whole parts per column left 3/4 ... 3/2
    rows, g = table(culled, 1024, 384)
    assert rows[:, 1].max() <= 1.10 * 1.08 * culled.sum() / 1024 + 64, (rows[:, 1].max(), culled.sum() / 1024)
    # two workgroups per column on average (the WIDE histogram kernel's table) and columns of 0 ... 60 000 points: chunks across
    # groups win by more than the threshold and are used
    rows, g = table(culled, 512, 1024)
    assert (rows[-1, 3] >> 8) + 1 > len(rows) and rows[:, 1].max() <= 1.08 * culled.sum() / 512 + 64


def test_small_clouds_get_fewer_chunks_than_a_full_round():
    """The number of chunks of a pass (csrc/nidreg_plan.hip round_chunks / snap_to_groups, through a test hook): CUs/2 at 100k points,
    growing with the square root of the cloud, the full round of co-resident workgroups from 6.4M points on; whole multiples of
    the non-empty column groups and never fewer chunks than groups (profiles/archive/r04i_small_cloud_sweep.jsonl: what each of
    these choices was measured against)."""
    import ctypes

    from direct_visual_lidar_calibration_amd import _lib

    lib = _lib.load()
    lib.nidreg_debug_round_chunks.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_int]

    def chunks(counts, per_cu=4, cus=256):
        g = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        return lib.nidreg_debug_round_chunks(per_cu, cus, g.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), len(counts))

    one = [chunks([n]) for n in (1, 30_000, 100_000, 300_000, 1_000_000, 3_000_000, 6_400_000, 10_000_000, 50_000_000)]
    assert one == [1, 70, 128, 256, 512, 768, 1024, 1024, 1024]  # (from 0.6 CUs on: whole multiples of the CU count, round 5)
    assert chunks([10_000_000], per_cu=2) == 512 and chunks([10_000_000], per_cu=3) == 768  # the WIDE histogram kernel / the fisheye gradient kernel
    for bins_groups in (16, 64, 256):  # B = 64 / 128 / 256
        for n in (30_000, 100_000, 1_000_000, 3_000_000, 10_000_000):
            c = chunks(np.full(bins_groups, n // bins_groups))
            assert c % bins_groups == 0 and bins_groups <= c <= 1024 and abs(c - max(bins_groups, min(1024, chunks([n])))) <= bins_groups // 2 + (1024 % bins_groups)
    assert chunks(np.full(256, 117)) == 256 and chunks(np.full(256, 3906)) == 512 and chunks(np.full(256, 39062)) == 1024
    assert chunks(np.full(256, 39062), per_cu=2) == 512 and chunks(np.full(256, 39062), per_cu=3) == 768
    culled = np.full(256, 400)
    culled[::2] = 0  # half of the columns empty: multiples of the 128 that hold points
    assert chunks(culled) % 128 == 0
    assert chunks(np.zeros(16, dtype=np.int64)) >= 1


def test_chunk_rule_lands_near_the_best_measured_count():
    """The rule against the sweep it was drawn through (profiles/archive/r05z_small_cloud_sweep_b16.jsonl: microseconds per synchronous
    cost+Jacobian evaluation on an MI355X with the number of chunks forced, 100k ... 6.4M points, 16 bins, this round's kernels;
    round 4's sweep of the earlier kernels: profiles/archive/r04i_small_cloud_sweep.jsonl): at the rule's count the measured time
    -- interpolated between the two nearest measured counts -- is within 5 % of the best measured one."""
    import ctypes
    import json
    import os

    from direct_visual_lidar_calibration_amd import _lib

    lib = _lib.load()
    lib.nidreg_debug_round_chunks.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "archive", "r05z_small_cloud_sweep_b16.jsonl")
    rows = [json.loads(line) for line in open(path)]
    assert len(rows) == 11
    worst = 0.0
    for r in rows:
        n, bins = r["points"], r["bins"]
        groups = max(1, bins // max(1, 256 // bins)) if bins < 256 else 256  # column groups: GW = max(1, 256 / B) columns each
        g = np.concatenate([[0], np.cumsum(np.full(groups, n // groups))]).astype(np.int64)
        chosen = lib.nidreg_debug_round_chunks(4, 256, g.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), groups)
        pts = sorted({(r["chunks"][k], r["us_per_eval"][k]) for k in r["chunks"] if k != "0"})
        xs, ys = np.array([p[0] for p in pts], dtype=float), np.array([p[1] for p in pts], dtype=float)
        at_rule = float(np.interp(chosen, xs, ys))
        best = float(ys.min())
        worst = max(worst, at_rule / best)
        assert at_rule <= 1.05 * best, (n, bins, chosen, at_rule, best)
    assert worst > 1.0  # (the table is real data: the rule is not the argmin of every row)


def test_estimate_camera_fov_runs_on_the_host_and_equals_the_oracle():
    """vlcal::estimate_camera_fov (estimate_fov.cpp:17-51) is set-up work on the host in the reference and here
    (nidreg_estimate_camera_fov: NelderMead<2> natively, on the device's scalar projection code compiled for the host -- no GPU
    involved): the oracle's value on every camera model and every BASELINE config camera, equal to the Python spelling of the
    same procedure, and the host projection itself against the oracle's."""
    import oracle_lib
    from direct_visual_lidar_calibration_amd import nid, synth
    from test_gpu_parity import CAMERAS

    rng = np.random.default_rng(4)
    for name, (model, intr, dist, W, H) in list(CAMERAS.items()) + list(synth.CONFIG_CAMERAS.items()):
        proj = nid.create_camera(model, intr, dist)
        fov = nid.estimate_camera_fov(proj, (W, H))
        assert abs(fov - oracle_lib.estimate_camera_fov(model, intr, dist, W, H)) <= 1e-12, name
        assert abs(fov - nid.estimate_camera_fov_py(proj, (W, H))) <= 1e-12, name
        p = rng.normal(size=(40, 3))
        p[:, 2] = np.abs(p[:, 2]) + 1.0
        uv, jac = proj.project(p, device=-1, jacobian=True)
        ref_uv, ref_jac = oracle_lib.project_jacobian(model, intr, dist, p)
        assert np.allclose(uv, ref_uv, rtol=1e-12, atol=1e-9), name
        assert np.allclose(jac, ref_jac, rtol=1e-9, atol=1e-8), name
