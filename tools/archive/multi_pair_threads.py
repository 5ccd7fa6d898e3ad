import ctypes, json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from direct_visual_lidar_calibration_amd import _lib, nid, synth
total = 10_000_000
out = {}
for k in (1, 2, 4, 8):
    scenes = [synth.make_scene("pinhole_1080p", num_points=total // k, seed=100 + i, device="cuda:0") for i in range(k)]
    proj = nid.create_camera(scenes[0].model, scenes[0].intrinsics, scenes[0].distortion)
    costs = [nid.NIDCost(proj, s.image_f64, s.points, s.intensities, 256, target_blocks=(256 if k >= 4 else 0)) for s in scenes]
    rng = np.random.default_rng(3)
    poses = np.ascontiguousarray([synth.random_pose_near(scenes[0].T_camera_lidar_true, rng) for _ in range(40)])
    for c in costs:
        c.eval_batch(poses[:3])
    # one thread per pair, each running its own synchronous evaluations (the reference's OpenMP loop over pairs,
    # visual_camera_calibration.cpp:161): a barrier per pose like the optimiser imposes
    bar = threading.Barrier(k)
    reps = 6
    def work(c):
        for _ in range(reps):
            for x in poses:
                c(x)
                bar.wait()
    th = [threading.Thread(target=work, args=(c,)) for c in costs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = (time.perf_counter() - t0) / (reps * len(poses))
    out[f"threads_{k}"] = {"points_per_pair": total // k, "us_per_multi_eval": round(1e6 * dt, 2)}
    for c in costs: c.close()
print(json.dumps(out))
