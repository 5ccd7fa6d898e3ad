#!/usr/bin/env python3
"""scene.npz (tools/make_scene_cache.py) -> one raw file for the C++ harnesses (tools/omp_pairs.cpp):
int64 N, int32 W, int32 H, double T_true[7], double intr[4], double dist[5], float32 xyz[N][3], float32 intensity[N], uint8 image[H][W].
Usage: dump_scene_raw.py scene.npz scene.raw"""
import struct
import sys

import numpy as np

z = np.load(sys.argv[1])
pts = np.ascontiguousarray(z["points"][:, :3], dtype=np.float32)
ints = np.ascontiguousarray(z["intensities"], dtype=np.float32)
img = np.ascontiguousarray(z["image_u8"], dtype=np.uint8)
intr = [float(v) for v in z["intrinsics"]][:4]
dist = ([float(v) for v in z["distortion"]] + [0.0] * 5)[:5]
with open(sys.argv[2], "wb") as f:
    f.write(struct.pack("<qii", pts.shape[0], img.shape[1], img.shape[0]))
    f.write(struct.pack("<7d", *[float(v) for v in z["T_true"]]))
    f.write(struct.pack("<4d", *intr))
    f.write(struct.pack("<5d", *dist))
    f.write(pts.tobytes())
    f.write(ints.tobytes())
    f.write(img.tobytes())
print("wrote", sys.argv[2], pts.shape[0], img.shape)
