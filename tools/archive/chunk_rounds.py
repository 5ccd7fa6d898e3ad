#!/usr/bin/env python3
"""How many workgroups do the chunk tables of a scene's pair subsets hold, against the slots of one round?  (CPU only.)
Pair p of k takes the points i = p mod k of the cached scene (tools/make_scene_cache.py), like tools/omp_pairs.cpp; the
column of a point is floor(intensity * bins).  Prints, per k, the table sizes under the rule of rounds 1-3 (CH = N / target,
every column split into ceil(count / CH) chunks) and the slots (512 WIDE histogram workgroups, 1024 gradient workgroups on
256 CUs).  Usage: chunk_rounds.py scene.npz"""
import sys

import numpy as np

z = np.load(sys.argv[1])
ints = z["intensities"].astype(np.float64)
N, B = len(ints), 256
col = np.minimum((ints * B).astype(np.int64), B - 1)


def old_rule(cnts, n, target, threads):
    ch = -(-n // target)
    ch = max(threads, -(-ch // threads) * threads)
    return int(sum(-(-c // ch) for c in cnts if c > 0)), ch


for k in (1, 2, 4, 8):
    tot_h = tot_g = 0
    lo, hi = 1 << 62, 0
    for p in range(k):
        c = np.bincount(col[p::k][: N // k], minlength=B)
        lo, hi = min(lo, c.min()), max(hi, c.max())
        tot_h += old_rule(c, N // k, max(1, 512 // k), 512)[0]
        tot_g += old_rule(c, N // k, max(1, 1024 // k), 256)[0]
    print(f"{k} pair(s) x {N // k} points: column populations {lo}..{hi}; histogram workgroups {tot_h} for 512 slots, gradient workgroups {tot_g} for 1024 slots")
