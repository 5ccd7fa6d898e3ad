#!/usr/bin/env python3
"""Regenerates csrc/nid_log_table.hpp: the 128-entry table of fast_log (nid_device.hpp).  A positive double x is split as
x = 2^k m with m in [0.6875, 1.375) (high word - 0x3fe60000: k = bits 20.., index = bits 13..19 of the difference); entry i holds
r_i = 1 / (centre of the i-th sub-interval) rounded to double, and -log(r_i) correctly rounded (mpmath, 60 digits), so that
log(m) = -log(r_i) + log1p(m r_i - 1) with |m r_i - 1| <= 0.004.
Usage: gen_log_table.py > direct_visual_lidar_calibration_amd/csrc/nid_log_table.hpp"""
import struct

import mpmath as mp

mp.mp.dps = 60
N = 128
OFF = 0x3FE60000
rows = []
for i in range(N):
    lo_hi = OFF + (i << 13)          # high word of the sub-interval's first double
    hi_hi = OFF + ((i + 1) << 13)    # ... and of the next one's
    lo = struct.unpack("<d", struct.pack("<Q", lo_hi << 32))[0]
    hi = struct.unpack("<d", struct.pack("<Q", hi_hi << 32))[0]
    c = (mp.mpf(lo) + mp.mpf(hi)) / 2
    r = float(1 / c)
    rows.append((r, float(-mp.log(mp.mpf(r)))))
print("// nid_log_table.hpp -- fast_log's table (tools/gen_log_table.py, mpmath at 60 digits): for the i-th of 128 sub-intervals of")
print("// [0.6875, 1.375) the pair r_i = 1 / centre (a double) and -log(r_i) correctly rounded.")
print("#pragma once\nnamespace nidreg {\nconstexpr int kLogTableN = 128;\n#define NID_LOG_TABLE_VALUES \\")
print(" \\\n".join("  %s, %s," % (r.hex(), l.hex()) for r, l in rows))
print("}  // namespace nidreg")
