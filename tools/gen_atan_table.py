#!/usr/bin/env python3
"""Regenerates csrc/nid_atan_table.hpp: atan(i / 256), i = 0 .. 256, correctly rounded to double (mpmath, 60 digits), as
hexadecimal floating literals.  Usage: gen_atan_table.py > direct_visual_lidar_calibration_amd/csrc/nid_atan_table.hpp"""
import mpmath as mp

mp.mp.dps = 60
N = 256
print("// nid_atan_table.hpp -- atan(i / 256), i = 0 .. 256, correctly rounded to double (tools/gen_atan_table.py, mpmath at")
print("// 60 digits).  fast_atan2 (nid_device.hpp): atan(t) = atan(t0) + atan((t - t0) / (1 + t t0)) with t0 = round(256 t) / 256.")
print("#pragma once\nnamespace nidreg {\nconstexpr int kAtanTableN = 256;\n#define NID_ATAN_TABLE_VALUES \\")
print(" \\\n".join("  %s," % float(mp.atan(mp.mpf(i) / N)).hex() for i in range(N + 1)))
print("}  // namespace nidreg")
