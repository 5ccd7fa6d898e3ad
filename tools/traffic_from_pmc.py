#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from a tools/profile_pmc.sh output directory: per-kernel HBM bytes per
launch.  FETCH_SIZE / WRITE_SIZE are KiB from separate --pmc passes; on gfx950 FETCH_SIZE counts a wide
(16 B/lane) coalesced stream at half its bytes (MI355X_MICROARCH.md, HBM section), and the point
stream is >97 % of what these kernels read, so the corrected figure is 2*FETCH + WRITE.
Usage: traffic_from_pmc.py gpurun_out/pmc_<tag> profiles/<tag>_traffic.json points width height bins precision [camera]"""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
points, width, height, bins = (int(v) for v in sys.argv[3:7])
precision = sys.argv[7]
kernels = {}
cur = None
for line in open(f"{src}/summary.txt"):
    if not line.startswith(" "):
        m = re.match(r"nidreg::(k_\w+)", line.strip())
        cur = m.group(1) if m else None
        continue
    if cur is None:
        continue
    f = line.split()
    kernels.setdefault(cur, {})[f[0]] = float(f[2])  # every counter of the passes (per-launch means)
out = {}
for k, v in kernels.items():
    if ("FETCH_SIZE" in v and "WRITE_SIZE" in v) or "SQ_INSTS_VALU" in v:  # (a pass set without the byte counters still yields the instruction counts)
        out[k] = {}
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            out[k] = {"fetch_kib": v["FETCH_SIZE"], "write_kib": v["WRITE_SIZE"], "hbm_bytes_raw": int((v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024),
                      "hbm_bytes_corrected": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)}
        # wave-instruction counts per launch (the VALU-issue roof bench.py reports next to the HBM one)
        for name, key in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_INSTS_LDS", "lds_insts"), ("SQ_INSTS_SALU", "salu_insts"), ("SQ_INSTS_VMEM_RD", "vmem_rd_insts"),
                          ("SQ_ACTIVE_INST_VALU", "valu_active_quad_cycles"), ("GRBM_GUI_ACTIVE", "gui_active_cycles_all_xcd"), ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict_cycles"),
                          ("SQ_LDS_IDX_ACTIVE", "lds_idx_active_cycles")):
            if name in v:
                out[k][key] = v[name]
        # the stall counters of the same build: waiting / busy wave-cycles, LDS and L2 behaviour
        out[k]["counters"] = {n: v[n] for n in sorted(v) if n not in ("FETCH_SIZE", "WRITE_SIZE")}
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from direct_visual_lidar_calibration_amd import _lib  # noqa: E402

json.dump({"source": src, "kernel_build": _lib.stamp_or_refuse(), "workload": dict(points=points, width=width, height=height, bins=bins, precision=precision, camera=sys.argv[8] if len(sys.argv) > 8 else "pinhole_1080p"), "kernels": out},
          open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
