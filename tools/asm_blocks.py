#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).

usage: asm_blocks.py file.s <substring of the mangled kernel name> [--dump LABEL]
Prints, for every basic block of the kernel, the number of VALU fp64 / other VALU / SALU / LDS / VMEM instructions,
so that the per-point instruction budget of the hot loop can be read off without a GPU (the PMC SQ_INSTS_VALU count
of a run divided by the points per lane agrees with the loop blocks' VALU totals).
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(m):
    if m.startswith(("v_cmp", "v_cmpx")):
        return "valu64" if "f64" in m else "valu"
    if m.startswith("v_"):
        if m.endswith("_f64") or "_f64_" in m or m.startswith(("v_cvt_f64", "v_fract_f64", "v_rcp_f64", "v_rsq_f64", "v_floor_f64")):
            return "valu64"
        return "valu"
    if m.startswith("s_"):
        return "salu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--dump" else None
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False:
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        blocks[cur].append(s.split()[0])
    tot = Counter()
    print(f"{'block':<14}{'n':>6}{'valu64':>8}{'valu':>7}{'salu':>7}{'lds':>6}{'vmem':>6}")
    for name, ins in blocks.items():
        c = Counter(classify(m) for m in ins)
        tot.update(c)
        if len(ins) >= 8:
            print(f"{name:<14}{len(ins):>6}{c['valu64']:>8}{c['valu']:>7}{c['salu']:>7}{c['lds']:>6}{c['vmem']:>6}")
        if dump and name == dump:
            mc = Counter(ins)
            for k, v in sorted(mc.items(), key=lambda kv: -kv[1]):
                print(f"    {k:<28}{v:>5}")
    print(f"{'total':<14}{sum(tot.values()):>6}{tot['valu64']:>8}{tot['valu']:>7}{tot['salu']:>7}{tot['lds']:>6}{tot['vmem']:>6}")


if __name__ == "__main__":
    main()
