#!/usr/bin/env python3
"""Summarise a hipcc -save-temps gfx950 .s file: registers, occupancy, LDS and the instruction mix per
kernel.  Usage: isa_summary.py file.s [name-substring ...]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pats = sys.argv[2:]
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*\.end_amdhsa_kernel", s, flags=re.M | re.S):
    name, body = m.group(1), m.group(2)
    if pats and not any(p in name for p in pats):
        continue
    code = body.split(".amdhsa_kernel")[0]
    ops = collections.Counter(re.findall(r"^\s+([a-z][a-z_0-9]+)\s", code, flags=re.M))
    meta = body
    def g(k):
        r = re.search(r"\.amdhsa_" + k + r"\s+(\S+)", meta)
        return r.group(1) if r else "?"
    print(name)
    print("  next_free_vgpr", g("next_free_vgpr"), "next_free_sgpr", g("next_free_sgpr"), "accum_offset", g("accum_offset"), "lds", g("group_segment_fixed_size"), "scratch", g("private_segment_fixed_size"))
    print("  static instrs", sum(ops.values()))
    groups = collections.Counter()
    for k, v in ops.items():
        if k.startswith("v_") and "f64" in k: groups["valu_f64"] += v
        elif k.startswith("v_") and "f32" in k: groups["valu_f32"] += v
        elif k.startswith("v_"): groups["valu_int/other"] += v
        elif k.startswith("s_"): groups["salu"] += v
        elif k.startswith("ds_"): groups["lds"] += v
        elif k.startswith("global_") or k.startswith("flat_") or k.startswith("buffer_"): groups["vmem"] += v
        else: groups["other"] += v
    print("  groups", dict(groups))
    print("  top", ops.most_common(28))
