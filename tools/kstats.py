#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in a gfx950 assembly file (hipcc --save-temps=obj ... -> *.s)."""
import re
import sys
txt = open(sys.argv[1]).read()
for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size: *\d+", txt, flags=re.S):
    f = dict(re.findall(r"\.(\w+): +(\S+)", blk))
    name = re.sub(r"^_ZN6dabhip12_GLOBAL__N_1\d+", "", f.get("name", "?"))[:48]
    print("%-50s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s spill_v %3s" % (name, f.get("vgpr_count"), f.get("agpr_count"), f.get("sgpr_count"),
          f.get("group_segment_fixed_size"), f.get("private_segment_fixed_size"), f.get("vgpr_spill_count")))
