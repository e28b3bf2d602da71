"""Registers / spills / scratch of every kernel in a `hipcc -S --cuda-device-only` listing:  python tools/probes/regs.py x.s [filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if flt in name:
        print(f"{name[:110]:110s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} lds {g('group_segment_fixed_size')}")
