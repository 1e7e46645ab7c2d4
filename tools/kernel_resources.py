#!/usr/bin/env python3
"""Register / LDS / scratch table of the kernels in an ISA listing made by `make asm`: tools/kernel_resources.py [csrc/wavefront.s ...]"""
import re, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or [os.path.join(ROOT, "distributed-path-tracer_amd", "csrc", f) for f in ("wavefront.s", "kernels.s")]
for f in files:
    txt = open(f).read()
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
        name = g("name")
        short = re.sub(r"^_ZN3ptx\d+", "", name)[:60]
        print(f"{short:62s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} spill v{g('vgpr_spill_count')}/s{g('sgpr_spill_count')} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")
