#!/usr/bin/env python3
"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks: registers, scratch, occupancy and LDS per kernel.
  python scripts/kernel_resources.py file.res [...]   (file.res = stderr of hipcc ... -Rpass-analysis=kernel-resource-usage)"""
import re
import sys

for f in sys.argv[1:]:
    txt = open(f).read()
    for blk in txt.split("Function Name: ")[1:]:
        name = re.sub(r"_ZN5pnode\d+", "", blk.split()[0])[:44]
        get = lambda k: re.search(k + r": (\d+)", blk).group(1)
        print("%-46s VGPR %s AGPR %s scratch %s occ %s LDS %s" % (name, get("VGPRs"), get("AGPRs"), get(r"ScratchSize \[bytes/lane\]"),
                                                               get(r"Occupancy \[waves/SIMD\]"), get(r"LDS Size \[bytes/block\]")))
