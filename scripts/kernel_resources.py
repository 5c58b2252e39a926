"""Tabulate the compiler's resource report of a HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage ... 2> file):
registers, AGPRs (spills of non-MFMA kernels), scratch bytes per lane, occupancy.  usage: python scripts/kernel_resources.py report.txt"""
import re
import subprocess
import sys


def rows(path):
    txt = open(path).read()
    out = []
    for blk in txt.split("Function Name: ")[1:]:
        name = blk.split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void pnode::", "")
        g = lambda k: int(re.search(k + r": (\S+)", blk).group(1))
        out.append((dem, g("TotalSGPRs"), g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
    return out


if __name__ == "__main__":
    for r in rows(sys.argv[1]):
        print("%-46s sgpr %3d vgpr %3d agpr %3d scratch %4d occupancy %d" % r)
