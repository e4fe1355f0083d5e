#!/usr/bin/env python3
"""Dev tool: registers / scratch / occupancy / LDS of every kernel, from `hipcc -Rpass-analysis=kernel-resource-usage`
(`make -C opticalraytrace_amd/csrc resource-usage 2> ru.txt`).   usage: python tools/resource_usage.py ru.txt [filter]"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: Function Name: ", t)[1:]
names = [b.split()[0] for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scr':>5} {'occ':>4} {'LDS':>6}  kernel")
for b, d in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    d = re.sub(r"\(anonymous namespace\)::", "", d).split("(")[0].replace("void ", "")
    if flt in d:
        cols = [g(" VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("ScratchSize .bytes/lane."), g("Occupancy .waves/SIMD."),
                g("LDS Size .bytes/block.")]
        print("{:>5} {:>5} {:>5} {:>5} {:>4} {:>6}  {}".format(*cols, d))
