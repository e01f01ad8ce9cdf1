#!/usr/bin/env python3
"""Registers, scratch and occupancy of the library's kernels, from hipcc's resource-usage remarks.

    hipcc ... -Rpass-analysis=kernel-resource-usage ... 2> usage.txt
    python tools/kernel_resources.py usage.txt [substring ...]
"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2:]
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    print("VGPR AGPR SGPR scratch occ  LDS  kernel")
    for b, d in zip(blocks, dem):
        if want and not all(w in d for w in want):
            continue

        def g(k):
            m = re.search(re.escape(k) + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        vals = [g("VGPRs"), g("AGPRs"), g("SGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"),
                g("LDS Size [bytes/block]")]
        print("%4d %4d %4d %7d %3d %5d %s" % (*vals, d[:170]))


if __name__ == "__main__":
    main()
