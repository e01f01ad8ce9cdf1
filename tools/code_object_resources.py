#!/usr/bin/env python3
"""Registers, scratch and LDS of the kernels inside a built libtedd_hip.so (read from the code object's metadata;
no recompilation).   python tools/code_object_resources.py [substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd", "lib", "libtedd_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    want = sys.argv[1:]
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        # (with no output file llvm-objcopy rewrites its input in place, which changes the library's hash -- the
        # identity the counter summaries under profiles/ are stamped with: the rewritten copy goes to the scratch directory)
        subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, LIB, os.path.join(d, "copy.so")])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    rows = []
    for e in re.split(r"\n\s*- \.agpr_count", notes)[1:]:
        get = lambda k: int(re.search(re.escape(k) + r":\s+(\d+)", e).group(1))
        rows.append((re.search(r"\.name:\s+(\S+)", e).group(1), get(".vgpr_count"), get(".sgpr_count"),
                     get(".private_segment_fixed_size"), get(".group_segment_fixed_size")))
    dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("VGPR SGPR scratch   LDS  kernel")
    for (n, v, s, sc, lds), name in zip(rows, dem):
        if all(w in name for w in want):
            print("%4d %4d %7d %5d  %s" % (v, s, sc, lds, name[:150]))


if __name__ == "__main__":
    main()
