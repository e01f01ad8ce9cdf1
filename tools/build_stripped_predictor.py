"""Build a PROBE copy of the library whose marching predictor body does no arithmetic: PredictFn::node sums the values
the launcher hands it and stores the sums, so the launch keeps its loads, stores, arrays, segments and grid and loses
its 570 vector instructions per node.  Timed beside the real library inside one context
(AB_LIBRARY=tools/bin/libtedd_stripped_predictor.so python tools/ab_options.py march_occ=0) it gives the memory floor of
the predictor launch where its buffers really lie.  The copy computes nonsense: only its `predict` time means anything.
The sources are patched in a temporary directory; nothing under the package changes."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd")
OUT = os.path.join(ROOT, "tools", "bin", "libtedd_stripped_predictor.so")

ANCHOR = """    TEDD_HD Out node_vals(int li, int j, const Geo& ge, const St5* s, const double* cv) const {
        const bool in = is_interior(g, li, j);
"""
STRIPPED = """    TEDD_HD Out node_vals(int li, int j, const Geo& ge, const St5* s, const double* cv) const {
        {
            double a = ge.rhw + ge.rks;
            for (int k = 0; k < 4; ++k) a += (s[k].c + s[k].e) + (s[k].w + s[k].n) + s[k].s;
            for (int k = 0; k < 9; ++k) a += cv[k];
            return Out{a, a + 1.0, a + 2.0, a + 3.0, a + 4.0, a + 5.0, a + 6.0, a + 7.0};
        }
        const bool in = is_interior(g, li, j);
"""


def main():
    with tempfile.TemporaryDirectory() as tmp:
        # (the sources include ../../include/tedd.h: keep that layout)
        csrc = os.path.join(tmp, "pkg", "csrc")
        shutil.copytree(os.path.join(PKG, "csrc"), csrc)
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
        path = os.path.join(csrc, "tedd_nodes.hpp")
        text = open(path).read()
        if text.count(ANCHOR) != 1:
            sys.exit("the marching body of PredictFn is not where this script expects it")
        open(path, "w").write(text.replace(ANCHOR, STRIPPED))
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on",
               "-fPIC", "-shared", "-o", OUT, os.path.join(csrc, "tedd_hip.hip")]
        print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    print(OUT)


if __name__ == "__main__":
    main()
