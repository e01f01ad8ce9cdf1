"""Exact solutions stated as plain NumPy callables f(t, xx, yy), shared by the script that makes the golden
vectors (tests/golden/make_golden.py, where they go into the reference's MMSCaseFromAnalytic) and by the
tests (where they go into the mirrored class)."""
import numpy as np

BASES = dict(
    cp_base=lambda t, xx, yy: 2.0 + np.exp(-t) * np.sin(xx + 0.3) * np.cos(yy),
    T_base=lambda t, xx, yy: 1.0 + 0.5 * np.exp(-0.5 * t) * np.sin(np.pi * xx) * np.sin(np.pi * yy),
    cl_base=lambda t, xx, yy: 1.5 + 0.3 * np.cos(t) * xx * yy,
    cd_base=lambda t, xx, yy: 0.5 + 0.1 * np.sin(t + xx - yy),
    cs_base=lambda t, xx, yy: 0.8 - 0.5 * t + 0.1 * xx * yy,
)
