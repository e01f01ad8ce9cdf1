"""The host-compiled twin of the device source shares the context code of the HIP library (buffer pool, valid-row
bookkeeping of the ghost zones, the ring of step records, the step sequencing): built with AddressSanitizer and
UndefinedBehaviorSanitizer (-O1 -g -fsanitize=address,undefined, the first undefined operation aborts), the
C-ABI parity tests, the slab tests (two and three gloo ranks) and the convergence-guard tests run once more on it,
in a child process started with the sanitizer runtime preloaded.  A report of either sanitizer fails this test."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FILES = ["test_twin_cabi_parity.py", "test_distributed_gloo.py", "test_convergence_guard.py"]


def _runtime():
    """The sanitizer runtime and the C++ runtime, in that order: the interpreter is a C program, and with the C++
    runtime loaded only later (by the twin) the sanitizer would not find the __cxa_throw it intercepts."""
    out = []
    for name in ("libasan.so", "libstdc++.so.6"):
        path = subprocess.check_output(["g++", "-print-file-name=" + name], text=True).strip()
        if not (os.path.isabs(path) and os.path.exists(path)):
            return None
        out.append(os.path.realpath(path))
    return ":".join(out)


@pytest.mark.skipif(_runtime() is None, reason="no AddressSanitizer runtime beside g++")
def test_twin_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    env = dict(os.environ)
    env["TEDD_TWIN_SANITIZE"] = "1"
    # (the runtime goes in front of whatever is preloaded already: it must come first in the library list)
    env["LD_PRELOAD"] = ":".join(p for p in (_runtime(), env.get("LD_PRELOAD", "")) if p)
    log = str(tmp_path / "asan")
    # leaks are not checked: the interpreter and torch hold allocations of their own until the process ends
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:exitcode=97:log_path=" + log
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1:exitcode=98:log_path=" + log
    env["OMP_NUM_THREADS"] = "2"
    run = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] +
                         [os.path.join(HERE, f) for f in FILES], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=1500)
    reports = [open(os.path.join(tmp_path, f)).read() for f in sorted(os.listdir(tmp_path)) if f.startswith("asan")]
    tail = run.stdout[-3000:] + run.stderr[-3000:]
    assert not reports, "sanitizer report:\n" + "\n".join(r[:4000] for r in reports)
    assert run.returncode == 0, tail
    assert " passed" in run.stdout and "failed" not in run.stdout, tail
    summary = os.environ.get("TEDD_SANITIZER_SUMMARY")
    if summary:      # (written for profiles/ when asked)
        with open(summary, "w") as fh:
            fh.write("g++ " + " ".join(["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                                        "-fno-omit-frame-pointer"]) + "  tests/cpu_twin/tedd_twin.cpp\n")
            fh.write("files: " + " ".join(FILES) + "\n")
            fh.write(run.stdout[-600:])
            fh.write("sanitizer reports: none\n")
