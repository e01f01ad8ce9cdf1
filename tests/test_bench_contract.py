"""bench.py: the bookkeeping behind its numbers (CPU) and the shape of the line it prints (GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_array_passes_of_the_default_step():
    sweeps = bench.pick_sweeps(16384, (1.0 / 16384) ** 1.5, 1e-13)
    assert sweeps == [20, 7, 5]
    arrays, chains = bench.arrays_per_launch("strip", 1, sweeps, 10, 0)
    assert arrays["predict"] == 21 and arrays["forcing"] == 5 and arrays["cs_newton"] == 8
    # whole-mesh contexts do not store Y_T (nothing in the step reads it): one array fewer
    lazy, _ = bench.arrays_per_launch("strip", 1, sweeps, 10, 0, lazy_yt=True)
    assert lazy["predict"] == 20
    assert chains["solve_T"] == {"launches_per_solve": 2, "arrays_per_solve": 8}
    launches = {"forcing": 1, "predict": 1, "rhs_cl": 1, "rhs_cd": 1, "cs_newton": 1,
                "solve_T": 2, "solve_cl": 1, "solve_cd": 1}
    assert sum(arrays[k] * n for k, n in launches.items()) == 71
    # the predictor launch makes the forcing arrays itself on small slabs: no forcing launch, 5 more writes
    fused, _ = bench.arrays_per_launch("strip", 1, sweeps, 10, 1)
    assert "forcing" not in fused and fused["predict"] == 23
    # a warm-started T solve of 10 sweeps is one launch: b, cp*, T_n, T_{n-1} read, T_new written
    warm, chains = bench.arrays_per_launch("strip", 1, [10, 7, 5], 10, 0, warm=(True, False, False))
    assert chains["solve_T"] == {"launches_per_solve": 1, "arrays_per_solve": 5}
    # small slabs: the cl / cd solves read what their right-hand-side launches would have
    small, _ = bench.arrays_per_launch("strip", 1, [10, 5, 4], 10, 1, fused_rhs=True)
    assert small["solve_cl"] == 7 and small["solve_cd"] == 10


def test_sweep_counts_follow_the_contraction_bound():
    for N, expect in ((256, [10, 5, 4]), (4096, [15, 6, 5])):
        assert bench.pick_sweeps(N, (1.0 / N) ** 1.5, 1e-13) == expect


def test_traffic_is_quoted_only_from_counter_passes_of_the_library_that_is_timed(tmp_path):
    """roofline.traffic comes from a committed summary of rocprofv3 counter passes; a summary is used only if its
    stamp names the library being timed (or the same sources after a rebuild) -- otherwise the figure is null, with
    a note, never a number measured on other kernels."""
    cfg = {"mesh_N": 16384, "solver": "strip", "sweeps_per_launch": 10, "forcing": "arrays per time level"}
    now = bench.library_stamp()
    kb = 16385 * 16385 * 8 / 1024.0
    summ = {"config": dict(cfg), "stamp": {"library_sha256": "0" * 64, "source_sha256": "1" * 64, "git_rev": "abc"},
            "kernels": {"void tedd::k_march<tedd::PredictFn<-1, 1, -1>, 2, true>(...)":
                        {"FETCH_SIZE": {"mean": 13.2 * kb / 2, "n": 4}, "WRITE_SIZE": {"mean": 8.0 * kb, "n": 4}}}}
    path = tmp_path / "rXX_pmc_16384.json"
    path.write_text(json.dumps(summ))
    tr = bench.pmc_traffic("predict", cfg, profiles_dir=str(tmp_path))
    assert tr["bytes_per_launch"] is None and tr["source"] is None and "another stamp" in tr["note"]
    summ["stamp"]["library_sha256"] = now["library_sha256"]        # the very library file
    path.write_text(json.dumps(summ))
    tr = bench.pmc_traffic("predict", cfg, profiles_dir=str(tmp_path))
    assert tr["bytes_per_launch"] == pytest.approx(21.2 * 16385 * 16385 * 8) and tr["stamp"]["matched"] == "library file"
    summ["stamp"] = {"library_sha256": "0" * 64, "source_sha256": now["source_sha256"], "git_rev": "abc"}
    path.write_text(json.dumps(summ))
    tr = bench.pmc_traffic("predict", cfg, profiles_dir=str(tmp_path))
    if now["library_not_older_than_sources"]:                       # same sources, library rebuilt since
        assert tr["bytes_per_launch"] is not None and tr["stamp"]["matched"].startswith("sources")
    else:
        assert tr["bytes_per_launch"] is None
    assert bench.pmc_traffic("predict", dict(cfg, mesh_N=123), profiles_dir=str(tmp_path))["bytes_per_launch"] is None


def test_committed_pmc_summary_of_this_library_is_plausible():
    """If the tree holds the counter summary of its own library (the round's final profiles), every launch family's
    traffic is within a few per cent of its algorithmic bytes."""
    cfg = {"mesh_N": 16384, "solver": "strip", "sweeps_per_launch": 10, "forcing": "arrays per time level"}
    tr = bench.pmc_traffic("predict", cfg)
    if tr["bytes_per_launch"] is None:
        pytest.skip(tr["note"])
    algorithmic = 8.0 * 16385 * 16385 * 20      # 13 read, 7 written (Y_T is not stored on a whole-mesh context)
    assert algorithmic < tr["bytes_per_launch"] < 1.05 * algorithmic     # every array row is read once
    # (the right-hand sides march with their next row travelling into LDS: the two 16-byte granules a wave asks for
    # outside its own 64 columns are requests of their own, a few per cent of the launch's traffic)
    for fam, arrays, slack in (("rhs_cl", 7, 1.08), ("rhs_cd", 10, 1.08), ("forcing", 5, 1.05), ("cs_newton", 8, 1.05)):
        got = bench.pmc_traffic(fam, cfg)["bytes_per_launch"]
        assert 0.99 * arrays < got / (8.0 * 16385 * 16385) < slack * arrays, fam


@pytest.mark.gpu
def test_bench_prints_one_line_with_the_agreed_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mesh", "1024", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-9)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.0 < r["frac"] < 1.0
    assert "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    assert np.isfinite(d["config"]["error_H_after_run"])


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_launches_its_own_ranks(scaling):
    """`python bench.py --gpus 2` outside torch.distributed.run: the process starts the two ranks itself (gloo
    on the host twin here, RCCL on GPUs), passes rank 0's line through and returns the launcher's status."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_twin_cabi_parity import TWIN, build_twin
    build_twin()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mesh", "64", "--steps", "2",
                          "--warmup", "1", "--library", TWIN, "--backend", "gloo", "--device", "cpu", "--scaling", scaling],
                         capture_output=True, text=True, timeout=280, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["scaling"] == scaling
    assert d["steps"] == 2 and d["warmup"] == 1
    units = 2 * (2 if scaling == "weak" else 1)
    assert d["value"] == pytest.approx(units * 1e3 / (2 * d["ms_per_step"]), rel=1e-9)
    assert ("[0,2]x[0,1]" in d["config"]["workload"]) == (scaling == "weak")
    assert d["config"]["error_H_after_run"] < 1e-3


def test_bench_fails_when_a_rank_fails():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mesh", "64", "--steps", "1",
                          "--warmup", "0", "--library", "/nonexistent/libtedd.so", "--backend", "gloo", "--device", "cpu"],
                         capture_output=True, text=True, timeout=280, cwd=ROOT, env=env)
    assert out.returncode != 0
