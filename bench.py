#!/usr/bin/env python3
"""Benchmark of the hot path: predictor-corrector time steps of the MMSCaseExpSin problem.

    python bench.py --gpus N --steps K --warmup W

Workload (all ranks together): the ExpSin MMS problem on the synthetic 16384 x 16384-cell mesh
(4096x the cells of the finest convergence-study mesh, N=256), notebook model constants,
eta = 50, dt = h^1.5, state = exact solution at t = 0 (generated on the device).  With N > 1 the
mesh rows are split into N slabs, one ghost-row exchange per step over RCCL:
  --scaling strong (default)  the same 16384^2 mesh for every N; value = steps / second
  --scaling weak              every rank keeps a 16384-row slab: the domain is [0, N] x [0, 1] with the
                              same cell size, time step and sweep counts; value = N x steps / second
                              (time steps of 16384^2-cell slabs per second, all ranks together)
A "step" is one call of the predictor-corrector integrator (reference
P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple.step).

`python bench.py --gpus N` with N > 1 launches its own ranks (one process per GPU through
torch.distributed.run on 127.0.0.1) before anything touches a GPU, relays rank 0's JSON line and exits
with the launcher's status; started by torch.distributed.run itself (RANK/WORLD_SIZE set) it is a rank.

The JSON line also carries
  roofline      HBM roofline of the kernel family with the largest share of the timed region
                (algorithmic bytes per launch / average launch duration from HIP events)
  study_mesh    steps/s on the finest convergence-study mesh (N=256), rank 0 at --gpus 1 only
  cpu_baseline  the CPU oracle (NumPy/SciPy restatement of the reference, 1 core) timed on the
                N=256 mesh, rank 0 at --gpus 1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (8 TB/s)

NB_A = dict(R0=8.3144621, Ea=1.60217662e-19, K1=1e-3, K2=1e-3, K3=1e-3, K4=1e-3, DT=1e-3,
            Dl_max=1e-5, phi_l=1e-5, gamma_T=1e-9, Kd=1e-2, Sd=1, Dd_max=1e-6, phi_d=1e-5,
            phi_T=1.60217662e-19 / 8.3144621, r_sp=5e-2, T_ref=300)

# Distinct full-grid arrays read + written per launch of each kernel family (algorithmic traffic:
# 8 bytes per node per array; stencil neighbours, tile halos and 1-D tables not counted).
def arrays_per_launch(solver, cache_forcing, sweeps, hmax, fuse_forcing=1, warm=(False, False, False), fused_rhs=False,
                      lazy_yt=False):
    """-> ({family: arrays per launch (chain families: average over the chain's launches)},
           {chain family: {"launches_per_solve", "arrays_per_solve"}})
    warm[k]: the solve of system k starts from the increment of the step before (its first launch also
    reads the unknown of the step before); fused_rhs: single-launch cl / cd solves evaluate their right-hand
    side themselves (no rhs_cl / rhs_cd launch: the solve reads what that launch would have); lazy_yt: the
    predictor does not store Y_T (whole-mesh contexts with cached forcing and the register-resident solve: nothing
    in the step reads it, tedd_get_residuals remakes it)."""
    a = {}
    if cache_forcing:
        a["forcing"] = 5        # writes fcp (cell averages), fT, fcl, fcd, fcs of the new time level
        a["predict"] = 19       # reads cp,T,cl,cd,cs, the 5 forcing arrays at t0, fcp and fcs at t1; writes Y_T,Y_cl,Y_cd,cp*,cs*, and the t0 parts of the correctors (2)
        f = 1                   # the forcing array each rhs kernel reads
        if fuse_forcing:
            # the predictor evaluates the t1 forcing itself: no forcing launch, it reads the 5 arrays
            # of t0 only and writes the 5 of t1
            del a["forcing"]
            a["predict"] += 5 - 2
    else:
        a["predict"] = 14       # reads cp,T,cl,cd,cs + cached fcp(t0); writes Y_T,Y_cl,Y_cd,cp*,cs*,fcp(t1), corrector parts (2)
        f = 0
    a["cs_newton"] = 8          # reads the 2 corrector parts, cs, T_new, cl_new, cd_new; writes cp_new, cs_new
    if solver == "strip":
        # + writes b = rhs/diag of the T system (no rhs_T launch), reading fT(t1) unless it is evaluated here
        a["predict"] += 1 + (f if not (cache_forcing and fuse_forcing) else 0)
        if lazy_yt and cache_forcing:
            a["predict"] -= 1
    w = 1 if solver == "strip" else 2           # strip: writes b = rhs/diag; others: rhs and delta_1
    a["rhs_T"] = 3 + f + w      # reads cp*, T, Y_T
    a["rhs_cl"] = 5 + f + w     # reads cp*, T, T_new, cl, Y_cl
    a["rhs_cd"] = 8 + f + w     # reads cp*, cs*, T, T_new, cl, cl_new, cd, Y_cd
    a["jacobi_T"] = 4           # reads delta, rhs, cp*; writes delta (final sweep: +T, writes T_new)
    a["jacobi_cl"] = 5          # reads delta, rhs, cp*, T; writes delta
    a["jacobi_cd"] = 7          # reads delta, rhs, cp*, T, cl, cs*; writes delta
    chains = {}
    coef = {"T": 1, "cl": 2, "cd": 4}           # coefficient fields of the rows: cp* | cp*,T | cp*,T,cl,cs*
    for k, var in enumerate(("T", "cl", "cd")):
        L = max(1, -(-(sweeps[k] - 1) // hmax))
        if solver == "strip":
            # every launch reads b and the coefficient fields and writes one array; launches after the
            # first also read the iterate; the last also reads the lagged unknown
            arr = sum(1 + coef[var] + (1 if l > 0 else 0) + (1 if l == L - 1 else 0) + 1 for l in range(L))
            if warm[k]:
                arr += 1 + (0 if L == 1 else 1)     # u_{n-1}, and the lagged unknown unless the launch reads it anyway
            if fused_rhs and L == 1 and var != "T":
                arr = a["rhs_" + var] - 1 + 1 + (1 if warm[k] else 0)   # the rhs launch's reads, one array written
        else:
            continue
        chains["solve_" + var] = {"launches_per_solve": L, "arrays_per_solve": arr}
        a["solve_" + var] = arr / L
    return a, chains


# kernel-name fragment of each launch family in the rocprofv3 counter files
FAMILY_KERNELS = {"predict": ["PredictFn"], "forcing": ["ForcingFn"], "cs_newton": ["k_cs_newton"],
                  "rhs_T": ["NewtonRhsFn<1,"], "rhs_cl": ["NewtonRhsFn<2,"], "rhs_cd": ["NewtonRhsFn<3,"],
                  "solve_T": ["k_solve_strip<1,"], "solve_cl": ["k_solve_strip<2,"], "solve_cd": ["k_solve_strip<3,"],
                  "jacobi_T": ["JacobiFn<1,"], "jacobi_cl": ["JacobiFn<2,"], "jacobi_cd": ["JacobiFn<3,"]}


def library_stamp(library=None):
    """What identifies the kernels a measurement ran: SHA-256 of the library file and of the sources it is built
    from (csrc/ and include/tedd.h, in name order)."""
    import hashlib
    pkg = os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd")
    lib = library or os.path.join(pkg, "lib", "libtedd_hip.so")
    srcs = sorted(os.path.join(pkg, "csrc", f) for f in os.listdir(os.path.join(pkg, "csrc")))
    srcs.append(os.path.join(ROOT, "include", "tedd.h"))
    hs = hashlib.sha256()
    for path in srcs:
        with open(path, "rb") as fh:
            hs.update(os.path.basename(path).encode() + b"\0" + fh.read())
    with open(lib, "rb") as fh:
        hl = hashlib.sha256(fh.read()).hexdigest()
    fresh = all(os.path.getmtime(lib) >= os.path.getmtime(p) for p in srcs)
    return {"library_sha256": hl, "source_sha256": hs.hexdigest(), "library_not_older_than_sources": fresh}


def pmc_traffic(fam, config, library=None, profiles_dir=None):
    """HBM bytes per launch of a family from a committed PMC summary of this same configuration AND this same
    library (profiles/*pmc_<N>.json, written by tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE
    passes): (2 x FETCH_SIZE + WRITE_SIZE) KB -- on gfx950 FETCH_SIZE tallies the 128-byte requests
    of coalesced streaming reads at 64 bytes.  A summary is used only if its stamp names the library file that is
    being timed (or, after a rebuild, the same sources with a library not older than they are); otherwise the
    traffic is unknown: {"bytes_per_launch": None, "note": why}."""
    import glob
    now = library_stamp(library)
    seen = 0
    pdir = profiles_dir or os.path.join(ROOT, "profiles")
    for path in sorted(glob.glob(os.path.join(pdir, f"*pmc_{config['mesh_N']}.json")), reverse=True):
        with open(path) as fh:
            summ = json.load(fh)
        if any(summ.get("config", {}).get(k) != config[k] for k in ("mesh_N", "solver", "sweeps_per_launch", "forcing")):
            continue
        stamp = summ.get("stamp") or {}
        same_lib = stamp.get("library_sha256") == now["library_sha256"]
        same_src = stamp.get("source_sha256") == now["source_sha256"] and now["library_not_older_than_sources"]
        if not (same_lib or same_src):
            seen += 1
            continue
        tot, launches = 0.0, 0
        for name, c in summ["kernels"].items():
            if any(frag in name for frag in FAMILY_KERNELS.get(fam, [])) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                n = c["FETCH_SIZE"]["n"]
                tot += (2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024.0 * n
                launches += n
        if launches:
            return {"bytes_per_launch": tot / launches, "source": os.path.relpath(path, ROOT),
                    "stamp": {"library_sha256": stamp.get("library_sha256"), "git_rev": stamp.get("git_rev"),
                              "matched": "library file" if same_lib else "sources (library rebuilt)"}}
    return {"bytes_per_launch": None, "source": None,
            "note": f"no counter summary under profiles/ was taken with this library ({seen} of this configuration "
                    f"carry another stamp); library sha256 {now['library_sha256'][:16]}"}


def consts_vec(d):
    from tedd import _ffi
    return [float(d[k]) for k in _ffi.CONST_NAMES]


def pick_sweeps(N, dt, jacobi_tol, x=None, y=None, consts=None):
    """Sweep counts the library would pick from a zero iterate (same contraction bound), needed up-front to size
    the ghost zone.  Default: the uniform N x N mesh with the notebook constants."""
    from tedd.distributed import a_priori_sweeps
    u = np.linspace(0.0, 1.0, N + 1)
    return a_priori_sweeps(u if x is None else x, u if y is None else y,
                           consts_vec(NB_A) if consts is None else consts, dt, jacobi_tol)


def study_mesh_run(N=256, steps=400, warmup=20):
    from tedd import _ffi
    x = np.linspace(0, 1, N + 1)
    ctx = _ffi.Context(x, x, consts_vec(NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_EXPSIN)
    dt = (1.0 / N) ** 1.5
    ctx.set_state_exact(0.0)
    ctx.advance(0.0, dt, warmup, want_errors=False)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.advance(warmup * dt, dt, steps, want_errors=False)
    ctx.sync()
    el = time.perf_counter() - t0
    stats = ctx.get_stats()
    ctx.close()
    return {"workload": f"ExpSin N={N}, dt=h^1.5, eta=50, device-resident steps", "steps": steps,
            "steps_per_s": steps / el, "ms_per_step": 1e3 * el / steps, "jacobi_sweeps": stats["sweeps"]}


def other_configs_run():
    """The smaller configurations of the study: latency of one full step on the mid-resolution mesh, and the
    whole run of the stress case (SlowlyChangingPeaks_Fast1e1, N=256, Tf=1: 4096 steps with the error norms of
    the harness after every step)."""
    from tedd import _ffi
    out = {}
    N = 32
    x = np.linspace(0, 1, N + 1)
    ctx = _ffi.Context(x, x, consts_vec(NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_EXPSIN)
    dt = (1.0 / N) ** 1.5
    ctx.set_state_exact(0.0)
    ctx.advance(0.0, dt, 20, want_errors=False)
    ctx.sync()
    lat = []
    t = 20 * dt
    for _ in range(200):
        t0 = time.perf_counter()
        ctx.step_pc(t, dt)
        ctx.sync()
        lat.append(time.perf_counter() - t0)
        t += dt
    ctx.close()
    out["expsin_N32_one_step"] = {"latency_us_median": 1e6 * float(np.median(lat)),
                                  "workload": "ExpSin N=32, one predictor-corrector step, enqueue to completion"}
    N = 256
    x = np.linspace(0, 1, N + 1)
    nb_b = dict(NB_A, Dl_max=8.01e-4, Dd_max=2.46e-6)
    ctx = _ffi.Context(x, x, consts_vec(nb_b), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_PEAKS, mms_params=(1.0, 10.0))
    dt = (1.0 / N) ** 1.5
    ctx.set_state_exact(0.0)
    t0 = time.perf_counter()
    series = ctx.advance(0.0, dt, 4096, integrator=1)
    el = time.perf_counter() - t0
    ctx.close()
    assert np.isfinite(series).all()
    out["peaks_fast1e1_N256_full_run"] = {"steps": 4096, "seconds": el, "steps_per_s": 4096 / el,
                                          "workload": "SlowlyChangingPeaks_Fast1e1 N=256, Tf=1, dt=h^1.5, error norms "
                                                      "after every step (the notebook's finest level)"}
    return out


def _oracle_steps(N, steps, reference_call_counts):
    """steps/s of the NumPy/SciPy oracle on the N-mesh.  reference_call_counts: evaluate the forcing terms as
    often as the reference's step does (the cell-averaged fcp and fcs five times, fT, fcl, fcd three times per
    step, src/prob1base.py:2953-2965, 3117-3149, 3631-3702) and the three Newton residuals, instead of once
    per time level and no residuals."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tedd_oracle as orc
    model = orc.Model(NB_A, 2)
    pb = orc.Problem(orc.Mesh.uniform(N), model, orc.make_case("expsin", model), "reg", 50.0)
    dt = (1.0 / N) ** 1.5
    S = pb.exact_state(0.0)
    t = 0.0
    t0 = time.perf_counter()
    f_prev = pb.forcing(t)
    for _ in range(steps):
        if reference_call_counts:
            for _rep in range(2):           # fT, fcl, fcd: 3 evaluations per step; fcp, fcs: 5
                pb.forcing(t + dt)
            for _rep in range(2):
                orc.forcing_pointwise(pb.case, model, t + dt, pb.mesh.X, pb.mesh.Y, "reg", 50.0, which=("fcs",))
                orc.cell_average(lambda P, Q: orc.forcing_pointwise(pb.case, model, t + dt, P, Q, "reg", 50.0,
                                                                    which=("fcp",))["fcp"], pb.mesh)
            S, _ = orc.step_pc(pb, S, t, dt, want_residual=True)
        else:
            f_next = pb.forcing(t + dt)
            S, _ = orc.step_pc(pb, S, t, dt, forcing0=f_prev, forcing1=f_next, want_residual=False)
            f_prev = f_next
        t += dt
    return steps / (time.perf_counter() - t0)


_TWIN_TIMER = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from tedd import _ffi
_ffi.use_library(sys.argv[2])
N, steps = int(sys.argv[3]), int(sys.argv[4])
consts = [float(v) for v in sys.argv[5].split(",")]
x = np.linspace(0, 1, N + 1)
ctx = _ffi.Context(x, x, consts, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0, mms_case=_ffi.MMS_EXPSIN)
dt = (1.0 / N) ** 1.5
ctx.set_state_exact(0.0)
ctx.advance(0.0, dt, 1, want_errors=False)
t0 = time.perf_counter()
ctx.advance(dt, dt, steps, want_errors=False)
print(steps / (time.perf_counter() - t0))
"""


def _twin_steps(N, steps, threads):
    """steps/s of the host-compiled twin of the device source (tests/cpu_twin: the same node functors and step
    sequencing under a loop backend, Jacobi solves, OpenMP over the rows) in a child process that never touches
    the GPU, with `threads` OpenMP threads.  None if the twin library is not there."""
    import subprocess
    twin = os.path.join(ROOT, "tests", "cpu_twin", "libtedd_twin.so")
    if not os.path.exists(twin):
        return None
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="false")
    pkg = os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd")
    r = subprocess.run([sys.executable, "-c", _TWIN_TIMER, pkg, twin, str(N), str(steps),
                        ",".join(repr(float(NB_A[k])) for k in CONST_KEYS)],
                       capture_output=True, text=True, env=env, timeout=120)
    if r.returncode != 0:
        return None
    return float(r.stdout.strip().splitlines()[-1])


CONST_KEYS = ("R0", "Ea", "K1", "K2", "K3", "K4", "DT", "Dl_max", "phi_l", "gamma_T", "Kd", "Sd", "Dd_max",
              "phi_d", "phi_T", "r_sp", "T_ref")


def cpu_baseline_run(N=256, steps=5):
    """The CPU side of the comparison on the host cores of this box, on a bounded sample (a few seconds in all).
    `value` is the NumPy/SciPy oracle with the forcing evaluated once per time level (the fastest honest
    single-process NumPy/SciPy form of the reference's algorithm); `variants` adds the same oracle with the
    reference's call counts, and the host-compiled C++ twin on one core and on all cores of this GPU's share."""
    cores_all = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
    v = _oracle_steps(N, steps, False)
    variants = {
        "numpy_scipy_port": {"steps_per_s": v, "cores": 1, "steps": steps,
                             "what": "oracle/tedd_oracle.py, spsolve, forcing once per time level, no residuals"},
        "numpy_scipy_port_reference_call_counts": {
            "steps_per_s": _oracle_steps(N, max(2, steps // 2), True), "cores": 1, "steps": max(2, steps // 2),
            "what": "the same with the forcing terms evaluated as often as the reference's step does and the Newton "
                    "residuals computed (the reference's per-call Grid() reconstruction is not reproduced)"},
    }
    for name, th in (("cpp_twin_1_core", 1), ("cpp_twin_all_cores", cores_all)):
        r = _twin_steps(N, 40 if th == 1 else 200, th)
        if r is not None:
            variants[name] = {"steps_per_s": r, "cores": th, "steps": 40 if th == 1 else 200,
                              "what": "tests/cpu_twin: the device source compiled for the host (g++ -O2 -fopenmp), "
                                      "Jacobi solves"}
    return {"value": v, "unit": "time-steps/s", "cores": 1, "kind": "port",
            "sample": f"{steps} predictor-corrector steps of ExpSin on the N={N} mesh (finest convergence-study "
                      f"mesh) with the NumPy/SciPy oracle (spsolve), forcing evaluated once per time level",
            "host_cpu_count": os.cpu_count(), "omp_num_threads_env": os.environ.get("OMP_NUM_THREADS"),
            "variants": variants}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--settle", type=int, default=8,
                    help="untimed steps before the warm-up: the library starts its Jacobi solves from the history of "
                         "the steps before, which a run has from its sixth step on (studies take thousands)")
    ap.add_argument("--mesh", type=int, default=16384, help="N of the (N+1)^2-node synthetic mesh")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: one N x N mesh split over the ranks; weak: an N-row slab per rank")
    ap.add_argument("--no-overlap", action="store_true",
                    help="several ranks: wait for the ghost rows after every step instead of letting them travel under "
                         "the next step's first launch")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for CPU rehearsals)")
    ap.add_argument("--device", choices=["cuda", "cpu"], default="cuda",
                    help="cpu: drive a host-compiled library given with --library (rehearsal of the multi-rank path)")
    ap.add_argument("--library", default=None, help="path of the C-ABI library (default: the HIP build)")
    ap.add_argument("--jacobi-tol", type=float, default=1e-13)
    ap.add_argument("--fixed-sweeps", action="store_true",
                    help="fix the Jacobi sweep counts at the a-priori bound instead of the library's stopping rule")
    ap.add_argument("--no-extras", action="store_true", help="skip the N=256 study-mesh and CPU-baseline legs")
    ap.add_argument("--solver", choices=["strip", "sweeps"], default="strip",
                    help="Newton solve: register-resident chain | one launch per Jacobi sweep")
    ap.add_argument("--hmax", type=int, default=10, help="Jacobi sweeps fused per launch of a solve chain")
    ap.add_argument("--cache-forcing", type=int, default=1,
                    help="1: forcing arrays evaluated once per time level; 0: inside the step kernels")
    ap.add_argument("--fuse-forcing", type=int, default=-1,
                    help="with --cache-forcing 1: the predictor launch evaluates the forcing arrays of t1 itself "
                         "(-1: the library's rule, slabs of at most 4 M nodes)")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (repeatable)")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a child launcher (nothing
    in this process has touched a GPU or imported torch), pass its output through and return its status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    from tedd import _ffi
    from tedd.distributed import SlabStepper, ghost_rows_needed, partition_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    on_gpu = args.device == "cuda"
    if args.library:
        _ffi.use_library(args.library)
    if on_gpu:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")
    if world > 1:
        if on_gpu and args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)
    ranks_seen = dist.get_world_size() if world > 1 else 1
    backend_name = dist.get_backend() if world > 1 else None

    N = args.mesh
    weak = args.scaling == "weak"
    rows_global = (N * world if weak else N) + 1     # weak: the domain grows along x with the ranks, same cell size
    dt = (1.0 / N) ** 1.5
    x = np.linspace(0.0, float(world) if weak else 1.0, rows_global)
    y = np.linspace(0.0, 1.0, N + 1)
    sweeps_cap = pick_sweeps(N, dt, args.jacobi_tol, x=x, y=y)   # a-priori counts: they size the ghost zone
    ghost = ghost_rows_needed(sweeps_cap, library_chosen=not args.fixed_sweeps)
    slab = partition_rows(rows_global, world, rank, ghost)
    ctx = _ffi.Context(x, y, consts_vec(NB_A), model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0,
                       mms_case=_ffi.MMS_EXPSIN, row_lo=slab.row_lo, row_hi=slab.row_hi, ghost=slab.ghost,
                       device=local_rank if on_gpu else -1)
    stepper = SlabStepper(ctx, slab, device, overlap=not args.no_overlap)
    ctx.set_option("strip_solve", 1 if args.solver == "strip" else 0)
    ctx.set_option("cache_forcing", args.cache_forcing)
    ctx.set_option("fuse_forcing", args.fuse_forcing)
    if args.fuse_forcing < 0:
        args.fuse_forcing = int(ctx.nrows * (N + 1) <= (1 << 22))
    ctx.set_option("strip_hmax", args.hmax)
    for kv in args.opt:
        name, value = kv.split("=")
        ctx.set_option(name, float(value))
    ctx.set_state_exact(0.0)
    kw = dict(jacobi_sweeps=sweeps_cap if args.fixed_sweeps else None, jacobi_tol=args.jacobi_tol)

    def barrier():
        if on_gpu:
            torch.cuda.synchronize(device)
        else:
            ctx.sync()
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize(device)

    t = 0.0
    for _ in range(max(args.settle, 0) + args.warmup):
        stepper.step_pc(t, dt, **kw)
        t += dt
    ctx.profile_enable(True)
    stepper.finish()      # (the timed region posts and completes the ghost-row exchanges of exactly its own steps)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stepper.step_pc(t, dt, **kw)
        t += dt
    stepper.finish()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    stats = ctx.get_stats()
    stepper.check_converged()
    sweeps = stats["sweeps"]
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # sanity: the run must still track the exact solution
    norms = ctx.error_norms(t)
    if world > 1:
        tn = torch.tensor(norms, dtype=torch.float64, device=device)
        dist.all_reduce(tn)
        norms = tn.cpu().numpy()
    err_H = float(np.sqrt(np.sum(norms[:5])))

    n_nodes_local = (slab.row_hi - slab.row_lo) * (N + 1)
    arrays, chains = arrays_per_launch(args.solver, args.cache_forcing, sweeps, args.hmax, args.fuse_forcing,
                                       warm=stats["warm_started"],
                                       fused_rhs=args.solver == "strip" and "rhs_cl" not in prof,
                                       lazy_yt=args.solver == "strip" and world == 1)

    def gbps(fam):
        per_launch_s = prof[fam]["total_ms"] / prof[fam]["launches"] * 1e-3
        return 8.0 * n_nodes_local * arrays.get(fam, 0) / per_launch_s / 1e9

    fam = max(prof, key=lambda k: prof[k]["total_ms"]) if prof else None
    roofline = None
    if fam is not None:
        per_launch_ms = prof[fam]["total_ms"] / prof[fam]["launches"]
        roofline = {"bound": "hbm", "kernel": fam, "achieved": gbps(fam), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbps(fam) / HBM_PEAK_GBS, "traffic": None,
                    "arrays_per_launch": arrays.get(fam, 0), "avg_launch_ms": per_launch_ms,
                    "launches": prof[fam]["launches"], "nodes_per_launch": n_nodes_local}
    kernels = {k: {"launches": v["launches"], "avg_ms": v["total_ms"] / v["launches"],
                   "share": v["total_ms"] / (1e3 * elapsed), "GBps": gbps(k)} for k, v in prof.items()}
    # arrays per step = sum over the launches the step actually made
    arrays_step = sum(arrays.get(k, 0) * v["launches"] for k, v in prof.items()) / args.steps
    nodes_all = rows_global * (N + 1)
    step_gbs = 8.0 * nodes_all * arrays_step * args.steps / elapsed / 1e9

    forcing_mode = ("closed form inside the step kernels" if not args.cache_forcing else
                    "arrays per time level, made by the predictor launch" if args.fuse_forcing else "arrays per time level")
    cfg_key = {"mesh_N": N, "solver": args.solver, "sweeps_per_launch": args.hmax, "forcing": forcing_mode,
               "jacobi_sweeps": list(sweeps)}
    if roofline is not None and world == 1:   # the committed counter passes are whole-mesh launches on one GPU
        tr = pmc_traffic(fam, cfg_key, args.library)
        roofline["traffic"] = tr["bytes_per_launch"]
        roofline["traffic_source"] = tr["source"]
        if tr["bytes_per_launch"] is None:
            roofline["traffic_note"] = tr["note"]
        else:
            roofline["traffic_stamp"] = tr["stamp"]
        roofline["algorithmic_bytes_per_launch"] = 8.0 * n_nodes_local * arrays.get(fam, 0)
    units = args.steps * (world if weak else 1)
    mesh_txt = (f"{rows_global - 1}x{N}-cell mesh on [0,{world}]x[0,1] ({nodes_all} nodes, a {N}-row slab per rank)"
                if weak else f"{N}x{N}-cell mesh ({nodes_all} nodes)")
    out = {
        "metric": "time-steps/sec, MMSCaseExpSin predictor-corrector (reghcstriple), synthetic 4096x mesh",
        "value": units / elapsed, "unit": "time-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "settle_steps": max(args.settle, 0), "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "ranks_seen": ranks_seen, "backend": backend_name,
        "config": {"workload": f"MMSCaseExpSin reghcstriple, {mesh_txt}, dt=h^1.5, "
                               f"eta=50, notebook model constants, 1 PC x 1 Newton step, 5 cs-Newton iterations",
                   "mesh_N": N, "jacobi_sweeps": list(sweeps), "jacobi_sweeps_a_priori": sweeps_cap,
                   "jacobi_tol": args.jacobi_tol,
                   "solver": args.solver, "sweeps_per_launch": args.hmax, "solve_chains": chains,
                   "forcing": forcing_mode,
                   "value_counts": ("time steps of the whole mesh" if not weak else
                                    "time steps of one rank's slab, summed over the ranks"),
                   "parallelism": f"row slabs x{world}, ghost={slab.ghost}" + (", ghost rows travel under the next step's predictor"
                                                                          if stepper.overlap else ""),
                   "field_placement": ctx.placement,
                   "error_H_after_run": err_H},
        "roofline": roofline,
        "step_hbm": {"algorithmic_arrays_per_step": arrays_step, "achieved_GBps_all_gpus": step_gbs,
                     "frac_of_peak": step_gbs / (HBM_PEAK_GBS * world)},
        "kernels": kernels,
        "jacobi_last_update": stats["last_update"], "jacobi_increment": stats["increment"],
    }
    ctx.close()
    if rank == 0 and world == 1 and on_gpu and not args.no_extras:
        sm = study_mesh_run()
        cb = cpu_baseline_run()
        sm["speedup_vs_cpu_baseline"] = sm["steps_per_s"] / cb["value"]
        out["study_mesh"] = sm
        out["cpu_baseline"] = cb
        out["other_configs"] = other_configs_run()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
