"""A notebook's whole convergence study (eight-level spatial sweep + its temporal study), the levels sharded
across the ranks of one node, checked against the numbers the notebook prints.

    python tools/sweep_sharded.py [--case nfs_h2_h3]                       one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        --master-port 29511 tools/sweep_sharded.py --case nfs_h2_h3        one level per GPU, RCCL only for the gather

Each level is an independent simulation on the rank's own GPU (no halo exchange: the study meshes are tiny
next to one GPU); rank 0 prints errors, 3-point rates and the wall time."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="nfs_h2_h3", help="a key of tests/golden/published_notebook_errors.json, or all")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--library", default=None, help="path of the C-ABI library (default: the HIP build)")
    ap.add_argument("--max-n", type=int, default=None)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "nccl":
        torch.cuda.set_device(local_rank)
    if world > 1:
        kw = {"device_id": torch.device("cuda", local_rank)} if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, **kw)
    from tedd import _ffi
    if args.library:
        _ffi.use_library(args.library)
    # the mirrored classes create their device contexts on the current HIP device, set above
    import api_cases as ac
    from tedd.cvg_studies_base import calculate_observed_rates
    from tedd.distributed import run_levels_sharded

    for case in (list(ac.PUBLISHED) if args.case == "all" else [args.case]):
        cls, mset, Tf = ac.CASES[case]
        pub = ac.PUBLISHED[case]
        model = ac.model_of(mset)
        levels = [("spatial", lv["N"], (1.0 / lv["N"]) ** 1.5, Tf, lv["overall"]) for lv in pub["levels"]]
        levels += [("temporal", lv.get("N", pub["temporal"]["N"]), lv["dt"], pub["temporal"]["Tf"], lv["overall"])
                   for lv in pub["temporal"]["levels"]]
        if args.max_n:
            levels = [lv for lv in levels if lv[1] <= args.max_n]
        costs = [max(N, 1) ** 2 * max(T / dt, 1.0) + 1e4 for _k, N, dt, T, _e in levels]

        def run_one(level):
            _kind, N, dt, T, _published = level
            return float(ac.reg_trial(cls, model, N).run_for_errors(Tf=T, dt=dt).overall_combined_error)

        t0 = time.perf_counter()
        errors = run_levels_sharded(levels, run_one, costs)
        wall = time.perf_counter() - t0
        if int(os.environ.get("RANK", "0")) == 0:
            worst = max(abs(e - lv[4]) / lv[4] for e, lv in zip(errors, levels) if lv[4] > 0)
            spatial = [e for e, lv in zip(errors, levels) if lv[0] == "spatial"]
            rates = [r for r, _s in calculate_observed_rates(spatial, 2.0)] if len(spatial) >= 3 else []
            print(json.dumps({"case": case, "ranks": world, "levels": len(levels), "wall_s": wall,
                              "worst_relative_difference_to_notebook": worst, "spatial_rates_3pt": rates,
                              "notebook_rates_3pt": pub["rates_3pt"]}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
