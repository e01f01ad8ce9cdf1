"""The RCCL code paths of the slab stepper on one MI355X: a process group of ONE rank over the "nccl" backend
(two ranks cannot share one device under RCCL, so the slabs of a mesh cannot be rehearsed here; the two-slab
parity runs stage through gloo).  What this covers is everything of the multi-GPU path except the peer being
another device: collectives on the library's own device memory wrapped through raw addresses, called back from
inside a step (the reduce hook between the two passes of the cs corrector), the ordering between the
context's stream and RCCL's, the grouped send/receive of the ghost-row exchange (both neighbours played by the
rank itself) and the collectives of `SlabStepper.check_converged` and of `bench.py`.

The worker runs in a child process under a time limit, so that a hung collective fails this test instead of
the whole run."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(ROOT, "na-nonlinear-temperature-enhanced-diffusion-model-dd_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tedd import _ffi
from tedd.distributed import HaloExchange, Slab, SlabStepper, partition_rows
from test_distributed_gloo import STRESS, _early_exit_run

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(PORT), HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
_ffi.use_library(_ffi.DEFAULT_LIBRARY)

# 1. the reduce hook: an all-reduce (MAX / MIN) of the corrector's per-iteration record, on the library's own
#    device memory, called from inside the step between the record pass and the deciding pass
N, M, nsteps = 47, 21, 3
kw = dict(cs_newton_iterations=12, cs_consec_xs_rtol=2e-3, jacobi_sweeps=[4, 3, 2])
x, y = np.linspace(0, 1, N + 1), np.linspace(0, 1, M + 1)
def make():
    return _ffi.Context(x, y, STRESS, model_kind=2, reaction_kind=_ffi.REACT_REGHCSTRIPLE, eta=50.0, mms_case=-1)
ctx = make()
kept_ref = _early_exit_run(ctx, None, None, N, M, nsteps, kw)
ref = ctx.get_state()
ctx.close()
assert all(1 <= k < 12 for k in kept_ref), kept_ref
ctx = make()
slab = partition_rows(N + 1, 1, 0, 0)
stepper = SlabStepper(ctx, slab, dev)
calls = [0]
def hook(pmax, pmin, n):
    calls[0] += 1
    stepper._reduce_record(pmax, pmin, n)
ctx.set_reduce_hook(hook)
kept = _early_exit_run(ctx, stepper, None, N, M, nsteps, kw)
out = ctx.get_state()
assert calls[0] == nsteps, calls
assert kept == kept_ref, (kept, kept_ref)
for a, b in zip(out, ref):
    np.testing.assert_array_equal(a, b)
stepper.check_converged()
ctx.close()
print("reduce hook over RCCL: ok", flush=True)

# 2. the collectives of check_converged and of the bench, on device tensors
flag = torch.tensor([2], dtype=torch.int32, device=dev)
dist.all_reduce(flag, op=dist.ReduceOp.MAX)
tn = torch.arange(8, dtype=torch.float64, device=dev)
dist.all_reduce(tn)
dist.barrier()
torch.cuda.synchronize()
assert int(flag.item()) == 2 and tn.cpu().tolist() == list(range(8))
print("collectives: ok", flush=True)

# 3. the ghost-row exchange: a middle slab whose two neighbours are this rank itself; the grouped sends and
#    receives pair up in the order they are posted, so the lower ghost rows receive the slab's own first owned
#    rows and the upper ghost rows its last
g = 7
slab = Slab(rank=1, world=3, n_rows=300, row_lo=100, row_hi=200, ghost=g)
pitch = 48
gen = torch.Generator(device="cpu").manual_seed(5)
host = [torch.rand((slab.local_rows, pitch), dtype=torch.float64, generator=gen) for _ in range(5)]
fields = [h.to(dev) for h in host]
halo = HaloExchange(slab)
halo.peer_lo = halo.peer_hi = 0
ran = []
for rep in range(3):   # the later exchanges reuse the message buffers
    if rep < 2:
        halo.exchange(fields, before_wait=lambda: ran.append(1))
    else:
        # posted, left in flight under other work on the compute stream, completed later (the overlap mode of
        # the slab stepper)
        posted = halo.post(fields)
        busy = torch.ones((4096, 4096), dtype=torch.float64, device=dev)
        for _ in range(5):
            busy = busy * 1.0000001
        halo.complete(posted)
    torch.cuda.synchronize()
    o_lo, o_hi = slab.owned_local
    for h, f in zip(host, fields):
        got = f.cpu()
        assert torch.equal(got[o_lo:o_hi], h[o_lo:o_hi])
        assert torch.equal(got[:g], h[o_lo:o_lo + g])
        assert torch.equal(got[o_hi:], h[o_hi - g:o_hi])
        h.copy_(got)
        h[o_lo:o_hi] += 1.0
    for h, f in zip(host, fields):
        f.copy_(h)
assert ran == [1, 1]
print("ghost-row exchange over RCCL: ok", flush=True)
dist.destroy_process_group()
'''


def test_rccl_paths_with_one_rank(tmp_path):
    import socket
    with socket.socket() as sk:   # a port nobody listens on right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {ROOT!r}\nPORT = {port}\n" + WORKER)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, "-u", str(script)], env=env, capture_output=True, text=True, timeout=100)
    except subprocess.TimeoutExpired as e:
        pytest.fail(f"an RCCL path hung; output so far:\n{e.stdout}\n{e.stderr}")
    assert r.returncode == 0, r.stdout + "\n" + r.stderr
    for part in ("reduce hook over RCCL: ok", "collectives: ok", "ghost-row exchange over RCCL: ok"):
        assert part in r.stdout, r.stdout + "\n" + r.stderr
