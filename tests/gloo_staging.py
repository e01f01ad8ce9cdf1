"""Two ranks cannot share one GPU under RCCL, and gloo has no device-memory transport: the tests that run two
slabs on ONE GPU move the ghost rows and the small reductions of tedd.distributed through host memory over gloo.
Test infrastructure only; the product module talks RCCL on device tensors and gloo on host tensors."""
import torch
import torch.distributed as dist

from tedd.distributed import HaloExchange, SlabStepper


class HostStagedHaloExchange(HaloExchange):
    def exchange(self, fields, before_wait=None):
        s = self.slab
        if s.world == 1:
            return
        if not fields[0].is_cuda:
            return super().exchange(fields, before_wait)
        o_lo, o_hi = s.owned_local
        ops, unpack = [], []
        if s.rank > 0:
            n_send = min(s.ghost, s.n_rows - s.row_lo)
            send = torch.stack([f[o_lo:o_lo + n_send] for f in fields]).cpu()
            recv = torch.empty((len(fields), s.ghost_lo, fields[0].shape[1]), dtype=fields[0].dtype)
            ops += [dist.P2POp(dist.isend, send, self.peer_lo, self.group),
                    dist.P2POp(dist.irecv, recv, self.peer_lo, self.group)]
            unpack.append((recv, 0, s.ghost_lo))
        if s.rank < s.world - 1:
            n_send = min(s.ghost, s.row_hi)
            send = torch.stack([f[o_hi - n_send:o_hi] for f in fields]).cpu()
            recv = torch.empty((len(fields), s.ghost_hi, fields[0].shape[1]), dtype=fields[0].dtype)
            ops += [dist.P2POp(dist.isend, send, self.peer_hi, self.group),
                    dist.P2POp(dist.irecv, recv, self.peer_hi, self.group)]
            unpack.append((recv, o_hi, o_hi + s.ghost_hi))
        works = dist.batch_isend_irecv(ops)
        if before_wait is not None:
            before_wait()
        for w in works:
            w.wait()
        for recv, a, b in unpack:
            r = recv.to(fields[0].device)
            for k, f in enumerate(fields):
                f[a:b].copy_(r[k])


class HostStagedSlabStepper(SlabStepper):
    halo_cls = HostStagedHaloExchange

    def _all_reduce(self, tensor, op):
        if not tensor.is_cuda:
            return super()._all_reduce(tensor, op)
        host = tensor.cpu()
        dist.all_reduce(host, op=op, group=self.group)
        tensor.copy_(host)
