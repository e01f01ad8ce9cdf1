"""Two ranks cannot share one GPU under RCCL, and gloo has no device-memory transport: the tests that run two
slabs on ONE GPU move the ghost rows and the small reductions of tedd.distributed through host memory over gloo.
Test infrastructure only; the product module talks RCCL on device tensors and gloo on host tensors."""
import torch
import torch.distributed as dist

from tedd.distributed import HaloExchange, SlabStepper


class HostStagedHaloExchange(HaloExchange):
    """Messages live in host memory: packed by a device-to-host copy, unpacked by a host-to-device copy."""

    def _pack(self, key, fields, a, b):
        if not fields[0].is_cuda:
            return super()._pack(key, fields, a, b)
        return torch.stack([f[a:b] for f in fields]).cpu()

    def _recv_buffer(self, key, fields, rows):
        if not fields[0].is_cuda:
            return super()._recv_buffer(key, fields, rows)
        return torch.empty((len(fields), rows, fields[0].shape[1]), dtype=fields[0].dtype)

    def _unpack(self, recv, fields, a, b):
        if not fields[0].is_cuda:
            return super()._unpack(recv, fields, a, b)
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
