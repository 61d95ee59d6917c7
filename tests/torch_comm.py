"""TEST INFRASTRUCTURE: a raft_amd.comm-style communicator on top of torch.distributed (gloo), so that the sharded
sweep drivers are also exercised over the process-group stack the multi-GPU launcher uses.  The product itself
(raft_amd/) does not import torch: its transports are raft_amd.comm.RcclComm / HostComm."""
import numpy as np


class GlooComm:
    kind = "torch.distributed-gloo"

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def barrier(self):
        self.dist.barrier()

    def broadcast_arrays(self, arrays=None):
        box = [arrays if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def gather_rows(self, local, counts=None):
        import torch
        local = np.ascontiguousarray(local)
        n = torch.tensor([local.shape[0]], dtype=torch.int64)
        ns = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        self.dist.all_gather(ns, n)
        ns = [int(x.item()) for x in ns]
        pad = max(ns)
        is_c = np.iscomplexobj(local)
        arr = local.view(np.float64) if is_c else local
        buf = np.zeros((pad,) + arr.shape[1:], dtype=arr.dtype)
        buf[:arr.shape[0]] = arr
        t = torch.as_tensor(buf)
        parts = [torch.empty_like(t) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(t, parts, dst=0)
        if self.rank != 0:
            return None
        full = np.concatenate([p.numpy()[:c] for p, c in zip(parts, ns)], axis=0)
        return full.view(np.complex128) if is_c else full

    def gather_xi(self, ctx, counts=None, out=None):
        r = ctx.fetch_results(want_Xi=True)["Xi"]
        return self.gather_rows(r.reshape((-1,) + r.shape[2:]))

    def reduce_sum(self, arr):
        import torch
        a = np.ascontiguousarray(arr)
        is_c = np.iscomplexobj(a)
        t = torch.as_tensor(a.view(np.float64).copy() if is_c else a.astype(np.float64))
        self.dist.reduce(t, dst=0, op=self.dist.ReduceOp.SUM)
        if self.rank != 0:
            return None
        out = t.numpy()
        return out.view(np.complex128).reshape(a.shape) if is_c else out.reshape(a.shape)

    def close(self):
        self.dist.destroy_process_group()
