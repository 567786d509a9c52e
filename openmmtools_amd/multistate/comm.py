"""Replica sharding across ranks (one process per GPU) and the two collectives of the path.

The reference distributes replicas over MPI ranks with mpiplus (round-robin ``rank::size``,
multistatesampler.py:1296-1297, 1448-1449) and gathers pickled results on rank 0, then
broadcasts the mixed labels (replicaexchange.py:255).  Here replicas are block-partitioned,
positions never leave their GPU (state labels move, not coordinates: multistatesampler.py:
1316-1319), and the only data-path collective is an all-gather of each rank's u_kl rows
(``torch.distributed`` backend "nccl" = RCCL over xGMI on GPUs, "gloo" in CPU tests).  The
mix itself is replicated: every rank runs the same deterministic Philox-driven kernel on the
same gathered matrix, so the label "broadcast" of the reference degenerates to an optional
consistency check (``verify_labels``) that costs one tiny broadcast.
"""
import numpy as np


def block_partition(n_items, world_size):
    """Contiguous blocks; the first n_items % world_size ranks get one extra item."""
    base, rem = divmod(n_items, world_size)
    counts = [base + (1 if r < rem else 0) for r in range(world_size)]
    begins = [0]
    for c in counts[:-1]:
        begins.append(begins[-1] + c)
    return begins, counts


class SingleProcessComm:
    rank = 0
    world_size = 1

    def partition(self, n_replicas):
        return 0, n_replicas

    def all_gather_rows(self, rows, n_replicas, out=None):
        return rows

    def broadcast_labels(self, labels):
        return labels

    def gather_objects(self, obj):
        return [obj]

    def broadcast_object(self, obj):
        return obj

    def barrier(self):
        pass


class TorchDistributedComm:
    """Wraps an initialised torch.distributed process group."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)

    def partition(self, n_replicas):
        begins, counts = block_partition(n_replicas, self.world_size)
        self._begins, self._counts = begins, counts
        return begins[self.rank], counts[self.rank]

    def all_gather_rows(self, rows, n_replicas, out=None):
        """rows: torch tensor [R_local, K] (device or CPU) -> full [R, K] on every rank."""
        import torch
        K = rows.shape[1]
        if out is None:
            out = torch.empty((n_replicas, K), dtype=rows.dtype, device=rows.device)
        if rows.is_cuda and self.dist.get_backend(self.group) != 'nccl':
            # a host-only backend (gloo: several ranks sharing one GPU in the tests, or a CPU fabric): stage the
            # <= 128 KiB of rows through host memory; the gathered matrix still ends up on the device for the mix
            full = self.all_gather_rows(rows.cpu(), n_replicas)
            out.copy_(full)
            return out
        if len(set(self._counts)) == 1:
            self.dist.all_gather_into_tensor(out, rows.contiguous(), group=self.group)
        else:
            # ragged blocks: pad every rank's block to the largest one, gather, then compact
            cmax = max(self._counts)
            send = torch.zeros((cmax, K), dtype=rows.dtype, device=rows.device)
            send[:rows.shape[0]] = rows
            recv = torch.empty((self.world_size * cmax, K), dtype=rows.dtype, device=rows.device)
            self.dist.all_gather_into_tensor(recv, send, group=self.group)
            for r, (b, c) in enumerate(zip(self._begins, self._counts)):
                out[b:b + c] = recv[r * cmax:r * cmax + c]
        return out

    def broadcast_labels(self, labels):
        import torch
        backend = self.dist.get_backend(self.group)
        dev = 'cuda' if backend == 'nccl' else 'cpu'
        t = torch.as_tensor(np.asarray(labels, dtype=np.int64), device=dev)
        self.dist.broadcast(t, src=0, group=self.group)
        return t.cpu().numpy()

    def gather_objects(self, obj):
        """Host-side gather to rank 0 (checkpoint snapshots only; the reference gathers every replica's pickled
        SamplerState on every iteration, multistatesampler.py:1303-1311).  Returns the list on rank 0, None elsewhere."""
        out = [None] * self.world_size if self.rank == 0 else None
        self.dist.gather_object(obj, out, dst=0, group=self.group)
        return out

    def broadcast_object(self, obj):
        """Small host object from rank 0 to every rank (the offline analysis' error estimate, which decides completion)."""
        box = [obj]
        self.dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0]

    def barrier(self):
        self.dist.barrier(group=self.group)
