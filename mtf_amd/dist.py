"""Independent-target sharding over the GPUs of one node.

The only batch axis of the hot path that spans GPUs with a real exchange step is the PF candidate set, and that one lives
behind the C ABI: mtfhip_pf_set_comm + mtfhip_allgather_scores (RCCL directly; mtf_amd.sm.Comm / ParticleFilter) -- every rank
scores a contiguous block of the particles and ONE in-place all-gather leaves the flat weight vector on every rank
(SM/src/PF.cc:262-277).  Independent targets (GridTracker patches, concurrent trackers) shard with no collective at all;
this module holds that partition and the final gather of their results.
"""
import numpy as np


def shard_bounds(n_items, rank, world):
    """Contiguous block partition [lo, hi) of n_items over `world` ranks (sizes differ by at most 1)."""
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_sizes(n_items, world):
    return [shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0] for r in range(world)]


def padded_shard(n_items, rank, world):
    """the particle filter's partition (mtfhip_pf_shard_bounds): ceil(n / world) items per rank of a world x ceil(n / world) buffer, the last
    ranks' blocks ragged or empty -> (lo, count, per_rank).  Row blocks of this shape go through ONE all_gather_into_tensor."""
    m = -(-int(n_items) // int(world))
    lo = min(rank * m, int(n_items))
    return lo, max(0, min(int(n_items), (rank + 1) * m) - lo), m


class ShardedTargets:
    """Independent targets (concurrent trackers of config 5, GridTracker patches) sharded over the ranks of a process
    group: rank r owns the contiguous block shard_bounds(n_targets, r, world) and tracks it with its own Batch -- no
    collective on the hot path (SM/src/GridTracker.cc:254-261, Examples/cpp/runMTF.cc:145-177).  The only exchange is the
    gather of the final states and corners (8 + 8 doubles per target) for reporting."""

    def __init__(self, n_targets, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_targets = int(n_targets)
        self.lo, self.hi = shard_bounds(self.n_targets, self.rank, self.world)
        self.device = device if device is not None else torch.device("cpu")

    def local(self, per_target):
        """this rank's block of an array whose leading axis runs over all targets"""
        return per_target[self.lo:self.hi]

    def gather(self, local_states, local_corners):
        """local_states (n_local, S), local_corners (n_local, 2, 4) -> (states (n_targets, S), corners (n_targets, 2, 4)) on
        every rank: one padded all-gather of 16 doubles per target."""
        torch, dist = self.torch, self.dist
        st = np.asarray(local_states, dtype=np.float64).reshape(self.hi - self.lo, -1)
        S = st.shape[1]
        row = np.concatenate([st, np.asarray(local_corners, dtype=np.float64).reshape(self.hi - self.lo, 8)], axis=1)
        if self.world == 1:
            return row[:, :S].copy(), row[:, S:].reshape(-1, 2, 4).copy()
        sizes = shard_sizes(self.n_targets, self.world)
        m = max(sizes)
        send = torch.zeros((m, S + 8), dtype=torch.float64, device=self.device)
        send[: self.hi - self.lo] = torch.as_tensor(row, device=self.device)
        buf = torch.empty((m * self.world, S + 8), dtype=torch.float64, device=self.device)
        dist.all_gather_into_tensor(buf, send, group=self.group)
        full = torch.cat([buf[r * m: r * m + sizes[r]] for r in range(self.world)]).cpu().numpy()
        return full[:, :S].copy(), full[:, S:].reshape(-1, 2, 4).copy()

