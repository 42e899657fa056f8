"""Candidate-axis sharding over the GPUs of one node.

The only batch axis of the hot path that spans GPUs with a real exchange step is the PF / NN candidate
set: every rank holds the frame and the template, scores a contiguous block of the candidates, and one
all-gather of the per-candidate scores (RCCL over xGMI when the backend is "nccl") gives every rank all
weights, so that resampling runs redundantly from identical data with no second collective
(SM/src/PF.cc:283-306).  Independent targets (GridTracker patches, concurrent trackers) shard with no
collective at all.
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


class ShardedScorer:
    """Scores C candidate states against one template, sharded over the ranks of a process group.

    score_fn(states_shard) -> (n_shard,) float64 array-like on `device`; the default uses the HIP
    scorer of a Batch (mtfhip_score_candidates_dev) with device-resident inputs and outputs.
    """

    def __init__(self, batch=None, group=None, device=None, score_fn=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.batch, self.group = batch, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = device if device is not None else torch.device("cpu")
        self.score_fn = score_fn
        self._buf = None

    def _score_local(self, states_shard):
        torch = self.torch
        if self.score_fn is not None:
            out = self.score_fn(states_shard)
            return torch.as_tensor(np.asarray(out, dtype=np.float64), device=self.device)
        if self.batch is None:
            raise RuntimeError("ShardedScorer needs a Batch (HIP scorer); there is no CPU fallback")
        st = torch.as_tensor(np.ascontiguousarray(states_shard, dtype=np.float64)).to(self.device)
        lik = torch.empty(st.shape[0], dtype=torch.float64, device=self.device)
        # The scorer runs on the Context's stream, which need not be torch's current stream (a default Context owns a private
        # non-blocking one): order the two explicitly.  `st` was produced on torch's stream -> the context waits for it; `lik` is
        # consumed on torch's stream (copy, all-gather) -> torch waits for the context; and the temporary `st` must outlive the
        # kernel that reads it, so it is only released after the context's stream has drained.
        cur = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        if cur is not None and cur.cuda_stream != (self.batch.ctx.stream or 0):
            cur.synchronize()
            self.batch.score_candidates_dev(st.data_ptr(), st.shape[0], lik.data_ptr())
            self.batch.ctx.synchronize()
        else:
            self.batch.score_candidates_dev(st.data_ptr(), st.shape[0], lik.data_ptr())
        return lik

    def score(self, states):
        """states: (C, S) on every rank (identical).  Returns the (C,) likelihoods on every rank."""
        torch, dist = self.torch, self.dist
        C = states.shape[0]
        lo, hi = shard_bounds(C, self.rank, self.world)
        local = self._score_local(states[lo:hi])
        if self.world == 1:
            return local
        sizes = shard_sizes(C, self.world)
        m = max(sizes)
        # equal-size all-gather (one collective, as ncclAllGather requires); ragged tails are padded
        send = torch.zeros(m, dtype=torch.float64, device=self.device)
        send[: hi - lo] = local
        if self._buf is None or self._buf.numel() != m * self.world:
            self._buf = torch.empty(m * self.world, dtype=torch.float64, device=self.device)
        dist.all_gather_into_tensor(self._buf, send, group=self.group)
        if all(s == m for s in sizes):
            return self._buf
        parts = [self._buf[r * m: r * m + sizes[r]] for r in range(self.world)]
        return torch.cat(parts)


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

