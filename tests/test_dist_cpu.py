"""CPU: the N > 1 paths with world_size-2 gloo process groups -- the particle filter's partition (the C ABI's own
mtfhip_pf_shard_bounds) + its single all-gather with the scoring kernel replaced by a stand-in, and the independent-target
shards.  The device side of the sharded filter runs in tests/test_gpu_trackers.py::test_pf_sharded_loopback_equals_unsharded."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mtf_amd import dist as mdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 8, 9, 10000, 10001):
        for w in (1, 2, 3, 8):
            b = [mdist.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == mdist.shard_sizes(n, w)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_cand, q):
    """the sharded particle filter's exchange step with the device work replaced by a stand-in: the rank's block from
    mtfhip_pf_shard_bounds (the C ABI's own partition, host arithmetic), scored into its global position of a ceil(n / world) x
    world buffer, ONE in-place all-gather -- the flat weight vector must come out in particle order on every rank"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mtf_amd.sm import Comm
        rng = np.random.default_rng(0)           # identical particles on every rank (the device generator is keyed by index)
        states = rng.normal(size=(n_cand, 8))
        fn = lambda s: np.exp(-np.abs(s).sum(axis=1))   # noqa: E731  stands in for k_pf_score
        lo, cnt, m = Comm.shard_bounds(n_cand, world, rank)
        wts = torch.full((m * world,), -1.0, dtype=torch.float64)
        wts[lo:lo + cnt] = torch.from_numpy(fn(states[lo:lo + cnt]))
        dist.all_gather_into_tensor(wts, wts[rank * m:(rank + 1) * m].clone())   # (gloo has no in-place form: same layout)
        got = wts[:n_cand].numpy()
        q.put((rank, np.array_equal(got, fn(states)), got.shape[0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_cand", [64, 101, 1])
def test_sharded_scoring_allgather_world2(n_cand):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_cand, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(ok and n == n_cand for _, ok, n in res)


def test_pf_shard_bounds_partition():
    """mtfhip_pf_shard_bounds (no device needed): blocks of ceil(n / world), contiguous, covering every particle once, so that
    rank-major blocks of per_rank weights ARE the flat vector"""
    from mtf_amd.sm import Comm
    for n in (0, 1, 5, 7, 8, 9, 10000, 10001):
        for w in (1, 2, 3, 8):
            b = [Comm.shard_bounds(n, w, r) for r in range(w)]
            m = b[0][2]
            assert m == -(-n // w) and all(x[2] == m for x in b)
            assert all(x[0] == min(n, r * m) for r, x in enumerate(b))
            assert sum(x[1] for x in b) == n and all(0 <= x[1] <= m for x in b)


def _worker_targets(rank, world, port, n_targets, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(1)           # the same target list on every rank
        corners0 = rng.uniform(50, 400, size=(n_targets, 2, 4))
        track = lambda c: (c.reshape(len(c), 8) * 0.01, c + 0.5)   # noqa: E731  stands in for Batch.track on this rank's block
        sh = mdist.ShardedTargets(n_targets)
        st, cr = track(sh.local(corners0))
        states, corners = sh.gather(st, cr)
        want_s, want_c = track(corners0)
        q.put((rank, np.array_equal(states, want_s) and np.array_equal(corners, want_c), sh.hi - sh.lo))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_targets", [64, 7])
def test_sharded_targets_gather_world2(n_targets):
    """Config 5 / grid axis: each rank tracks its own block of targets, one gather of (state, corners) at the end."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_targets, args=(r, world, port, n_targets, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sum(n for _, _, n in res) == n_targets


def test_binary_multinomial_resampling_matches_oracle(oracle):
    from mtf_amd.sm import ParticleFilter
    rng = np.random.default_rng(4)
    w = rng.uniform(0, 1, 200)
    u = rng.uniform(0, 1, 200)
    ids_o, _ = oracle.pf_binary_multinomial_resample(w, u)
    assert np.array_equal(ParticleFilter.binary_multinomial_resample(w, u), ids_o)
