"""One GPU: what the peer-store exchange adds to the kernels it lives in.  Eight loopback ranks (threads, one GPU) run config 4's
filter at 10 000 particles with each exchange; rank 0's HIP-event times of the scoring launch (its 1 250-particle block) and of the
scan + selection are printed for both.  The eight ranks' kernels run concurrently in both cases, so the two columns compare like
with like; the absolute values are not those of a rank that has a GPU to itself (bench.py pf_strong measures that)."""
import json
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mtf_amd  # noqa: E402
from mtf_amd import synth  # noqa: E402
from mtf_amd.sm import Comm, ParticleFilter  # noqa: E402


def run(world, n, exchange, updates=20, iters=10):
    frame = synth.make_frame(1024, 1024)
    corners = synth.square_corners(512, 512, 100)
    comms = Comm.loopback(world)
    out = [None] * world

    def work(r):
        ctx = mtf_amd.Context(0)
        ctx.set_image(frame)
        pf = ParticleFilter(ctx, mtf_amd.SSM_HOMOGRAPHY, 50, 50, n_particles=n, ssm_sigma=(1.0, 0.5, 1, 1, 1, 1, 1, 1), corner_based_sampling=1,
                            resampling_type=1, max_iters=iters, epsilon=-1.0, seed=5, comm=comms[r], exchange=exchange)
        pf.initialize(corners[None])
        for _ in range(3):
            pf.update()
        ctx.timing(1); ctx.timing_reset()
        for _ in range(updates):
            pf.update()
        out[r] = {"score_us": ctx.timing_get("pf_score")[0] * 1e3, "scan_select_us": ctx.timing_get("pf_resample")[0] * 1e3,
                  "estimate": np.asarray(pf.get_region()).ravel().tolist()}
        ctx.timing(False)
        pf.close(); ctx.close()
    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for c in comms:
        c.close()
    return out


if __name__ == "__main__":
    res = {}
    for n in (10000, 100000):
        a, b = run(8, n, "collective"), run(8, n, "peer")
        res[str(n)] = {"collective_rank0": {k: a[0][k] for k in ("score_us", "scan_select_us")},
                       "peer_rank0": {k: b[0][k] for k in ("score_us", "scan_select_us")},
                       "same_estimate_on_all_ranks_and_both_exchanges": all(x["estimate"] == a[0]["estimate"] for x in a + b)}
    print(json.dumps(res, indent=1))
