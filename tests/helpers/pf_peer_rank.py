"""One rank of a particle filter sharded over PROCESSES that share GPU 0 (tests/test_gpu_trackers.py::
test_pf_peer_exchange_between_processes): a detached communicator, the mailbox handles moved through files of a scratch directory
(the host program's own transport), the weights through the peer-store exchange.  argv: rank world scratch_dir n_particles."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mtf_amd  # noqa: E402
from mtf_amd import _lib as L, synth  # noqa: E402
from mtf_amd.sm import Comm, ParticleFilter  # noqa: E402


def main():
    rank, world, scratch, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    n += int(os.environ.get("PF_PEER_TEST_EXTRA_PARTICLES", "0"))   # (a rank created with another particle count: must be refused)

    def transport(mine):
        tmp = os.path.join(scratch, "handle_%d.tmp" % rank)
        with open(tmp, "wb") as f:
            f.write(mine)
        os.rename(tmp, os.path.join(scratch, "handle_%d.bin" % rank))   # (atomic: a reader never sees half a handle)
        out, t0 = [], time.time()
        for q in range(world):
            path = os.path.join(scratch, "handle_%d.bin" % q)
            while not os.path.exists(path):
                if time.time() - t0 > 120:
                    raise RuntimeError("rank %d never published its handle" % q)
                time.sleep(0.01)
            with open(path, "rb") as f:
                out.append(f.read())
        return out

    frame = synth.make_frame(480, 640)
    corners = synth.square_corners(250.0, 240.0, 80) + np.array([[0.3, -0.2, 0.1, 0.4], [0.2, 0.1, -0.3, 0.2]])
    frame_b = synth.warp_frame(frame, np.array([0, 0, 1.2, 0, 0, -0.8, 0, 0]), (250.0, 240.0))
    ctx = mtf_amd.Context(0)
    ctx.set_image(frame)
    comm = Comm.detached(rank, world) if world > 1 else None
    pf = ParticleFilter(ctx, L.SSM_HOMOGRAPHY, 24, 24, n_particles=n, ssm_sigma=(1.0, 0.6, 1, 1, 1, 1, 1, 1), likelihood_alpha=5.0,
                        seed=int(os.environ.get("PF_PEER_TEST_SEED", "123")), corner_based_sampling=1, resampling_type=1, max_iters=6, epsilon=-1.0,
                        comm=comm, exchange="peer" if comm is not None else "collective", exchange_transport=transport if comm is not None else None)
    pf.initialize(corners[None])
    ctx.set_image(frame_b)
    idle = os.environ.get("PF_PEER_TEST_IDLE_RANK", "")
    if idle != "":
        # one rank connects and then never iterates: the others' scans must give up (bounded spin) and say so, not hang
        if int(idle) == rank:
            time.sleep(float(os.environ.get("PF_PEER_TEST_IDLE_SECONDS", "12")))
            pf.close(); ctx.close(); comm.close()
            return
        t0 = time.time()
        try:
            pf.iteration()
        except mtf_amd.MtfHipError as e:
            print("GAVE_UP after %.1f s: %s" % (time.time() - t0, e))
            pf.close(); ctx.close(); comm.close()
            return
        raise SystemExit("the iteration came back although rank %s never stored its weights" % idle)
    rec = {}
    for it in range(3):              # the host in between ...
        pf.iteration()
        st, ar, w, ids = pf.particles()
        rec["st%d" % it], rec["w%d" % it], rec["ids%d" % it] = st.copy(), w.copy(), ids.copy()
    pf.update()                      # ... and six exchanges enqueued back to back
    st, ar, w, ids = pf.particles()
    rec["st_u"], rec["w_u"], rec["ids_u"], rec["corners"] = st.copy(), w.copy(), ids.copy(), np.asarray(pf.get_region()).copy()
    np.savez(os.path.join(scratch, "result_%d.npz" % rank), **rec)
    pf.close(); ctx.close()
    if comm is not None:
        comm.close()


if __name__ == "__main__":
    main()
