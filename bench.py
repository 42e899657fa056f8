#!/usr/bin/env python
"""bench.py -- LK iterations/s of the fused hot path on MI355X (driver contract).

Workload (BASELINE.json metric): ESM + SSD + Homography, 200x200 sample points per target.
One "step" = one Lucas-Kanade iteration of every target resident on the GPU: the fused
warp -> bilinear sample -> finite-difference gradient -> steepest-descent row -> J^T r / J^T J
kernel, the fixed-order partial reduction, and the 8x8 solve + compositional update + corner test,
all on the device with no host round trip (mtfhip_batch_track).  A single target does not shard
(SURVEY.md 8e), so N GPUs run N independent replicas of the per-GPU target set: weak scaling, no
data-path collective.  Inputs (frame, templates, warps) are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # before torch touches the GPU; see mtf_amd/__init__.py

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PF_STRONG_WATCHDOG_S = 300


class LineGuardian:
    """rank 0, around the sharded-filter record (the one part of this file that has never met more than one real GPU): a child process that holds the
    headline line and prints it -- with pf_strong = {"error": ...} -- if THIS process dies before it reports the line itself (a fault inside RCCL or
    a peer mapping is a signal in C code: no Python handler runs).  Either way exactly one line leaves rank 0."""
    SRC = ("import sys\nline = sys.stdin.readline()\nrest = sys.stdin.readline()\n"
           "if line and not rest.startswith('__released__'):\n    sys.stdout.write(line if line.endswith('\\n') else line + '\\n'); sys.stdout.flush()\n")

    def __init__(self, out):
        import subprocess
        rec = dict(out)
        rec["pf_strong"] = {"error": "the process died inside the sharded-filter record (signal in native code); headline unaffected"}
        self.p = None
        try:
            self.p = subprocess.Popen([sys.executable, "-c", self.SRC], stdin=subprocess.PIPE)
            self.p.stdin.write((json.dumps(rec) + "\n").encode()); self.p.stdin.flush()
        except Exception:   # noqa: BLE001  (no guardian: the record runs unguarded, as before)
            self.p = None

    def release(self):
        if self.p is None:
            return
        try:
            self.p.stdin.write(b"__released__\n"); self.p.stdin.flush(); self.p.stdin.close()
            self.p.wait(timeout=10)
        except Exception:   # noqa: BLE001
            pass
        self.p = None


def algorithmic_bytes_per_pixel(sm, materialize, unit_z=True, j0_recompute=True):
    """Interface-level traffic per sample point of one LK iteration:
    image texels 4 (1:1 sampling) + I0 8 + init_pts 16 (+8 init_z for non-parallelogram corners)
    [+ J0 for ESM / ICLK: 64 when its rows are read back, 16 (dI0_dx) when the kernel rebuilds them -- only bytes that
    are actually moved are credited] [+ It 8 + dIt_dx 16 + Jt 64 written when materialised]."""
    b = 4 + 8 + 16 + (0 if unit_z else 8)
    if sm in ("esm", "iclk"):
        b += 16 if j0_recompute else 64
    if materialize:
        b += 8 + (0 if sm == "iclk" else 16 + 64)
    return b


def kernel_sources_sha():
    """fingerprint of the HIP sources the library is built from (mtf_amd/csrc/*.hip, *.h): a PMC traffic figure is only quoted
    next to a bench line when it was collected on exactly these sources"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mtf_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(sm, mode, res, targets, per_launch=None):
    """HBM bytes per launch of the fused kernel from the PMC passes of tools/profile_round.sh (profiles/pmc_latest.json: separate
    --pmc runs, KiB units, gfx950 FETCH_SIZE x2 correction).  Only reported for the workload it was collected on AND only while
    the kernel sources are the ones it was collected on (kernel_sources_sha): a stale figure is withheld, not printed."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    pmc_traffic.note = None
    if not (sm == "esm" and mode == "full" and res == 200 and targets == 64 and os.path.exists(path)):
        return None
    try:
        d = json.load(open(path))
        if d.get("j0_recompute", False) != (os.environ.get("MTFHIP_J0_RECOMPUTE", "1") != "0"):
            return None      # the profile was taken with the other J0 source
        if per_launch is not None and float(d.get("targets_per_launch", targets)) != float(per_launch):
            pmc_traffic.note = "withheld: profiles/pmc_latest.json counts launches of %s targets, this run launches %s" % (d.get("targets_per_launch", targets), per_launch)
            return None
        pmc_traffic.commit = d.get("commit")
        if d.get("kernel_sources_sha") != kernel_sources_sha():
            pmc_traffic.note = "withheld: profiles/pmc_latest.json was collected on other kernel sources (%s, now %s); re-run tools/profile_round.sh" % (
                d.get("kernel_sources_sha"), kernel_sources_sha())
            return None
        return float(d["traffic_bytes_per_launch"])
    except Exception:
        return None


def pmc_secondary(workload, kernel):
    """per-launch PMC averages of a secondary workload's kernel (profiles/pmc_secondary_latest.json, tools/pmc_secondary_json.py):
    FETCH_SIZE / WRITE_SIZE -> traffic_bytes_per_launch, SQ_INSTS_VALU, LDS counters.  Quoted only when the file was collected on the
    kernel sources being run (kernel_sources_sha); otherwise withheld with the reason."""
    path = os.path.join(ROOT, "profiles", "pmc_secondary_latest.json")
    pmc_secondary.note = None
    try:
        d = json.load(open(path))
    except Exception:
        pmc_secondary.note = "no profiles/pmc_secondary_latest.json"
        return None
    if d.get("kernel_sources_sha") != kernel_sources_sha():
        pmc_secondary.note = "withheld: profiles/pmc_secondary_latest.json was collected on other kernel sources (%s, now %s)" % (d.get("kernel_sources_sha"), kernel_sources_sha())
        return None
    rec = (d.get(workload) or {}).get(kernel)
    if rec is None:
        pmc_secondary.note = "profiles/pmc_secondary_latest.json holds no %s / %s" % (workload, kernel)
    return rec


# FP64 issue ceiling of a SIMD at two waves per SIMD: cycles per v_fma_f64 at the nominal 2.4 GHz (tools/fp64_rate_test.hip,
# profiles/r04_fp64_rates.txt: 6.9 at one wave, 5.9 - 6.0 at two, 5.3 at four)
FP64_ISSUE_CYCLES = {1: 6.9, 2: 5.92, 4: 5.28}
N_SIMDS, NOMINAL_HZ = 1024, 2.4e9


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds, res, frame0, frame1, corners, am_name="ssd"):
    """The CPU oracle (a port of the reference's ESM loop) timed on one host core on the same
    workload shape: LK iterations/s of a single 200x200 ESM+SSD+Homography target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    am_kind = {"ssd": O.AM_SSD, "ncc": O.AM_NCC}[am_name]
    ssm = O.SSM(O.SSM_HOM, res, res)
    am = O.AM(am_kind, res, res)
    am.set_curr_img(frame0)
    trk = O.Tracker(O.SM_ESM, am, ssm, leven_marq=0, max_iters=10, epsilon=-1.0)
    trk.initialize(corners)
    am.set_curr_img(frame1)
    trk.update()  # warm
    iters, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        ssm.set_corners(corners)   # back to the initial region; template, J0 and H0 stay
        iters += trk.update()
    dt = time.perf_counter() - t0
    out = {"value": iters / dt, "unit": "iters/s", "cores": 1, "kind": "port",
           "sample": "%d ESM+%s iterations of one %dx%d target in %.1f s (oracle/mtf_oracle.cpp, -O3, 1 thread)" % (iters, am_name.upper(), res, res, dt),
           # what makes two CPU figures of "the same loop" differ (r03: 609 it/s here, 987 it/s on the drop-in line): the LM flag -- a rejected
           # LM step skips the gradient / Hessian work of that pass (NT/ESM.cc:186-232) -- and the host the box happens to have
           "leven_marq": 0, "max_iters_per_update": 10, "host_cpu": host_cpu_model()}
    # all host cores: one independent target per thread (the reference's OpenMP-over-targets pattern, PF.cc:195-197,
    # GridTracker.cc:254-256; its default build is single-threaded, so the 1-core figure above stays the like-for-like one).
    # ctypes releases the GIL for the duration of every oracle call.
    import threading
    n_thr = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    try:    # a container may see every hardware thread but own only a CPU quota (cgroup v2 cpu.max: "<quota> <period>")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n_thr = max(1, min(n_thr, int(-(-int(q) // int(per)))))
    except (OSError, ValueError):
        pass
    n_thr = min(n_thr, 64)
    budget = min(6.0, seconds / 2.0)
    counts = [0] * n_thr

    def worker(k):
        s_k = O.SSM(O.SSM_HOM, res, res)
        a_k = O.AM(am_kind, res, res)
        a_k.set_curr_img(frame0)
        t_k = O.Tracker(O.SM_ESM, a_k, s_k, leven_marq=0, max_iters=10, epsilon=-1.0)
        t_k.initialize(corners)
        a_k.set_curr_img(frame1)
        barrier.wait()
        t_start = time.perf_counter()
        while time.perf_counter() - t_start < budget:
            s_k.set_corners(corners)
            counts[k] += t_k.update()

    if n_thr > 1 and budget > 0.5:
        barrier = threading.Barrier(n_thr + 1)
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_thr)]
        for th in threads:
            th.start()
        barrier.wait()
        t1, c1 = time.perf_counter(), time.process_time()
        for th in threads:
            th.join()
        dt_all, cpu_all = time.perf_counter() - t1, time.process_time() - c1
        out["all_cores"] = {"value": sum(counts) / dt_all, "unit": "iters/s", "cores": n_thr,
                            "sample": "%d threads x one %dx%d target each, %d iterations in %.1f s; %.1f CPU-seconds per second actually obtained"
                                      % (n_thr, res, res, sum(counts), dt_all, cpu_all / dt_all)}
    return out


def parity_gate(ctx, am_name, res, frame0, frame1, corners):
    """SURVEY 8(d) "parity gate in the same run": the oracle's nt::ESM trace of one target of this workload against the
    fused path (mtfhip_batch_iterate + the oracle's pivoted QR for the step, following the oracle's trajectory), first
    five iterations and the last one.  Returns the worst relative errors; the budget is 1e-5 (north_star)."""
    import mtf_amd
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    am_o = {"ssd": O.AM_SSD, "ncc": O.AM_NCC}[am_name]
    am_d = {"ssd": mtf_amd.AM_SSD, "ncc": mtf_amd.AM_NCC}[am_name]
    ssm = O.SSM(O.SSM_HOM, res, res); am = O.AM(am_o, res, res); am.set_curr_img(frame0)
    trk = O.Tracker(O.SM_ESM, am, ssm, leven_marq=0, max_iters=12, epsilon=1e-6)
    trk.initialize(corners); am.set_curr_img(frame1); trk.update()
    trace = trk.trace()
    ctx.set_image(frame0)
    b = mtf_amd.Batch(ctx, am_d, mtf_amd.SSM_HOMOGRAPHY, res, res, 1)
    b.set_corners(corners[None])
    sm = mtf_amd.sm_desc(mtf_amd.SM_ESM, materialize=1, leven_marq=0, max_iters=12, epsilon=1e-6)
    b.init_template(sm)
    ctx.set_image(frame1)
    rel = lambda a, r: float(np.linalg.norm(np.asarray(a) - np.asarray(r)) / max(np.linalg.norm(r), 1e-300))
    worst = {"H": 0.0, "g": 0.0, "dp": 0.0}
    # the tolerance-mode arithmetic (what the `lean` sub-record and every device-side loop run: FMA, one reciprocal per point,
    # closed-form slope of the bilinear cell x the rounded finite-difference step) at the same states: against the reference-parameter
    # oracle (the gate), and -- recorded -- against the same restatement with grad_eps = 1e-6 / 1e-7
    sm_f = mtf_amd.sm_desc(mtf_amd.SM_ESM, materialize=0, leven_marq=0, max_iters=12, epsilon=1e-6)
    low = []
    for eps in (1e-6, 1e-7):
        s_l = O.SSM(O.SSM_HOM, res, res); a_l = O.AM(am_o, res, res, grad_eps=eps); a_l.set_curr_img(frame0)
        t_l = O.Tracker(O.SM_ESM, a_l, s_l, leven_marq=0, max_iters=1)
        t_l.initialize(corners); a_l.set_curr_img(frame1)
        low.append((t_l, s_l, a_l))
    fast_low = {"H": 0.0, "g": 0.0, "dp": 0.0}; fast_ref = {"H": 0.0, "g": 0.0, "dp": 0.0}
    checked = []
    for it, rec in enumerate(trace):
        if it < 5 or it == len(trace) - 1:
            g_scale = np.sqrt(abs(np.trace(rec["H"]))) * (1.0 if am_name == "ncc" else np.sqrt(abs(2 * rec["f"])))
            errs = lambda H, g, dp, r: {"H": rel(H, r["H"]), "g": float(np.linalg.norm(g - r["g"]) / max(np.linalg.norm(r["g"]), g_scale)),
                                        "dp": min(rel(dp, r["dp"]), float(np.abs(dp - r["dp"]).max() / 1e-7))}
            b.set_math_mode(mtf_amd.MATH_FAST)
            _, gf, Hf = b.iterate(sm_f)
            b.set_math_mode(mtf_amd.MATH_REPLAY)
            dpf = -O.colpiv_qr_solve(Hf[0], gf[0])
            st = b.get_state()[0]
            el = None
            for t_l, s_l, _ in low:
                s_l.set_state(st); t_l.update()
                e = errs(Hf[0], gf[0], dpf, t_l.trace()[0])
                el = e if el is None else {q: min(el[q], e[q]) for q in e}
            e8 = errs(Hf[0], gf[0], dpf, rec)
            for q in fast_low:
                fast_low[q] = max(fast_low[q], el[q]); fast_ref[q] = max(fast_ref[q], e8[q])
            f, g, H = b.iterate(sm)
            dp = -O.colpiv_qr_solve(H[0], g[0])
            e = errs(H[0], g[0], dp, rec)
            for q in worst:
                worst[q] = max(worst[q], e[q])
            checked.append(it)
        b.compositional_update(rec["dp"][None])
    b.close()
    fast_ok = bool(max(fast_ref.values()) <= 1e-5)   # r04: judged against the reference-parameter oracle, like the replay mode
    worst.update({"iterations_checked": checked, "budget": 1e-05, "pass": bool(max(worst["H"], worst["g"], worst["dp"]) <= 1e-5) and fast_ok,
                  "note": "replay arithmetic (materialising launch) vs the CPU oracle's nt::ESM trace on the device's own sample grid; g relative to "
                          "its Cauchy-Schwarz scale, dp relative or below 1e-12 absolute",
                  "fast": {"vs_low_noise_oracle": fast_low, "vs_reference_parameters_grad_eps_1e-8": fast_ref, "pass": fast_ok,
                           "note": "the lean tolerance-mode launch at the same states against the reference-parameter oracle (grad_eps 1e-8): since r04 "
                                   "its closed-form gradient carries the ROUNDED step the reference's central difference takes (fd_step, "
                                   "mtfhip_device.h), so it meets the 1e-5 budget against the reference's own parameters; the low-noise oracle "
                                   "(grad_eps 1e-6 / 1e-7, better of the two per quantity) is now the farther one, by that quantisation"}})
    return worst


def loop_parity(ctx, sm_kind, am, ssm, res, frame0, frame1, corners, max_iters, hess_type=None, am_kw=None):
    """One target of a secondary workload through the device-side loop in tolerance mode with the per-pass trace on
    (mtfhip_batch_track_trace), against the CPU trackers with grad_eps 1e-8 (the reference's) and 1e-6 (low finite-difference
    noise): H, g, dp of the first pass (identical state) and every later update relative to the first one, final corners."""
    import mtf_amd
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    kw = dict(leven_marq=0, max_iters=max_iters, epsilon=-1.0)
    if hess_type is not None:
        kw["hess_type"] = hess_type
    rel = lambda a, r: float(np.linalg.norm(np.asarray(a) - np.asarray(r)) / max(np.linalg.norm(r), 1e-300))
    ctx.set_image(frame0)
    am_kw = am_kw or {}
    o_kw = {{"mi_pou": "pou", "mi_n_bins": "n_bins"}.get(k, k): v for k, v in am_kw.items()}
    b = mtf_amd.Batch(ctx, am, ssm, res, res, 1, **am_kw)
    b.set_math_mode(mtf_amd.MATH_FAST)
    b.set_corners(corners[None])
    sm = mtf_amd.sm_desc(sm_kind, materialize=0, **kw)
    b.init_template(sm)
    ctx.set_image(frame1)
    b.track_trace(max_iters)
    n_it, final = b.track(sm)
    recs = b.read_track_trace(n_it)[0]
    out = {}
    otr = {}
    for eps, name in ((1e-8, "vs_reference_parameters_grad_eps_1e-8"), (1e-6, "vs_low_noise_oracle_grad_eps_1e-6")):
        o_ssm = O.SSM(ssm, res, res); o_am = O.AM(am, res, res, grad_eps=eps, **o_kw); o_am.set_curr_img(frame0)
        trk = O.Tracker(sm_kind, o_am, o_ssm, **kw)
        trk.initialize(corners); o_am.set_curr_img(frame1); trk.update()
        tr = trk.trace()
        otr[eps] = tr
        r0, d0 = tr[0], recs[0]
        gs = np.sqrt(abs(np.trace(r0["H"]))) * (np.sqrt(abs(2 * r0["f"])) if am == mtf_amd.AM_SSD else 1.0)
        e = {"g": float(np.linalg.norm(d0["g"] - r0["g"]) / max(np.linalg.norm(r0["g"]), gs)), "dp": rel(d0["dp"], r0["dp"])}
        if d0["has_H"]:
            e["H"] = rel(d0["H"], r0["H"])
        scale = np.linalg.norm(r0["dp"])
        e["later_updates_over_first"] = max(float(np.linalg.norm(recs[k]["dp"] - tr[k]["dp"]) / scale) for k in range(min(len(tr), len(recs))))
        e["final_corners_px"] = float(np.abs(final[0] - trk.get_region()).max())
        out[name] = e
    b.track_trace(0)
    b.close()
    # how far the two oracles' own trajectories are from each other: the reference's sensitivity to its finite-difference step
    a, c = otr[1e-8], otr[1e-6]
    spread = max(float(np.linalg.norm(a[k]["dp"] - c[k]["dp"]) / np.linalg.norm(a[0]["dp"])) for k in range(min(len(a), len(c))))
    low = out["vs_reference_parameters_grad_eps_1e-8"]   # r04: the gate is the reference-parameter oracle (the tolerance mode carries its step quantisation)
    first_ok = max(low[q] for q in ("H", "g", "dp") if q in low) <= 1e-5
    out.update({"iterations": int(n_it[0]), "budget": 1e-5, "oracle_1e-8_vs_1e-6_later_updates_over_first": spread,
                "pass": bool(first_ok and low["later_updates_over_first"] <= max(1e-5, 4 * spread)),
                "note": "device-side loop of one target, tolerance-mode arithmetic, per-pass trace vs the CPU trackers: H, g, dp of the first pass "
                        "(identical state) within the budget; later updates within the budget or four times the two oracles' own disagreement"})
    return out


def pf_parity(ctx, frame0, corners, n=32):
    """candidate weights of the tolerance-mode (fast) scorer against the CPU oracle on the bench's own template: n particles
    proposed from shared draws, scored on the device and by the oracle's nt::PF iteration"""
    import mtf_amd
    from mtf_amd.sm import ParticleFilter
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    rng = np.random.default_rng(5)
    sigma = (1.0, 0.5, 1, 1, 1, 1, 1, 1)
    ssm = O.SSM(O.SSM_HOM, 50, 50); am = O.AM(O.AM_SSD, 50, 50); am.set_curr_img(frame0)
    ssm.set_corners(corners); am.initialize_pix_vals(ssm.get("curr_pts")); am.initialize_similarity()
    pp = O.pf_params(n, corner_based_sampling=1, sigma=sigma, resampling_type=0)
    normals, uniforms = rng.normal(size=(n, 10)), rng.uniform(size=n)
    _, _, w_o, _, _ = O.pf_iteration(am, ssm, pp, np.zeros((n, 8)), np.zeros((n, 8)), normals, uniforms, 0.0)
    out = {}
    for name, mode in (("fast", mtf_amd.MATH_FAST), ("replay", mtf_amd.MATH_REPLAY)):
        pf = ParticleFilter(ctx, mtf_amd.SSM_HOMOGRAPHY, 50, 50, n_particles=n, ssm_sigma=sigma, corner_based_sampling=1, resampling_type=0)
        pf.batch.set_math_mode(mode)
        pf.initialize(corners[None])
        pf.iteration(normals, uniforms)
        w_d = pf.particles()[2]
        out[name] = float(np.max(np.abs(w_d - w_o) / np.abs(w_o)))
        pf.close()
    out.update({"candidates": n, "budget": 1e-9, "pass": bool(max(out["fast"], out["replay"]) <= 1e-9),
                "note": "max relative error of the particle weights vs the CPU oracle's nt::PF iteration on shared draws"})
    return out


def self_spawn_if_needed(args, argv):
    """`python bench.py --gpus N` from a clean shell (no RANK / WORLD_SIZE): re-execute under torch.distributed.run with one rank per
    GPU on 127.0.0.1 and relay its exit code -- the command the driver types works as it is.  (The driver's own form,
    `python -m torch.distributed.run ... bench.py --gpus N`, arrives here with WORLD_SIZE set and goes straight on.)"""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = os.environ.copy()
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    raise SystemExit(subprocess.call(cmd, env=env))


CONFIGS_CHILD_TIMEOUT_S = 90


def _compact(line):
    """the fields of a secondary workload's line the headline's `configs` block keeps"""
    roof = line.get("roofline") or {}
    cfg = line.get("config") or {}
    akm = roof.get("avg_kernel_ms")
    if isinstance(akm, dict):
        kernel_us = {k: (v * 1e3 if v is not None else None) for k, v in akm.items()}
    else:
        kernel_us = akm * 1e3 if akm else None
    par = line.get("parity")
    if isinstance(par, dict) and "pass" in par:
        par = {"pass": par["pass"], "budget": par.get("budget"),
               "first_pass_g_dp": [(par.get("vs_reference_parameters_grad_eps_1e-8") or {}).get("g"), (par.get("vs_reference_parameters_grad_eps_1e-8") or {}).get("dp")]}
    cb = line.get("cpu_baseline") or None
    rec = {"metric": line.get("metric"), "value": line.get("value"), "unit": line.get("unit"), "ms_per_step": line.get("ms_per_step"), "steps": line.get("steps"),
           "workload": cfg.get("workload"), "dominant_kernel": roof.get("kernel"), "dominant_kernel_us_by_events": kernel_us,
           "roofline": {"bound": roof.get("bound"), "frac": roof.get("frac"), "achieved": roof.get("achieved"), "peak": roof.get("peak"), "unit": roof.get("unit")},
           "cpu_baseline": ({"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"), "sample": cb.get("sample")} if cb else None),
           "parity": par}
    for k in ("us_per_iteration", "iterations_per_step", "frame_us", "kernel_us", "score_kernel_ms", "resample_kernels_ms"):
        if k in cfg:
            rec[k] = cfg[k]
    vl = (cfg.get("cpp_driver") or {}).get("video_loop")
    if vl:
        rec["video_loop_update_us"] = {k: v.get("update_us") for k, v in vl.items() if isinstance(v, dict)}
        rec["video_loop_set_image_us"] = {k: v.get("set_image_us") for k, v in vl.items() if isinstance(v, dict)}
    if "hbm_frac" in roof:
        rec["roofline"]["hbm_frac"] = roof["hbm_frac"]
    return rec


def configs_block(args):
    """r05 verdict item 2: configs 3 / 4 / 5 of BASELINE.json in the driver's own line.  Each is this file's secondary workload run as a
    child process at a reduced step count (its own context on the same GPU; the parent's device work is over) -- a failure or a hang
    there (timeout: the child is killed) becomes {"error": ...} and cannot take the headline down."""
    import subprocess
    me = os.path.abspath(__file__)
    cpu = "%.1f" % min(args.cpu_seconds, 1.5)
    specs = [
        ("config3_grid_256_iclk_ncc_affine", ["--workload", "grid", "--steps", "100", "--warmup", "10", "--cpu-seconds", cpu]),
        ("config4_pf_10k_chained", ["--workload", "pf", "--particles", "10000", "--pf-iters", "10", "--steps", "40", "--warmup", "5", "--cpu-seconds", cpu]),
        ("config4_pf_10k_host_stepped", ["--workload", "pf", "--particles", "10000", "--pf-iters", "1", "--steps", "200", "--warmup", "20", "--no-cpu"]),
        ("config5_mi_64x400x400", ["--workload", "mi", "--steps", "6", "--warmup", "2", "--cpu-seconds", cpu]),
        ("config5_mi_64x400x400_shipped_10_bins_pou", ["--workload", "mi", "--mi-bins", "10", "--mi-pou", "1", "--steps", "6", "--warmup", "2", "--cpu-seconds", cpu]),
        ("nn_dataset_10k_50x50", ["--workload", "nn", "--steps", "10", "--warmup", "2", "--cpu-seconds", cpu]),
    ]
    env = os.environ.copy()
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = {}
    for name, argv in specs:
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, me] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=CONFIGS_CHILD_TIMEOUT_S)
            line = None
            for ln in r.stdout.decode("utf-8", "replace").splitlines():
                if ln.startswith("{"):
                    line = ln
            if r.returncode != 0 or line is None:
                out[name] = {"error": "rc %d: %s" % (r.returncode, r.stderr.decode("utf-8", "replace")[-300:])}
            else:
                out[name] = _compact(json.loads(line))
        except subprocess.TimeoutExpired:
            out[name] = {"error": "no line within %d s (child killed)" % CONFIGS_CHILD_TIMEOUT_S}
        except Exception as e:   # noqa: BLE001
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        out[name]["wall_s"] = time.perf_counter() - t0
    out["note"] = ("configs 3 / 4 / 5 of BASELINE.json (+ the NN dataset axis) at reduced step counts, each `python bench.py --workload ...` as a child process on the "
                   "same GPU after the headline's timed regions; full lines: profiles/r06_secondary_bench_lines.jsonl")
    return out



def stub_mode():
    """MTFHIP_BENCH_STUB=1: the launch / rendezvous / barrier / max-over-ranks / JSON skeleton of this file with the device work replaced
    by stand-ins and gloo instead of RCCL, so that tests/test_bench_cpu.py can run `bench.py --gpus 2` on a box without GPUs.  A stub
    line says so ("data": "stub") and is never a measurement."""
    return os.environ.get("MTFHIP_BENCH_STUB") == "1"


# the expected 1 -> 8 curve of the sharded filter, from single-GPU terms (us per iteration)
PF_STRONG_MODEL = {   # (chained update() form; one-GPU terms of profiles/r06_pf_strong_one_rank.json by HIP events, `score` = one rank's block launched alone (r04 measurement), the all-gather estimated)
    10000: {"T1_us": 49, "T8_terms_us": {"score": 12.5, "allgather": 25, "scan_select": 13.6}, "T8_us": 51, "speedup": 0.96,
            "T8_us_peer_stores": 30, "speedup_peer_stores": 1.6,
            "north_star_6x": "not reachable: T1 / 6 = 8.2 us is less than one rank's scoring launch (12.5 us) and less than the replicated selection pass (13.6 us)",
            "selection_sharded_too": "slower: the selection pass is latency-sized (11-13 us whatever the block) and every rank would need the 64-byte states of the "
                                     "ancestors the others selected: >= 25 us for 640 kB on top"},
    100000: {"T1_us": 335, "T8_terms_us": {"score": 48, "allgather": 35, "scan_select": 27.6}, "T8_us": 111, "speedup": 3.0,
             "selection_sharded_too": "27.6 -> ~13 us of selection, + ~45 us for 6.4 MB of states: slower"},
    1000000: {"T1_us": 3250, "T8_terms_us": {"score": 412, "allgather": 75, "scan_select": 197}, "T8_us": 684, "speedup": 4.7,
              "selection_sharded_too": "197 -> ~25 us of selection, + ~190 us for 64 MB of states at ~300 GB/s per GPU: 702 us, no gain -- the replicated term "
                                       "(6.5 % of the scorer: the 5.3x asymptote) costs what exchanging the particle set costs, at every size (r05 verdict item 5: stated, not built)"},
}


PF_PEER_CHILD_TIMEOUT_S = 150
PF_FILTER_KW = dict(ssm_sigma=(1.0, 0.5, 1, 1, 1, 1, 1, 1), corner_based_sampling=1, dynamic_model=0, update_type=1, likelihood_func=0,
                    resampling_type=1, mean_type=0, likelihood_alpha=1.0, epsilon=-1.0)


def pf_file_transport(scratch, rank, world, timeout_s=60.0):
    """the host program's own transport for the 64-byte mailbox handles of the peer-store exchange: one file per rank in a directory
    every rank can see (mtf_amd.sm.ParticleFilter(exchange_transport=...))"""
    def transport(mine):
        tmp = os.path.join(scratch, "handle_%d.tmp" % rank)
        with open(tmp, "wb") as f:
            f.write(mine)
        os.rename(tmp, os.path.join(scratch, "handle_%d.bin" % rank))
        out, t0 = [], time.time()
        for q in range(world):
            path = os.path.join(scratch, "handle_%d.bin" % q)
            while not os.path.exists(path):
                if time.time() - t0 > timeout_s:
                    raise RuntimeError("rank %d never published its mailbox handle" % q)
                time.sleep(0.005)
            with open(path, "rb") as f:
                out.append(f.read())
        return out
    return transport


def pf_peer_child(argv):
    """`bench.py --pf-peer-child rank world device scratch particles steps iters`: ONE rank of the sharded filter with the peer-store
    exchange, in a process of its own -- a detached communicator (no RCCL, no torch.distributed), the mailbox handles through files.
    PfDeviceEngine.peer_row starts one per rank and reads result_<rank>.json: the exchange has never met a second GPU, and whatever it
    does there (an unmappable peer, a fault on a remote store) ends this process, not the one that owns the headline."""
    rank, world, device, scratch, n, steps, iters = int(argv[0]), int(argv[1]), int(argv[2]), argv[3], int(argv[4]), int(argv[5]), int(argv[6])
    import mtf_amd
    from mtf_amd import synth
    from mtf_amd.sm import Comm, ParticleFilter
    ctx = mtf_amd.Context(device)
    ctx.set_image(synth.make_frame(1024, 1024))
    comm = Comm.detached(rank, world, device)
    pf = ParticleFilter(ctx, mtf_amd.SSM_HOMOGRAPHY, 50, 50, n_particles=n, max_iters=iters, seed=synth.DEFAULT_SEED, comm=comm, exchange="peer",
                        exchange_transport=pf_file_transport(scratch, rank, world), **PF_FILTER_KW)
    pf.initialize(synth.square_corners(512, 512, 100)[None])
    for _ in range(3):
        pf.update()
    t0 = time.perf_counter()
    for _ in range(steps):
        pf.update()             # (returns with the estimate on the host: the stream has drained)
    dt = time.perf_counter() - t0
    ctx.timing(True); ctx.timing_reset()
    for _ in range(3):
        pf.update()
    res = {"seconds": dt, "score_kernel_ms": ctx.timing_get("pf_score")[0], "scan_select_ms": ctx.timing_get("pf_resample")[0],
           "allgather_ms": ctx.timing_get("pf_allgather")[0], "peer_exchanges_timed": ctx.timing_get("pf_peer_exchange")[1],
           "estimate": [float(v) for v in np.asarray(pf.get_region()).ravel()]}
    ctx.timing(False)
    tmp = os.path.join(scratch, "result_%d.tmp" % rank)
    with open(tmp, "w") as f:
        json.dump(res, f)
    os.rename(tmp, os.path.join(scratch, "result_%d.json" % rank))
    pf.close(); ctx.close(); comm.close()


class PfDeviceEngine:
    """the sharded particle filter on the GPUs: mtf_amd.sm.ParticleFilter over the C-ABI, RCCL communicator bootstrapped through
    torch.distributed (Comm.torch_bootstrap broadcasts the 128-byte unique id)"""

    def __init__(self, ctx, dev, dist, world, local_rank):
        import torch
        import mtf_amd
        from mtf_amd import synth
        from mtf_amd.sm import Comm
        self.torch, self.mtf, self.synth, self.ctx, self.dev, self.dist = torch, mtf_amd, synth, ctx, dev, dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world, self.local_rank = world, local_rank
        self.corners = synth.square_corners(512, 512, 100)
        ctx.set_image(synth.make_frame(1024, 1024))
        self.comm = Comm.torch_bootstrap(local_rank) if world > 1 else None

    def n_ranks(self):
        return int(self.comm.world_as_seen()) if self.comm is not None else 1

    # the forms of the sharded filter: the RCCL all-gather in this process; the peer-store exchange in a child process per rank (peer_row)
    exchanges = ("collective", "peer")
    peer_in_child_process = True

    def make_filter(self, n, sharded, iters, exchange="collective"):
        from mtf_amd.sm import ParticleFilter
        pf = ParticleFilter(self.ctx, self.mtf.SSM_HOMOGRAPHY, 50, 50, n_particles=n, max_iters=iters, seed=self.synth.DEFAULT_SEED,
                            comm=self.comm if sharded else None, exchange=exchange if sharded else "collective", **PF_FILTER_KW)
        pf.initialize(self.corners[None])
        return pf

    def peer_row(self, n, steps, iters, scratch):
        """this rank's share of the sharded_peer form: a child process (pf_peer_child) on this rank's GPU; -> its result dict, or
        {"error": ...}.  The parent's own context stays idle meanwhile."""
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--pf-peer-child", str(self.rank), str(self.world), str(self.local_rank), scratch,
               str(n), str(steps), str(iters)]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                                "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        try:
            p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=PF_PEER_CHILD_TIMEOUT_S)
        except subprocess.TimeoutExpired:
            return {"error": "the rank's process did not finish within %d s" % PF_PEER_CHILD_TIMEOUT_S}
        path = os.path.join(scratch, "result_%d.json" % self.rank)
        if p.returncode != 0 or not os.path.exists(path):
            return {"error": "exit code %d: %s" % (p.returncode, (p.stdout or "").strip()[-400:])}
        with open(path) as f:
            return json.load(f)

    def estimate(self, pf):
        return [float(v) for v in np.asarray(pf.get_region()).ravel()]

    def sync(self):
        self.torch.cuda.synchronize(self.dev)

    def reduce_max(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def kernel_times(self, pf):
        self.ctx.timing(True); self.ctx.timing_reset()
        for _ in range(3):
            pf.update()
        self.sync()
        self.ctx.timing(False)
        return {"score_kernel_ms": self.ctx.timing_get("pf_score")[0], "scan_select_ms": self.ctx.timing_get("pf_resample")[0],
                "allgather_ms": self.ctx.timing_get("pf_allgather")[0]}

    def close(self):
        if self.comm is not None:
            self.comm.close()


class PfStubEngine:
    """MTFHIP_BENCH_STUB=1 (tests/test_bench_cpu.py): the same record with the device work replaced by NumPy stand-ins and the collective
    by gloo -- every rank scores ITS block of identical particles into its global position and one all-gather leaves the flat weight
    vector everywhere (the layout of mtfhip_allgather_scores)"""

    class _Filter:
        def __init__(self, eng, n, sharded, iters):
            self.e, self.n, self.sharded, self.iters = eng, n, sharded, iters
            self.states = np.random.default_rng(0).normal(size=(n, 8))   # identical on every rank
            self.checksum = 0.0

        def update(self):
            import torch
            e = self.e
            for _ in range(self.iters):
                score = lambda s: np.exp(-np.abs(s).sum(axis=1))   # noqa: E731
                if self.sharded and e.world > 1:
                    m = -(-self.n // e.world)
                    lo, hi = min(self.n, e.rank * m), min(self.n, (e.rank + 1) * m)
                    wts = torch.full((m * e.world,), -1.0, dtype=torch.float64)
                    wts[lo:hi] = torch.from_numpy(score(self.states[lo:hi]))
                    t0 = time.perf_counter()
                    e.dist.all_gather_into_tensor(wts, wts[e.rank * m:(e.rank + 1) * m].clone())
                    e.t_gather += time.perf_counter() - t0; e.n_gather += 1
                    w = wts[:self.n].numpy()
                else:
                    w = score(self.states)
                self.checksum = float(w.sum())

        def close(self):
            pass

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world
        self.t_gather, self.n_gather = 0.0, 0

    def n_ranks(self):
        return self.dist.get_world_size() if self.dist is not None else 1

    exchanges = ("collective", "peer")

    def make_filter(self, n, sharded, iters, exchange="collective"):
        # (the stub's peer exchange IS its collective; MTFHIP_BENCH_STUB_PEER_FAIL=<rank> makes that rank fail to set it up, the case
        # pf_strong_record has to survive without leaving the other ranks in a collective of their own)
        if sharded and exchange == "peer" and os.environ.get("MTFHIP_BENCH_STUB_PEER_FAIL", "") == str(self.rank):
            raise RuntimeError("stub: rank %d cannot map the peers' mailboxes" % self.rank)
        return PfStubEngine._Filter(self, n, sharded, iters)

    def estimate(self, pf):
        return [pf.checksum]

    def sync(self):
        pass

    def reduce_max(self, x):
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def kernel_times(self, pf):
        self.t_gather, self.n_gather = 0.0, 0
        pf.update()
        return {"score_kernel_ms": 0.0, "scan_select_ms": 0.0, "allgather_ms": self.t_gather / max(self.n_gather, 1) * 1e3, "checksum": pf.checksum}

    def close(self):
        pass


def pf_strong_peer_children(eng, dist, world, C, steps, iters_per_update):
    """the sharded_peer form on the GPUs: every rank runs its share in a child process (PfDeviceEngine.peer_row), the ranks' results are
    gathered, the slowest rank sets the time.  The children synchronise through the exchange itself: nobody gets through an iteration
    before everybody's weights have arrived."""
    import tempfile
    box = [tempfile.mkdtemp(prefix="mtfhip_pf_peer_") if eng.rank == 0 else None]
    if dist is not None:
        dist.broadcast_object_list(box, src=0)
    mine = eng.peer_row(C, steps, iters_per_update, box[0])
    if dist is not None:
        allr = [None] * world
        dist.all_gather_object(allr, mine)
    else:
        allr = [mine]
    if eng.rank == 0:
        import shutil
        shutil.rmtree(box[0], ignore_errors=True)
    bad = [(r, a["error"]) for r, a in enumerate(allr) if "error" in a]
    if bad:
        return {"error": "; ".join("rank %d: %s" % b for b in bad)}
    t = max(a["seconds"] for a in allr)
    return {"value": C * iters_per_update * steps / t, "unit": "candidates/s", "us_per_iteration": t / (steps * iters_per_update) * 1e6,
            "score_kernel_ms_per_rank": [a["score_kernel_ms"] for a in allr], "scan_select_ms_per_rank": [a["scan_select_ms"] for a in allr],
            "allgather_ms": max(a["allgather_ms"] for a in allr), "peer_exchanges_timed_per_rank": [a["peer_exchanges_timed"] for a in allr],
            "estimates_equal_across_ranks": all(a["estimate"] == allr[0]["estimate"] for a in allr), "estimate": allr[0]["estimate"],
            "process": "one child process per rank (detached communicator, mailbox handles through files): isolated from the headline"}


def pf_strong_record(eng, dist, world, sizes=((10000, 100), (100000, 30), (1000000, 8)), iters_per_update=10):
    """north_star's split (SM/src/PF.cc:195-306): the particle axis sharded over the ranks, ONE all-gather of the weights per iteration
    through the C-ABI collective (RCCL), scan + selection replicated.  For every size: the unsharded filter on one GPU (every rank runs
    it on its own GPU at the same time; rank 0's own clock is value(1)), then the sharded one over all ranks -- strong scaling, chained
    update() form (epsilon < 0: `iters_per_update` iterations enqueued back to back, one read-back), barrier + synchronize on both
    sides, MAX over ranks.  Collective calls inside: every rank must call this; rank 0 prints the result."""
    rec = {"n_ranks": eng.n_ranks(), "n_ranks_note": "the communicator's own count (mtfhip_comm_world of the RCCL communicator)",
           "iterations_per_update": iters_per_update, "sizes": [],
           "split": "rank r scores particles [r m, (r + 1) m), m = ceil(n / R); one in-place all-gather of 8 B per particle; cumulative weights, "
                    "resampling and estimate replicated on identical data (SM/src/PF.cc:262-306)",
           "expected_model_DESIGN_section_6": {str(k): v for k, v in PF_STRONG_MODEL.items()}}

    def all_ok(ok):
        # a step that can fail on ONE rank (mapping a peer's mailbox, a wait that gave up) is followed by this agreement, so that
        # either every rank goes on or none does -- nobody is left alone in the next collective
        return eng.reduce_max(0.0 if ok else 1.0) == 0.0

    labels = [("one_gpu", False, "collective")]
    if world > 1:
        labels += [("sharded" if x == "collective" else "sharded_" + x, True, x) for x in getattr(eng, "exchanges", ("collective",))]
    rec["exchanges"] = {"sharded": "one in-place RCCL all-gather per iteration (the default)",
                        "sharded_peer": "mtfhip_pf_set_exchange(PEER): the scoring kernel stores into every rank's mailbox, the scan waits "
                                        "for arrival counters; opt-in, first run between GPUs is this record's"}
    peer_gave_up = None
    for C, steps in sizes:
        row = {"particles": C, "updates_timed": steps}
        estimates = {}
        for label, sharded, exchange in labels:
            if exchange == "peer" and getattr(eng, "peer_in_child_process", False):
                if peer_gave_up:   # (one failure is the answer; the remaining sizes would only run into the same time-outs)
                    row[label] = {"error": "skipped: " + peer_gave_up}
                    continue
                row[label] = pf_strong_peer_children(eng, dist, world, C, steps, iters_per_update)
                if "error" in row[label]:
                    peer_gave_up = "the form failed at %d particles" % C
                if "estimate" in row[label]:
                    estimates[label] = row[label].pop("estimate")
                continue
            pf, err = None, None
            try:
                pf = eng.make_filter(C, sharded, iters_per_update, exchange)
            except Exception as e:   # noqa: BLE001  (reported in the record)
                err = "set-up: %s" % (e,)
            if not all_ok(err is None):
                row[label] = {"error": err or "another rank could not set this exchange up"}
                if pf is not None:
                    pf.close()
                continue
            # warm-up, then the timed updates: each phase ends in the agreement above, which is also the barrier of the timed region
            # (every rank starts its clock behind one all-reduce and stops it behind its own synchronize; MAX over ranks below)
            dt = 0.0
            for phase, count in (("warm-up", 3), ("run", steps)):
                try:
                    t0 = time.perf_counter()
                    for _ in range(count):
                        pf.update()
                    eng.sync()
                    dt = time.perf_counter() - t0
                except Exception as e:   # noqa: BLE001
                    err = "%s: %s" % (phase, e)
                if not all_ok(err is None):
                    err = err or "another rank failed while running this exchange"
                    break
            if err is not None:
                row[label] = {"error": err}
                pf.close()
                continue
            dt_max = eng.reduce_max(dt)
            mine = eng.kernel_times(pf)
            mine["estimate"] = eng.estimate(pf)
            if dist is not None:
                allr = [None] * world
                dist.all_gather_object(allr, mine)
            else:
                allr = [mine]
            # one_gpu: each rank timed its own filter; value(1) is rank 0's own clock, the slowest GPU is reported beside it
            t_use = dt if label == "one_gpu" else dt_max
            row[label] = {"value": C * iters_per_update * steps / t_use, "unit": "candidates/s", "us_per_iteration": t_use / (steps * iters_per_update) * 1e6,
                          "score_kernel_ms_per_rank": [a["score_kernel_ms"] for a in allr],
                          "scan_select_ms_per_rank": [a["scan_select_ms"] for a in allr],
                          "allgather_ms": max(a["allgather_ms"] for a in allr),
                          "estimates_equal_across_ranks": all(a["estimate"] == allr[0]["estimate"] for a in allr)}
            estimates[label] = allr[0]["estimate"]
            if "checksum" in mine:
                row[label]["checksums_equal_across_ranks"] = len({a["checksum"] for a in allr}) == 1
            if label == "one_gpu" and dist is not None:
                row[label]["us_per_iteration_slowest_gpu"] = dt_max / (steps * iters_per_update) * 1e6
            pf.close()
        # every form ran the same number of iterations from the same seed on the same frame: the estimates must be the same bits
        row["estimates_equal_across_forms"] = all(v == estimates["one_gpu"] for v in estimates.values()) if "one_gpu" in estimates else None
        for label in ("sharded", "sharded_peer"):
            if label in row and "value" in row[label]:
                row["speedup_valueN_over_value1" + ("" if label == "sharded" else "_peer")] = row[label]["value"] / row["one_gpu"]["value"]
        rec["sizes"].append(row)
    eng.close()
    return rec


def stub_main(args):
    """MTFHIP_BENCH_STUB=1: the distributed skeleton of the lk line on gloo with the device work replaced by stand-ins (stub_mode)."""
    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    B = args.targets

    def region():
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        time.sleep(1e-4 * args.steps * (1 + 0.1 * rank))   # stands in for batch.track: the slowest rank sets the time
        if dist is not None:
            dist.barrier()
        d = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([d], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d
    region_s = [region() for _ in range(max(1, args.repeats))]
    dt = float(sorted(region_s)[(len(region_s) - 1) // 2])
    pf_strong = None
    line = {"metric": "STUB (no device work) LK iters/sec", "value": B * world * args.steps / dt, "unit": "iters/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "stub",
            "config": {"workload": "stub", "parallelism": "replicas x%d" % world}, "pf_strong": None}
    if args.pf_strong == 1 or (args.pf_strong < 0 and world > 1):
        guardian = LineGuardian(line) if rank == 0 else None
        if os.environ.get("MTFHIP_BENCH_STUB_CRASH") == "1" and rank == 0:   # (tests/test_bench_cpu.py: a fault in native code inside the record)
            import signal
            os.kill(os.getpid(), signal.SIGSEGV)
        pf_strong = pf_strong_record(PfStubEngine(dist, rank, world), dist, world, sizes=((1000, 3), (1003, 2)), iters_per_update=2)
        if guardian is not None:
            guardian.release()
    if rank == 0:
        line["pf_strong"] = pf_strong
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def secondary_workload(args):
    """Secondary metrics of BASELINE.md section 3 (not the driver's headline line): config 3 grid
    patch-iterations/s, config 4 PF candidates/s (sharded over ranks, one RCCL all-gather of the scores per
    step = strong scaling), config 5 ESM+MI target-iterations/s (per-function entry points)."""
    import torch
    import mtf_amd
    from mtf_amd import synth
    from mtf_amd.sm import GridTracker, NTSearchMethod
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = mtf_amd.Context(local_rank, torch.cuda.current_stream(dev).cuda_stream)
    rng = np.random.default_rng(synth.DEFAULT_SEED + 2)
    out = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
           "vs_baseline": None, "dtype": "f64", "data": "synthetic"}

    def timed(fn):
        import gc
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize(dev)
        # these workloads loop in Python: a generation-2 collection over torch's ~10^6 live objects is a 30-40 ms pause
        # (seen as one outlier step in 200), so the collector is parked for the timed region
        gc.collect(); gc.disable()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        if os.environ.get("BENCH_STEP_TIMES"):   # debugging aid: distribution of the per-step wall times
            ts = []
            for _ in range(args.steps):
                t1 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t1)
            ts = np.array(ts) * 1e6
            print("step times us: median %.0f p95 %.0f max %.0f, first 5 %s, last 5 %s" % (np.median(ts), np.percentile(ts, 95), ts.max(),
                  np.round(ts[:5]), np.round(ts[-5:])), file=sys.stderr)
        else:
            for _ in range(args.steps):
                fn()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        if dist is not None:
            tm = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt = float(tm.item())
        return dt

    def kernel_pass(fn, n=40):
        # per-kernel average durations (HIP events on the launch stream, inside the library) come from a pass of their own:
        # two event records per launch are host work the timed region should not carry
        ctx.timing(True); ctx.timing_reset()
        for _ in range(max(n, 1)):
            fn()
        torch.cuda.synchronize(dev)
        ctx.timing(False)

    if args.workload == "grid":
        frame0 = synth.make_frame(1024, 1024)
        p_true = synth.random_small_homography(rng, 0.3)
        frame1 = synth.warp_frame(frame0, p_true, (512.0, 512.0))
        region = synth.square_corners(512, 512, 400)
        gt = GridTracker(ctx, grid_size=16, patch_size=25, max_iters=args.grid_iters, epsilon=-1.0)
        ctx.set_image(frame0)
        gt.initialize(region)
        ctx.set_image(frame1)
        patches = gt.patch_corners(region)

        gframe, gd_, gsm_ = gt.tracker.batch.grid_frame, gt.gd, gt.tracker.sm

        def step():
            gframe(gd_, gsm_, region)      # the patches laid over the region (the reference's layout), setRegion + update of every patch tracker: one C-ABI call, one launch
        dt = timed(step)
        kernel_pass(step)
        kms, kn = ctx.timing_get("iclk_track")
        # What a frame MOVES ("never credit bytes that were not moved"): the patch operands are read once and stay in
        # registers / LDS for the ten iterations -- I0 8 B + the template's SD rows 8 S B per pixel, the 64 x 64 texel window (16 KB), and
        # the template grid the region mode lays out for later callers (INIT_PTS 16 + INIT_HXY 16 + INIT_Z 8 B per pixel, written) --
        # ~19 MB per frame, not the 84 B x N x iterations (134 MB) an HBM roofline of the ICLK iteration would credit.  The frame is a
        # LATENCY figure: 256 workgroups, one per patch, ten dependent iterations each.
        moved = (8.0 + 8.0 * 6 + 40.0) * 625 * 256 + 256 * 64 * 64 * 4.0
        pm = pmc_secondary("grid", "k_iclk_track")
        frame_us = dt / args.steps * 1e6
        out.update({"metric": "grid patch-iterations/sec, GridTracker 256 patches ICLK+NCC+Affine 25x25",
                    "value": 256 * args.grid_iters * args.steps * world / dt, "unit": "patch-iters/s",
                    "ms_per_step": dt / args.steps * 1e3, "scaling": "weak",
                    "config": {"workload": "256 patches x %d ICLK iterations per step, one launch per frame; the reference's shipped patch layout "
                                           "(patch_centroid_inside = 1: 17 x 17 grid SSM points, patches centred on the cells' centroids, "
                                           "mtfhip_grid_frame = layout + setRegion + update in one C-ABI call)" % args.grid_iters,
                               "frame_us": frame_us, "kernel_us": kms * 1e3, "kernel_share_of_frame": kms * 1e3 / frame_us if frame_us > 0 else None},
                    "roofline": {"bound": "latency", "note": "256 workgroups, one per patch, ten dependent iterations each: neither HBM nor FP64 issue "
                                 "bounds a frame; achieved / frac are the bytes actually moved per launch over the kernel time, for what they are",
                                 "achieved": moved / (kms * 1e-3) / 1e9 if kms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": moved / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS if kms > 0 else None,
                                 "traffic": pm.get("traffic_bytes_per_launch") if pm else None, "traffic_note": getattr(pmc_secondary, "note", None),
                                 "kernel": "k_iclk_track", "avg_kernel_ms": kms, "launches_timed": kn, "moved_bytes_per_launch_estimate": moved,
                                 "kernel_sources_sha": kernel_sources_sha()}})
        # the same frame with the loop on the C++ side (mtf::hip::Grid, libmtfhost.so): no Python in the timed path
        if rank == 0:
            try:
                from mtf_amd import host
                cg = host.CppGridTracker(grid_size=16, patch_size=25, patch_sm=mtf_amd.SM_ICLK, patch_am=mtf_amd.AM_NCC, patch_ssm=mtf_amd.SSM_AFFINE,
                                         grid_ssm=mtf_amd.SSM_HOMOGRAPHY, reset_at_each_frame=2, max_iters=args.grid_iters, epsilon=-1.0, hess_type=0, device=local_rank)
                cg.set_image(frame0); cg.initialize(region); cg.set_image(frame1)
                cpp_us = cg.bench_frames(region, args.steps, 0)     # (its own untimed frames first, then EXACTLY args.steps frames between two clock reads; every frame ends with the host holding its results)
                out["config"]["cpp_driver"] = {"frame_us_c_abi_loop": cpp_us,
                                               "frame_us_grid_update_setregion_mode": cg.bench_frames(region, max(args.steps, 100), 1),
                                               "note": "mtfhip_grid_frame in a C++ loop | mtf::hip::Grid::update() = that launch + the all-points least-squares "
                                                       "estimator on the host + resetTrackers(setRegion), reset_at_each_frame = 2"}
                del cg
                # the video loop (setImage(next frame) + update() per frame, mtfhost_grid_bench_video) in the reset modes, and with the shipped
                # configuration's forward-backward estimation (Config/modules.cfg:80-82: reset_at_each_frame 1, fb_err_thresh 2, fb_reinit 1)
                video = {}
                for name, kw in (("reset2_setregion", dict(reset_at_each_frame=2)), ("reset1_reinit", dict(reset_at_each_frame=1)), ("reset0", dict(reset_at_each_frame=0)),
                                 ("shipped_reset1_fb2_reinit1", dict(reset_at_each_frame=1, fb_err_thresh=2.0, fb_reinit=1)),
                                 ("reset0_fb2_reinit0", dict(reset_at_each_frame=0, fb_err_thresh=2.0, fb_reinit=0))):
                    vg = host.CppGridTracker(grid_size=16, patch_size=25, patch_sm=mtf_amd.SM_ICLK, patch_am=mtf_amd.AM_NCC, patch_ssm=mtf_amd.SSM_AFFINE,
                                             grid_ssm=mtf_amd.SSM_HOMOGRAPHY, max_iters=args.grid_iters, epsilon=-1.0, hess_type=0, device=local_rank, **kw)
                    vg.set_image(frame0); vg.initialize(region)
                    u, im = vg.bench_video(frame0, frame1, max(args.steps, 50))
                    video[name] = {"update_us": u, "set_image_us": im}
                    del vg
                video["note"] = ("per frame setImage(the other of two 1024 x 1024 frames: a pageable host-to-device copy) + mtf::hip::Grid::update(); update_us is the "
                                 "time in update() alone; %d ICLK iterations per patch and pass (epsilon < 0), the backward pass of the forward-backward estimation included" % args.grid_iters)
                out["config"]["cpp_driver"]["video_loop"] = video
                if world == 1 and cpp_us > 0:
                    # r04 verdict item 6: no Python wrapper in the timed path -- the line's value is the C++ loop's; the Python loop's figure stays beside it
                    out["config"]["python_wrapper_loop"] = {"value": out["value"], "frame_us": frame_us}
                    out["value"] = 256 * args.grid_iters / (cpp_us * 1e-6)
                    out["ms_per_step"] = cpp_us * 1e-3
                    out["config"]["frame_us"] = cpp_us
                    out["config"]["kernel_share_of_frame"] = kms * 1e3 / cpp_us
                    out["config"]["timed_by"] = "mtfhost_grid_bench: mtfhip_grid_frame in a C++ loop (libmtfhost.so), steady_clock around exactly --steps frames"
            except Exception as e:   # (the host libraries are optional for this line)
                out["config"]["cpp_driver"] = {"error": str(e)[:200]}
        if rank == 0 and not args.no_cpu:
            import oracle_py as O
            ssm = O.SSM(O.SSM_AFF, 25, 25); am = O.AM(O.AM_NCC, 25, 25); am.set_curr_img(frame0)
            trk = O.Tracker(O.SM_ICLK, am, ssm, leven_marq=0, max_iters=args.grid_iters, epsilon=-1.0, hess_type=0)
            trk.initialize(patches[0]); am.set_curr_img(frame1)
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < args.cpu_seconds:
                ssm.set_corners(patches[n % 256]); n += trk.update()
            out["cpu_baseline"] = {"value": n / (time.perf_counter() - t0), "unit": "patch-iters/s", "cores": 1, "kind": "port",
                                   "sample": "%d ICLK+NCC+Affine 25x25 patch iterations" % n}
            out["parity"] = loop_parity(ctx, mtf_amd.SM_ICLK, mtf_amd.AM_NCC, mtf_amd.SSM_AFFINE, 25, frame0, frame1, patches[100], args.grid_iters, hess_type=0)
    elif args.workload == "pf":
        # config 4: the whole iteration of nt::PF::update's loop on the device -- sample generation (corner based homography
        # sampling, the reference's default), scoring, cumulative weights, multinomial resampling, estimate -- with the scoring
        # sharded over the ranks and ONE all-gather of the weights through the C-ABI collective (RCCL directly)
        from mtf_amd.sm import ParticleFilter, Comm
        frame0 = synth.make_frame(1024, 1024)
        corners = synth.square_corners(512, 512, 100)
        ctx.set_image(frame0)
        C = args.particles
        comm = Comm.torch_bootstrap(local_rank) if world > 1 else None
        KI = max(1, args.pf_iters)
        pf = ParticleFilter(ctx, mtf_amd.SSM_HOMOGRAPHY, 50, 50, n_particles=C, ssm_sigma=(1.0, 0.5, 1, 1, 1, 1, 1, 1), corner_based_sampling=1,
                            dynamic_model=0, update_type=1, likelihood_func=0, resampling_type=1, mean_type=0, likelihood_alpha=1.0,
                            max_iters=KI, epsilon=-1.0, seed=synth.DEFAULT_SEED, comm=comm)
        pf.batch.set_math_mode(mtf_amd.MATH_REPLAY if os.environ.get("MTFHIP_MATH", "fast")[0] == "r" else mtf_amd.MATH_FAST)
        pf.initialize(corners[None])

        def step():
            # KI = 1: one iteration + the read-back of its estimate (what nt::PF::update with max_iters = 1 is: the caller reads the
            # corners every frame).  KI > 1: update() with epsilon < 0 enqueues its KI iterations back to back, one read-back.
            pf.update()
        dt = timed(step)
        kernel_pass(step)
        kms, kn = ctx.timing_get("pf_score")
        rms, _ = ctx.timing_get("pf_resample"); gms, _ = ctx.timing_get("pf_allgather"); pms, pn = ctx.timing_get("pf_propose")
        pms = pms * pn / max(kn, 1)     # (a launch of its own only where the proposals could not be made ahead: per iteration)
        n_local = Comm.shard_bounds(C, world, rank)[1]
        flop_per_sample = 60.0   # ~60 FP64 flop per bilinear sample of a homography candidate
        tf = n_local * 2500 * flop_per_sample / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        us_iter = dt / (args.steps * KI) * 1e6
        out.update({"metric": "PF candidates/sec, PF+SSD+Homography 50x50, %d particles" % C,
                    "value": C * KI * args.steps / dt, "unit": "candidates/s", "ms_per_step": dt / args.steps * 1e3,
                    "scaling": "strong", "config": {"workload": "%d particles x 2500 px: proposal + scoring (sharded over %d rank(s), one in-place "
                                                                "all-gather of the weights) + cumulative weights + resampling + estimate; "
                                                                "%d iteration(s) per update(), one read-back per update()" % (C, world, KI),
                                                    "iterations_per_step": KI, "us_per_iteration": us_iter,
                                                    "score_kernel_ms": kms, "resample_kernels_ms": rms, "allgather_ms": gms, "propose_kernel_ms": pms,
                                                    # share of the scorer in the device time of an iteration (same event pass for all terms)
                                                    "scorer_share_of_iteration": kms / (kms + rms + gms + pms) if kms > 0 else None,
                                                    "samples_per_s": C * 2500 * KI * args.steps / dt},
                    "roofline": {"bound": "fp64-valu", "note": "the candidate scorer is not HBM bound (SURVEY 8d: ~1 MB of compulsory traffic for 25 M "
                                 "samples): fraction of the FP64 vector peak at ~60 flop per sample, and the texel gather rate served by L1 / L2",
                                 "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6,
                                 "traffic": (pmc_secondary("pf", "k_pf_score") or {}).get("traffic_bytes_per_launch") if C == 10000 and world == 1 else None,
                                 "traffic_note": getattr(pmc_secondary, "note", None) if C == 10000 and world == 1 else "PMC passes are taken at 10 000 particles on one rank",
                                 "kernel": "k_pf_score", "avg_kernel_ms": kms, "kernel_sources_sha": kernel_sources_sha(),
                                 "launches_timed": kn, "samples_per_launch": n_local * 2500,
                                 "texel_gather_GBs": n_local * 2500 * 16 / (kms * 1e-3) / 1e9 if kms > 0 else None}})
        if rank == 0 and world == 1 and not args.no_cpu:
            out["parity"] = pf_parity(ctx, frame0, corners)
        if rank == 0 and not args.no_cpu:
            import oracle_py as O
            ssm = O.SSM(O.SSM_HOM, 50, 50); am = O.AM(O.AM_SSD, 50, 50); am.set_curr_img(frame0)
            ssm.set_corners(corners); am.initialize_pix_vals(ssm.get("curr_pts")); am.initialize_similarity()
            pp = O.pf_params(500, corner_based_sampling=1, sigma=(1.0, 0.5, 1, 1, 1, 1, 1, 1))
            st, ar = np.zeros((500, 8)), np.zeros((500, 8))
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < args.cpu_seconds:
                st, ar, _, _, _ = O.pf_iteration(am, ssm, pp, st, ar, rng.normal(size=(500, 10)), rng.uniform(size=500), 0.0)
                n += 500
                if n % 5000 == 0:
                    st[:] = 0; ar[:] = 0; ssm.set_corners(corners)
            out["cpu_baseline"] = {"value": n / (time.perf_counter() - t0), "unit": "candidates/s", "cores": 1, "kind": "port",
                                   "sample": "%d particle evaluations of 2500 px incl. the 4-corner DLT per sample and resampling (500-particle filter)" % n}
        pf.close()
        if comm is not None:
            comm.close()
    elif args.workload == "nn":
        # nt::NN's dataset generation (SM/src/NT/NN.cc:131-191): C perturbed samples of a 50 x 50 template -> the C x N feature matrix the
        # index is built over, ONE launch (k_nn_dataset); rows block-partitioned over the ranks + one all-gather.  The one
        # batch kernel of the path that is bound by what it WRITES: 8 N C bytes (SSD / NCC features; MI 40 N C).
        import torch.distributed as tdist
        from mtf_amd import dist as mdist
        from mtf_amd.sm import NNDataset
        frame0 = synth.make_frame(1024, 1024)
        corners = synth.square_corners(512, 512, 100)
        ctx.set_image(frame0)
        C = args.samples
        am_id = {"ssd": mtf_amd.AM_SSD, "ncc": mtf_amd.AM_NCC, "mi": mtf_amd.AM_MI}[args.nn_am]
        am_kw = dict(mi_n_bins=args.mi_bins, mi_pou=args.mi_pou) if args.nn_am == "mi" else {}
        sigma = (0.01, 0.01, 2.0, 0.01, 0.01, 2.0, 1e-5, 1e-5)
        ds = NNDataset(ctx, am=am_id, ssm=mtf_amd.SSM_HOMOGRAPHY, resx=50, resy=50, n_samples=C, ssm_sigma=sigma, seed=synth.DEFAULT_SEED, am_params=am_kw)
        b = ds.batch
        b.set_corners(corners.reshape(1, 2, 4)); b.initialize_pix_vals()
        F = ds.feature_size()
        lo, cnt, m = mdist.padded_shard(C, rank, world)
        buf = torch.zeros((m * world, F), dtype=torch.float64, device=dev)
        perts = torch.zeros((C, 8), dtype=torch.float64, device=dev)
        d = b.nn_desc(C, sigma, None, synth.DEFAULT_SEED)
        mine = buf[rank * m:(rank + 1) * m]

        def step():
            b.nn_dataset_dev(d, mine.data_ptr(), lo, cnt, None, perts.data_ptr())
            if world > 1:
                ctx.synchronize()      # (the library's stream -> torch's: the collective reads the block)
                tdist.all_gather_into_tensor(buf, mine)
        dt = timed(step)
        kernel_pass(step)
        kms, kn = ctx.timing_get("nn_dataset")
        row_bytes = 8.0 * F
        gbs = row_bytes * cnt / (kms * 1e-3) / 1e9 if kms > 0 else None
        out.update({"metric": "NN dataset samples/sec, %s features of a 50x50 Homography template, %d samples" % (args.nn_am.upper(), C),
                    "value": C * args.steps / dt, "unit": "samples/s", "ms_per_step": dt / args.steps * 1e3, "scaling": "strong",
                    "config": {"workload": "nt::NN::generateDataset: %d samples x %d feature entries (perturbation draw, invertState, compositionalUpdate, updatePixVals, "
                                           "updateDistFeat) in one launch per rank, rows block-partitioned over %d rank(s)%s" %
                                           (C, F, world, " + one all-gather" if world > 1 else ""),
                               "kernel_us": kms * 1e3, "samples_per_rank": cnt, "feature_size": F, "dataset_bytes": row_bytes * C},
                    "roofline": {"bound": "hbm", "note": "write-bound: 8 B x feature_size x samples streamed out with non-temporal stores; the template grid (16 B/px) and "
                                 "the texels are re-read from L2 by every sample and not credited.  peak = the 8 TB/s HBM figure; a kernel that ONLY stores reaches "
                                 "4.5-6.4 TB/s on this device (tools/write_bw_test.hip, profiles/r06_write_bw.txt)",
                                 "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS if gbs else None,
                                 "traffic": (pmc_secondary("nn", "k_nn_dataset") or {}).get("traffic_bytes_per_launch") if (C == 10000 and world == 1 and args.nn_am == "ssd") else None,
                                 "traffic_note": getattr(pmc_secondary, "note", None),
                                 "kernel": "k_nn_dataset", "avg_kernel_ms": kms, "launches_timed": kn, "algorithmic_bytes_per_launch": row_bytes * cnt,
                                 "kernel_sources_sha": kernel_sources_sha()}})
        if rank == 0 and not args.no_cpu:
            import oracle_py as O
            o_ssm = O.SSM(O.SSM_HOM, 50, 50)
            o_am = O.AM({"ssd": O.AM_SSD, "ncc": O.AM_NCC, "mi": O.AM_MI}[args.nn_am], 50, 50, **({"n_bins": args.mi_bins, "pou": args.mi_pou} if args.nn_am == "mi" else {}))
            o_am.set_curr_img(frame0); o_ssm.set_corners(corners); o_am.initialize_pix_vals(o_ssm.get("curr_pts"))
            # parity in the same run: the first rows of this rank's block against the oracle's generateDataset on the perturbations the device drew
            k = min(64, cnt)
            torch.cuda.synchronize(dev)
            p_host = perts[lo:lo + k].cpu().numpy()
            want = O.nn_generate_dataset(o_am, o_ssm, p_host)
            got = mine[:k].cpu().numpy()
            err = float(np.abs(got - want).max())
            out["parity"] = {"pass": bool(err <= 1e-9), "budget": 1e-9, "max_abs_feature_difference": err, "rows": k,
                             "note": "device rows vs the oracle's NN::generateDataset on the device-drawn perturbations (pixel values 0..255)"}
            n, t0 = 0, time.perf_counter()
            chunk = rng.normal(size=(200, 8)) * np.asarray(sigma)
            while time.perf_counter() - t0 < args.cpu_seconds:
                O.nn_generate_dataset(o_am, o_ssm, chunk); n += len(chunk)
            out["cpu_baseline"] = {"value": n / (time.perf_counter() - t0), "unit": "samples/s", "cores": 1, "kind": "port",
                                   "sample": "%d samples of the oracle's generateDataset (50x50, same AM)" % n}
        ds.batch.close()
    elif args.workload == "dropin":
        # the literal drop-in boundary: C++ mtf::nt::ESM / FCLK / ICLK driving mtf::hip::HipAM / HipSSM through the
        # reference's virtuals (one C-ABI call per virtual), ONE target -- configs 1 / 2 as an MTF user runs them
        from mtf_amd import host
        H = W = 1024
        frame0 = synth.make_frame(H, W)
        frame1 = synth.warp_frame(frame0, synth.random_small_homography(rng, 0.5), (W / 2.0, H / 2.0))
        res = args.res
        c = synth.square_corners(W / 2.0, H / 2.0, float(res))
        sm_kind = {"esm": mtf_amd.SM_ESM, "fclk": mtf_amd.SM_FCLK, "iclk": mtf_amd.SM_ICLK}[args.sm]
        K = 50
        am_kind = {"ssd": mtf_amd.AM_SSD, "ncc": mtf_amd.AM_NCC}[args.am]
        tr = host.CppTracker(sm_kind, am=am_kind, resx=res, resy=res, max_iters=K, epsilon=-1.0, leven_marq=args.lm, device=local_rank,
                             device_loop=args.device_loop)
        tr.set_image(frame0); tr.initialize(c); tr.set_image(frame1)

        def step():
            tr.set_region(c)
            tr.update()
        dt = timed(step)
        how = "mtf::hip::LK (one C-ABI call per update(), loop on the device)" if args.device_loop else "nt::%s over the AM/SSM virtuals" % args.sm.upper()
        out.update({"metric": "drop-in LK iters/sec, one target, C++ %s, %s + %s + Homography %dx%d" % (how, args.sm.upper(), args.am.upper(), res, res),
                    "value": K * args.steps * world / dt, "unit": "iters/s", "ms_per_step": dt / args.steps * 1e3, "scaling": "weak",
                    "config": {"workload": "%d iterations per update(), %s, Levenberg-Marquardt %s, deferred fusion %s" %
                                           (K, "one C-ABI call per update()" if args.device_loop else "one C-ABI call per virtual",
                                            "on" if args.lm else "off", "off" if os.environ.get("MTFHIP_LAZY") == "0" else "on"),
                               "us_per_iter": dt / (K * args.steps) * 1e6}})
        if rank == 0 and not args.no_cpu:
            import oracle_py as O
            ssm = O.SSM(O.SSM_HOM, res, res); am = O.AM({"ssd": O.AM_SSD, "ncc": O.AM_NCC}[args.am], res, res); am.set_curr_img(frame0)
            trk = O.Tracker({"esm": O.SM_ESM, "fclk": O.SM_FCLK, "iclk": O.SM_ICLK}[args.sm], am, ssm, leven_marq=args.lm, max_iters=K, epsilon=-1.0)
            trk.initialize(c); am.set_curr_img(frame1)
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < args.cpu_seconds:
                ssm.set_corners(c); n += trk.update()
            out["cpu_baseline"] = {"value": n / (time.perf_counter() - t0), "unit": "iters/s", "cores": 1, "kind": "port",
                                   "sample": "%d iterations of one %dx%d target" % (n, res, res),
                                   "leven_marq": int(args.lm), "max_iters_per_update": K, "host_cpu": host_cpu_model()}
    else:  # mi
        H = W = 2048
        mc = args.channels == 3   # MCMI: MI over (pixel, channel) rows of a 32FC3 frame (AM/src/MCMI.cc)
        amp = dict(n_channels=3) if mc else {}
        if args.mi_bins != 8 or args.mi_pou:
            amp.update(mi_n_bins=args.mi_bins, mi_pou=args.mi_pou)
        amp = amp or None
        mi_kw = dict(mi_n_bins=args.mi_bins, mi_pou=args.mi_pou)
        frame0 = synth.make_frame_mc(H, W) if mc else synth.make_frame(H, W)
        p_true = synth.random_small_homography(rng, 0.3)
        frame1 = synth.warp_frame(frame0, p_true, (W / 2.0, H / 2.0))
        B, res = args.targets, args.res
        half = res / 2.0 + 12
        cx = rng.uniform(half, W - half, size=B); cy = rng.uniform(half, H - half, size=B)
        corners = np.stack([synth.square_corners(cx[i], cy[i], float(res)) for i in range(B)])
        ctx.set_image(frame0)
        KI = 1
        if args.mi_path == "fused":      # mtfhip_batch_iterate: four pixel-level launches per iteration + host solve
            from mtf_amd.sm import LKTracker
            nt = LKTracker(ctx, mtf_amd.SM_ESM, mtf_amd.SSM_HOMOGRAPHY, res, res, B, host_solve=True, am=mtf_amd.AM_MI, am_params=amp,
                           max_iters=1, epsilon=-1.0, leven_marq=0, materialize=0)
        elif args.mi_path == "device":   # mtfhip_batch_track: the same passes, solve + update on the device, KI iterations per call
            from mtf_amd.sm import LKTracker
            KI = 10
            nt = LKTracker(ctx, mtf_amd.SM_ESM, mtf_amd.SSM_HOMOGRAPHY, res, res, B, host_solve=False, am=mtf_amd.AM_MI, am_params=amp,
                           max_iters=KI, epsilon=-1.0, leven_marq=0, materialize=0)
        else:                            # one C-ABI call per reference virtual
            nt = NTSearchMethod(ctx, mtf_amd.SM_ESM, mtf_amd.AM_MI, mtf_amd.SSM_HOMOGRAPHY, res, res, B, max_iters=1,
                                epsilon=-1.0, leven_marq=0, am_params=amp)
        nt.initialize(corners)
        ctx.set_image(frame1)
        dt = timed(nt.update)
        kernel_pass(nt.update)
        p1, n1 = ctx.timing_get("mi_pass1"); p2, n2 = ctx.timing_get("mi_pass2")
        recompute = n1 > 0
        if recompute:     # the recompute form: pass 1 reads 28 B/px (texels 4, I0 8, grid point 16), pass 2 44 B/px (+ dI0_dx 16); nothing written
            Cn = 3 if mc else 1   # rows are (pixel, channel) pairs; the pixel's grid point (16 B) is shared by its rows
            bpr = (12.0 + 16.0 / Cn) + (28.0 + 16.0 / Cn)
            mi_bytes = bpr * res * res * Cn * B
            # The passes are bound by FP64 VALU issue at two waves per SIMD (r04 PMC: one VALU instruction per 5.8 cycles and SIMD against the
            # measured 5.9 ceiling), not by HBM: `frac` = the cycles the counted VALU instructions need at that ceiling / the cycles the
            # kernels took (both passes; counters from profiles/pmc_secondary_latest.json, quoted only for the kernel sources being run);
            # the HBM figure rides along as hbm_frac.
            standard = (not mc) and B == 64 and res == 400 and args.mi_bins == 8 and not args.mi_pou
            pm1 = pmc_secondary("mi", "k_mi_pass_hist") if standard else None
            pm2 = pmc_secondary("mi", "k_mi_pass_grad_hess") if standard else None
            note = getattr(pmc_secondary, "note", None) if standard else "PMC passes are taken on the 64 x 400 x 400 single-channel workload"
            valu = (pm1.get("SQ_INSTS_VALU", 0) + pm2.get("SQ_INSTS_VALU", 0)) if (pm1 and pm2) else None
            issue_s = valu * FP64_ISSUE_CYCLES[2] / N_SIMDS / NOMINAL_HZ if valu else None
            hbm_GBs = mi_bytes / ((p1 + p2) * 1e-3) / 1e9
            mi_roof = {"bound": "fp64-valu-issue", "note": "VALU issue at two waves per SIMD (one FP64 instruction per %.2f cycles and SIMD at the nominal "
                       "2.4 GHz, profiles/r04_fp64_rates.txt); 72 B/px (multi-channel: 50.7 B per (pixel, channel) row) are moved per iteration" % FP64_ISSUE_CYCLES[2],
                       "achieved": valu / ((p1 + p2) * 1e-3) / 1e9 if valu else None, "peak": N_SIMDS * NOMINAL_HZ / FP64_ISSUE_CYCLES[2] / 1e9,
                       "unit": "G VALU wave-instructions/s", "frac": issue_s / ((p1 + p2) * 1e-3) if issue_s else None,
                       "valu_instructions_per_launch": {"pass1": pm1.get("SQ_INSTS_VALU") if pm1 else None, "pass2": pm2.get("SQ_INSTS_VALU") if pm2 else None},
                       "traffic": (pm1["traffic_bytes_per_launch"] + pm2["traffic_bytes_per_launch"]) if (pm1 and pm2 and "traffic_bytes_per_launch" in pm1 and "traffic_bytes_per_launch" in pm2) else None,
                       "traffic_note": note, "algorithmic_bytes_per_iteration": mi_bytes,
                       "hbm_GBs": hbm_GBs, "hbm_frac": hbm_GBs / HBM_PEAK_GBS, "kernel": "k_mi_pass_hist + k_mi_pass_grad_hess",
                       "avg_kernel_ms": {"pass1": p1, "pass2": p2}, "launches_timed": n1, "algorithmic_bytes_per_row": bpr, "rows_per_target": res * res * Cn,
                       "kernel_sources_sha": kernel_sources_sha()}
        else:
            mi_roof = None
        out.update({"roofline": mi_roof})
        out.update({"metric": "ESM+MI target-iterations/sec, %dx%d, %d targets" % (res, res, B),
                    "value": B * KI * args.steps * world / dt, "unit": "target-iters/s", "ms_per_step": dt / args.steps * 1e3,
                    "scaling": "weak", "config": {"workload": "ESM+%s(%d bins%s)+Homography %dx%d%s x %d targets per GPU, %s" %
                                                  ("MCMI" if mc else "MI", args.mi_bins, ", partition of unity" if args.mi_pou else "", res, res, "x3" if mc else "", B, {"fused": "fused MI passes (mtfhip_batch_iterate) + host solve",
                                                                 "device": "fused MI passes, solve + update on the device (mtfhip_batch_track), %d iterations per step" % KI,
                                                                 "interface": "per-function entry points"}[args.mi_path]),
                                                  "iterations_per_step": KI}})
        if rank == 0 and not args.no_cpu and not mc:
            import oracle_py as O
            ssm = O.SSM(O.SSM_HOM, res, res); am = O.AM(O.AM_MI, res, res, n_bins=args.mi_bins, pou=args.mi_pou); am.set_curr_img(frame0)
            trk = O.Tracker(O.SM_ESM, am, ssm, leven_marq=0, max_iters=2, epsilon=-1.0)
            trk.initialize(corners[0]); am.set_curr_img(frame1)
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < args.cpu_seconds:
                ssm.set_corners(corners[0]); n += trk.update()
            out["cpu_baseline"] = {"value": n / (time.perf_counter() - t0), "unit": "target-iters/s", "cores": 1, "kind": "port",
                                   "sample": "%d ESM+MI iterations of one %dx%d target" % (n, res, res)}
            if args.mi_path == "device":
                out["parity"] = loop_parity(ctx, mtf_amd.SM_ESM, mtf_amd.AM_MI, mtf_amd.SSM_HOMOGRAPHY, res, frame0, frame1, corners[0], 4, am_kw=mi_kw)
                out["parity"]["budget_note"] = ("MI's update is ill-conditioned: the two oracles themselves differ by 1e-5 .. 1e-4 in dp "
                                                "(tests/test_oracle_relations.py::test_mi_update_noise_floor); H and g are the 1e-5 quantities")
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()    # handles released while the HIP runtime is whole (the library's atexit hook would do the same)
    if dist is not None:
        dist.destroy_process_group()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--pf-peer-child":
        return pf_peer_child(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="lk", choices=["lk", "grid", "pf", "mi", "nn", "dropin"],
                    help="lk = the headline metric (default); the others are the secondary metrics of BASELINE.md")
    ap.add_argument("--particles", type=int, default=10000)
    ap.add_argument("--samples", type=int, default=10000, help="nn workload: dataset samples (the shipped Config/modules.cfg nn_n_samples: 1000 .. 100 000 in use)")
    ap.add_argument("--nn-am", default="ssd", choices=["ssd", "ncc", "mi"], help="nn workload: appearance model whose updateDistFeat fills the rows")
    ap.add_argument("--pf-iters", type=int, default=1, help="pf workload: iterations per update() call (epsilon < 0: enqueued back to back)")
    ap.add_argument("--grid-iters", type=int, default=10)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--targets", type=int, default=64, help="targets per GPU (each 200x200)")
    ap.add_argument("--res", type=int, default=200)
    ap.add_argument("--channels", type=int, default=1, choices=[1, 3],
                    help="lk workload: 3 = the multi-channel models (MCSSD / MCNCC) on a 32FC3 frame through k_fused_mc")
    ap.add_argument("--sm", default="esm", choices=["esm", "fclk", "iclk"])
    ap.add_argument("--am", default="ssd", choices=["ssd", "ncc"], help="appearance model (lk and dropin workloads)")
    ap.add_argument("--mi-path", default="device", choices=["fused", "device", "interface"],
                    help="mi workload: fused iterate + host solve, the device-side loop, or one call per virtual")
    ap.add_argument("--mi-bins", type=int, default=8, help="mi workload: n_bins (BASELINE.json config 5: 8; the shipped Config/modules.cfg:115: 10)")
    ap.add_argument("--mi-pou", type=int, default=0, help="mi workload: partition of unity (the shipped Config/modules.cfg:117: 1)")
    ap.add_argument("--lm", type=int, default=1, help="dropin workload: Levenberg-Marquardt (the reference's class default is on)")
    ap.add_argument("--device-loop", action="store_true", help="dropin workload: the C++ search method is mtf::hip::LK (whole update() in one C-ABI call)")
    ap.add_argument("--mode", default="full", choices=["full", "lean"],
                    help="full: It, dIt_dx, Jt materialised in HBM as the AM/SSM interface exposes them; lean: registers only")
    ap.add_argument("--math", default="fast", choices=["fast", "replay"],
                    help="arithmetic of the non-materialising kernels: fast = FMA / one reciprocal per point / closed-form gradient "
                         "(within 1e-5), replay = the reference's rounding bit for bit (mtfhip_batch_set_math_mode)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-lean", action="store_true", help="skip the lean and single-target sub-records of the headline line")
    ap.add_argument("--event-stride", type=int, default=0, help="lk workload: HIP events around every n-th fused launch inside the timed regions; 0 (default) = none -- "
                    "at the driver's 20 steps a region is 1 ms and even every 4th launch instrumented costs 12 %% of it (1.19 M -> 1.05 M iters/s, r06 A/B)")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each (value = the median region)")
    ap.add_argument("--configs", type=int, default=-1, help="append the `configs` block (BASELINE.json's configs 3 / 4 / 5 at reduced step counts, child "
                    "processes) to the lk line: -1 = at --gpus 1 with the default workload shape (default), 0 = never, 1 = always")
    ap.add_argument("--pf-strong", type=int, default=-1, help="append the sharded-filter strong-scaling record (pf_strong) to the lk line: "
                    "-1 = when --gpus > 1 (default), 0 = never, 1 = always (on one GPU it exercises the code path with one rank)")
    args = ap.parse_args()
    self_spawn_if_needed(args, sys.argv[1:])
    if stub_mode():
        return stub_main(args)
    if args.workload != "lk":
        if args.workload == "mi" and args.res == 200 and args.targets == 64:
            args.res, args.targets = 400, 64   # config 5 of BASELINE.json
        return secondary_workload(args)

    import torch
    import mtf_amd
    from mtf_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or from a clean environment" % (args.gpus, world, args.gpus))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    res, B = args.res, args.targets
    H = W = 1024
    CH = args.channels
    frame0 = synth.make_frame(H, W) if CH == 1 else synth.make_frame_mc(H, W)
    rng = np.random.default_rng(synth.DEFAULT_SEED + 1)
    p_true = synth.random_small_homography(rng, 0.5)
    frame1 = synth.warp_frame(frame0, p_true, (W / 2.0, H / 2.0))
    # B target regions of res x res pixels (1:1 sampling) spread over the frame
    half = res / 2.0 + 12
    cx = rng.uniform(half, W - half, size=B)
    cy = rng.uniform(half, H - half, size=B)
    corners = np.stack([synth.square_corners(cx[i], cy[i], float(res)) for i in range(B)])

    stream = torch.cuda.current_stream(dev).cuda_stream
    ctx = mtf_amd.Context(local_rank, stream)
    f0 = torch.from_numpy(frame0).to(dev)
    f1 = torch.from_numpy(frame1).to(dev)
    sm_kind = {"esm": mtf_amd.SM_ESM, "fclk": mtf_amd.SM_FCLK, "iclk": mtf_amd.SM_ICLK}[args.sm]
    materialize = 1 if args.mode == "full" else 0
    am_kind = {"ssd": mtf_amd.AM_SSD, "ncc": mtf_amd.AM_NCC}[args.am]
    batch = mtf_amd.Batch(ctx, am_kind, mtf_amd.SSM_HOMOGRAPHY, res, res, B, n_channels=CH)
    batch.set_math_mode(mtf_amd.MATH_FAST if args.math == "fast" else mtf_amd.MATH_REPLAY)
    batch.set_corners(corners)
    def set_frame(dev_t, host_a):
        if CH == 1:
            ctx.set_image_device(dev_t.data_ptr(), H, W, keep=dev_t)
        else:
            ctx.set_image(host_a)     # 32FC3: uploaded once, outside the timed region
    set_frame(f0, frame0)
    sm = mtf_amd.sm_desc(sm_kind, materialize=materialize, leven_marq=0, epsilon=-1.0, max_iters=1)
    batch.init_template(sm)
    set_frame(f1, frame1)

    def run(n_iters):
        sm.max_iters = n_iters
        batch.set_region(corners, sm)   # restart every target from its initial region (setRegion of the search method)
        return batch.track(sm)

    # W untimed warm-up steps, and then more of them until the device has been busy for >= 50 ms: a fresh box needs that long to
    # leave its idle clocks, whatever --warmup says (the driver's --steps 20 --warmup 5 is a 1.7 ms run otherwise)
    run(max(1, args.warmup))
    torch.cuda.synchronize(dev)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.05:
        run(50)
        torch.cuda.synchronize(dev)
    # the W warm-up steps directly in front of the K timed ones (the driver contract's order): setRegion + W iterations, then every
    # target is put back on its initial region by setState(0) -- the SSM's own reset; a second setRegion here would re-derive the
    # template Jacobian of ESM (164 MB written) and leave the timed steps to start on a cold Infinity Cache
    # `--repeats` such regions (r03 verdict: one 1 ms region is one sample; the builder's own eight runs spanned 5 %): each is W untimed
    # warm-up steps + EXACTLY K timed steps between barrier + synchronize, MAX over ranks; value = the median region, min / max beside it
    # r05 verdict item 8: roofline.frac and value from the SAME execution -- HIP events around every --event-stride-th fused launch INSIDE the
    # timed region (mtfhip_timing_enable(ctx, n > 1): two event records per sampled launch, on the queue the launch goes to)
    def timed_region():
        run(max(1, args.warmup))
        batch.set_state(np.zeros((B, 8)))
        sm.max_iters = args.steps
        if args.event_stride > 0:
            ctx.timing(args.event_stride if args.event_stride > 1 else True)
            ctx.timing_reset()          # (drains and synchronises: in front of the barrier, outside the clock)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        n_it, final = batch.track(sm)
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        d = time.perf_counter() - t0
        assert int(n_it.min()) == args.steps and int(n_it.max()) == args.steps
        km_r, kn_r = (0.0, 0)
        if args.event_stride > 0:
            ctx.timing(False)
            km_r, kn_r = ctx.timing_get("fused_lk")
        if dist is not None:
            tmax = torch.tensor([d], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            d = float(tmax.item())
        return d, km_r, kn_r
    regions = [timed_region() for _ in range(max(1, args.repeats))]
    region_s = [r[0] for r in regions]
    dt, region_kern_ms, region_kern_n = sorted(regions)[(len(regions) - 1) // 2]   # (an actual region: ms_per_step x steps is one region's wall time)
    dt = float(dt)
    # kernel duration for the roofline: hipEvents around EVERY fused launch of a second, untimed pass of the same loop (at least 200
    # launches whatever --steps is -- with 40 the first launches after the region reset still weighed on the average, 0.720-0.727
    # against 0.734-0.738 for the same code at --steps 200; the events cost ~5 % of a step, which is why they stay out of the timed
    # region above)
    k_steps = max(200, args.steps)
    ctx.timing(1)
    ctx.timing_reset()
    t_ev = time.perf_counter()
    n_ev, _ = run(k_steps)
    torch.cuda.synchronize(dev)
    ev_pass_s = time.perf_counter() - t_ev   # (includes the setRegion in front of the loop: a few hundred microseconds of k_steps steps)
    kern_ms, kern_n = ctx.timing_get("fused_lk")
    # launches of the fused kernel overlap when the loop keeps two chunks of targets in flight on two queues: the time the kernel was
    # executing is the union of the launches' intervals (equal to their sum on one queue)
    busy_ms, busy_n = ctx.timing_get_busy("fused_lk")
    fin_ms, fin_n = ctx.timing_get("finish_track")
    ctx.timing(False)
    # the lean variant of the same workload (nothing materialised: the form the device-side loop needs): FP64-issue / latency bound
    lean = None
    if rank == 0 and args.mode == "full" and not args.no_lean and CH == 1:
        sm_l = mtf_amd.sm_desc(sm_kind, materialize=0, leven_marq=0, epsilon=-1.0, max_iters=k_steps)
        batch.set_region(corners, sm_l); batch.track(sm_l)
        torch.cuda.synchronize(dev)
        batch.set_region(corners, sm_l)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter(); batch.track(sm_l); torch.cuda.synchronize(dev); dt_l = time.perf_counter() - t1
        ctx.timing(1); ctx.timing_reset()
        batch.set_region(corners, sm_l); batch.track(sm_l); torch.cuda.synchronize(dev)
        lk_ms, lk_n = ctx.timing_get("fused_lk")
        ctx.timing(False)
        bpp_l = algorithmic_bytes_per_pixel(args.sm, 0, j0_recompute=os.environ.get("MTFHIP_J0_RECOMPUTE", "1") != "0" and args.sm in ("esm", "iclk"))
        flops_px = 190.0   # ~190 FP64 flop per pixel-iteration (5 samples, chain rule, SD row, 44 accumulations)
        lean = {"value": B * k_steps / dt_l, "unit": "iters/s", "us_per_step": dt_l / k_steps * 1e6, "kernel_us": lk_ms * 1e3,
                "launches_timed": lk_n, "math": args.math, "bound": "fp64-valu / latency (not HBM)",
                "algorithmic_bytes_per_pixel": bpp_l, "frac_of_hbm_peak": bpp_l * res * res * B / (lk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if lk_ms > 0 else None,
                "frac_of_fp64_vector_peak_78.6TF": flops_px * res * res * B / (lk_ms * 1e-3) / 78.6e12 if lk_ms > 0 else None}

    # the single-target, latency-bound figure -- configs 1 / 2 as one MTF tracker runs them (one target, the whole
    # update() on the device: a pixel pass + a finish launch per iteration)
    single = None
    if rank == 0 and args.mode == "full" and not args.no_lean and CH == 1 and B > 1:
        single = {}
        for s_res, s_mat, s_name in ((200, 1, "200x200_full"), (200, 0, "200x200_lean"), (50, 0, "50x50_lean_config1_shape")):
            b1 = mtf_amd.Batch(ctx, am_kind, mtf_amd.SSM_HOMOGRAPHY, s_res, s_res, 1)
            b1.set_math_mode(mtf_amd.MATH_FAST if args.math == "fast" else mtf_amd.MATH_REPLAY)
            c1 = synth.square_corners(W / 2.0 + 0.37, H / 2.0 - 0.21, float(s_res if s_res == 200 else 100))[None]
            b1.set_corners(c1)
            set_frame(f0, frame0)
            sm1 = mtf_amd.sm_desc(sm_kind, materialize=s_mat, leven_marq=0, epsilon=-1.0, max_iters=200)
            b1.init_template(sm1)
            set_frame(f1, frame1)
            ts = []
            for _ in range(6):
                b1.set_region(c1, sm1)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter(); b1.track(sm1); torch.cuda.synchronize(dev); ts.append(time.perf_counter() - t1)
            ts = sorted(ts[1:])
            single[s_name] = {"value": 200 / ts[len(ts) // 2], "unit": "iters/s", "us_per_iteration": ts[len(ts) // 2] / 200 * 1e6,
                              "min_us": ts[0] / 200 * 1e6, "max_us": ts[-1] / 200 * 1e6}
            b1.close()
        single["note"] = ("one target, %s, solve + update on the device, 200 iterations per call, median of 5 calls: two dependent launches per "
                          "iteration (pixel pass + finish) -- latency bound, not a roofline figure" % args.sm.upper())
        set_frame(f1, frame1)

    out = None
    if rank == 0:
        N = res * res * CH       # rows of the per-pixel arrays: (pixel, channel) pairs for the multi-channel models
        j0_rec = os.environ.get("MTFHIP_J0_RECOMPUTE", "1") != "0" and args.sm in ("esm", "iclk")
        bpp = algorithmic_bytes_per_pixel(args.sm, materialize, j0_recompute=j0_rec)
        if CH > 1:
            bpp = bpp - 16 + 16.0 / CH   # a pixel's grid point is shared by its C rows
        per_launch = batch.track_targets_per_launch(sm)   # all B, or an Infinity-Cache sized chunk of them
        if B % per_launch:                                 # a ragged last chunk would mix two launch sizes in the average
            per_launch = B / float(-(-B // per_launch))
        bytes_per_launch = float(bpp) * N * per_launch
        queues = batch.track_queues(sm)
        # algorithmic bytes of the timed launches / time during which the kernel was executing.  One queue: = bytes per launch / average
        # launch duration.  Two queues: the launches overlap, one of them sees bytes_per_launch / avg_kernel_ms, together they reach this
        ev_achieved = bytes_per_launch * busy_n / (busy_ms * 1e-3) / 1e9 if busy_ms > 0 else 0.0     # the separate, fully instrumented event pass
        ev_in_flight = kern_ms * kern_n / busy_ms if busy_ms > 0 else 0.0
        # the timed region itself (the execution `value` comes from): sampled launch durations, and the region's algorithmic bytes over its wall time
        launches_per_step = B / float(per_launch)
        same = region_kern_n > 0 and region_kern_ms > 0
        launch_ms_per_step = region_kern_ms * launches_per_step if same else None
        in_flight = launch_ms_per_step / (dt / args.steps * 1e3) if same else None
        # `achieved` / `frac` are the timed region's own: its algorithmic bytes over its wall time -- the execution `value` is read from, so the two
        # cannot disagree (frac = value x bytes per target-iteration / peak).  The wall time contains the solves, cold first passes and call
        # overhead: a lower bound on the rate while the fused kernel runs.  The event-instrumented second pass rides along as `event_pass`.
        achieved = float(bpp) * N * B * args.steps / dt / 1e9
        out = {
            "metric": "LK iters/sec (warp+grad+Hessian), ESM+%s%s+Homography 200x200" % ("MC" if CH > 1 else "", args.am.upper()),
            "value": B * world * args.steps / dt,
            "value_min": B * world * args.steps / max(region_s), "value_max": B * world * args.steps / min(region_s),
            "timed_regions": {"count": len(region_s), "ms": [r * 1e3 for r in region_s],
                              "note": "each region = W warm-up steps (untimed) + K timed steps between barrier + synchronize, MAX over ranks; value and "
                                      "ms_per_step are the median region's"},
            "unit": "iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s+%s%s+Homography %dx%d%s, %d independent targets per GPU, %s mode, solve+update on device"
                                   % (args.sm.upper(), "MC" if CH > 1 else "", args.am.upper(), res, res, "x%d" % CH if CH > 1 else "", B, args.mode),
                       "targets_per_gpu": B, "n_pix": N, "channels": CH, "mode": args.mode, "frame": "%dx%d float32%s" % (H, W, " x %d channels" % CH if CH > 1 else ""),
                       "parallelism": "replicas x%d" % world},
            "roofline": {"bound": "hbm+infinity-cache" if materialize else "fp64-valu / latency (hbm fraction reported for what it is)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # the part of `frac` that is DRAM: the read set (grid points, I0, dI0_dx) is re-read every iteration and stays in the
                         # 256 MB Infinity Cache, only the non-temporal stores of the materialised arrays reach HBM
                         "dram_frac": achieved / HBM_PEAK_GBS * float(88 if materialize and args.sm != "iclk" else (8 if materialize else 0)) / bpp,
                         "traffic": pmc_traffic(args.sm, args.mode, res, B, per_launch) if (args.am == "ssd" and CH == 1) else None,
                         "kernel": "k_fused_%s" % args.am, "avg_kernel_ms": region_kern_ms if same else kern_ms, "launches_timed": region_kern_n if same else kern_n,
                         "avg_kernel_ms_source": ("HIP events around every %d-th fused launch inside the median timed region" % args.event_stride) if same else
                                                 "the event pass below (no events inside the timed regions: --event-stride 0)",
                         "queues": queues, "launches_in_flight": in_flight if same else ev_in_flight,
                         "per_launch_GBs": bytes_per_launch / ((region_kern_ms if same else kern_ms) * 1e-3) / 1e9 if (region_kern_ms if same else kern_ms) > 0 else None,
                         "launch_ms_per_step": launch_ms_per_step, "ms_per_step": dt / args.steps * 1e3,
                         "formula": "same execution as `value`: achieved = algorithmic bytes of the K timed steps (bytes_per_launch x launches) / the median "
                                    "region's wall time; that wall time contains the kernel's busy time plus the solves, so achieved is a lower bound on the "
                                    "rate while the fused kernel runs and kernel time per step <= ms_per_step by construction",
                         "avg_finish_ms": fin_ms,
                         # a second, untimed execution with events around EVERY launch (the union of their intervals = the time the kernel was
                         # executing): the instrumented loop is the slower one, its figure rides along
                         "event_pass": {"steps": k_steps, "ms_per_step_wall": ev_pass_s / k_steps * 1e3, "kernel_busy_ms_per_step": busy_ms / k_steps,
                                        "avg_kernel_ms": kern_ms, "launches_timed": kern_n, "launches_in_flight": ev_in_flight, "achieved": ev_achieved,
                                        "frac": ev_achieved / HBM_PEAK_GBS, "formula": "bytes_per_launch x launches / union of the launches' event intervals"},
                         "algorithmic_bytes_per_pixel": bpp, "j0_rows": "rebuilt from dI0_dx" if j0_rec else "read back", "bytes_per_launch": bytes_per_launch,
                         "targets_per_launch": per_launch,
                         # what the figure means: algorithmic bytes / kernel time.  The read set of a launch (grid points, I0, dI0_dx:
                         # 40 B/px) is re-read every iteration and fits the 256 MB Infinity Cache, so after the first iteration it is
                         # served from there; the 88 B/px of non-temporal stores do reach HBM.
                         "infinity_cache_resident_read_bytes": float(bpp - (88 if materialize and args.sm != "iclk" else (8 if materialize else 0))) * N * per_launch,
                         "hbm_write_bytes": float(88 if materialize and args.sm != "iclk" else (8 if materialize else 0)) * N * per_launch,
                         "timing": "wall clock of the timed regions (events inside them with --event-stride n: every n-th fused launch, now %d) and hipEvents around every "
                                   "launch of an untimed second pass (%d launches, on the queue each launch goes to); rocprofv3 --kernel-trace of the same command with --no-cpu --no-lean --configs 0 (the lean, "
                                   "single-target and configs sub-records launch the same kernel at other sizes): profiles/r06_kernel_stats.csv (one queue: "
                                   "profiles/r06_kernel_stats_one_queue.csv)" % (args.event_stride, kern_n),
                         "traffic_source": "rocprofv3 --pmc passes of tools/profile_round.sh on these kernel sources (profiles/pmc_latest.json: "
                                           "(2 x FETCH_SIZE + WRITE_SIZE) KiB, the guide's gfx950 correction); counters cannot be read inside a timed run",
                         "traffic_commit": getattr(pmc_traffic, "commit", None), "traffic_note": getattr(pmc_traffic, "note", None),
                         "kernel_sources_sha": kernel_sources_sha()},
        }
        if lean is not None:
            out["lean"] = lean
        if single is not None:
            out["single_target"] = single
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, res, frame0, frame1, corners[0], args.am)
            out["parity"] = parity_gate(ctx, args.am, res, frame0, frame1, corners[0])
        elif not args.no_cpu:
            out["cpu_baseline"] = None
    # north_star's multi-GPU split is the particle axis: the strong-scaling record of the sharded filter rides on the N > 1 line.  It is
    # the one part of this file that has never met more than one real GPU (RCCL with world > 1 needs a multi-GPU node), so it cannot
    # take the headline down with it: an exception becomes {"error": ...}, and a rank that stops answering is cut off by a watchdog
    # that prints the line without the record and ends the process on every rank.
    if args.pf_strong == 1 or (args.pf_strong < 0 and world > 1):
        import threading

        guardian = None

        def give_up():
            if guardian is not None:
                guardian.release()
            if out is not None:
                out["pf_strong"] = {"error": "no answer within %d s (a rank hung in the sharded-filter record); headline unaffected" % PF_STRONG_WATCHDOG_S}
                print(json.dumps(out), flush=True)
            os._exit(0)
        dog = threading.Timer(PF_STRONG_WATCHDOG_S, give_up)
        dog.daemon = True
        dog.start()
        if out is not None:
            guardian = LineGuardian(out)
        try:
            rec = pf_strong_record(PfDeviceEngine(ctx, dev, dist, world, local_rank), dist, world)
        except Exception as e:   # noqa: BLE001
            rec = {"error": "%s: %s" % (type(e).__name__, e)}
        dog.cancel()
        if guardian is not None:
            guardian.release()
        if out is not None:
            out["pf_strong"] = rec
    if out is not None and (args.configs == 1 or (args.configs < 0 and world == 1 and not args.no_cpu and CH == 1 and args.mode == "full")):
        try:
            out["configs"] = configs_block(args)
        except Exception as e:   # noqa: BLE001
            out["configs"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if out is not None:
        print(json.dumps(out), flush=True)
    ctx.close()    # handles released while the HIP runtime is whole (the library's atexit hook would do the same)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
