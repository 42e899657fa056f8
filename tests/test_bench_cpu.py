"""CPU: `python bench.py --gpus 2` from a clean environment -- the command the driver types -- launches its own two ranks
(torch.distributed.run on 127.0.0.1), rendezvouses (gloo here, RCCL on the GPUs), times between barriers with MAX over ranks,
and rank 0 alone prints ONE JSON line that carries the sharded-filter strong-scaling record (pf_strong).  Device work is replaced
by stand-ins (MTFHIP_BENCH_STUB=1: bench.stub_main / PfStubEngine); the sharded layout is the C ABI's
(tests/test_dist_cpu.py::test_pf_shard_bounds_partition)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                            "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    env["MTFHIP_BENCH_STUB"] = "1"
    return env


def test_bench_gpus2_self_spawns_and_reports_pf_strong():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--repeats", "3"],
                       env=_clean_env(), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout      # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["data"] == "stub" and d["steps"] == 5 and d["scaling"] == "weak"
    assert abs(d["ms_per_step"] * d["steps"] - d["value"] ** -1 * d["steps"] * 64 * 2 * 1e3) < 1e-6 * d["ms_per_step"] * d["steps"] + 1e-9
    ps = d["pf_strong"]
    assert ps["n_ranks"] == 2 and ps["iterations_per_update"] == 2
    assert set(ps["expected_model_DESIGN_section_6"]) == {"10000", "100000", "1000000"}
    for row in ps["sizes"]:
        assert {"one_gpu", "sharded", "speedup_valueN_over_value1"} <= set(row)
        sh = row["sharded"]
        assert len(sh["score_kernel_ms_per_rank"]) == 2 and sh["allgather_ms"] >= 0 and sh["value"] > 0
        assert sh["checksums_equal_across_ranks"]      # one all-gather left the same flat weight vector on both ranks (ragged 1003 too)
        # the peer-store exchange is timed beside the collective, and every form must end on the same estimate
        assert row["sharded_peer"]["value"] > 0 and row["estimates_equal_across_forms"] and "speedup_valueN_over_value1_peer" in row


def test_pf_strong_survives_a_rank_that_cannot_set_the_peer_exchange_up():
    """One rank failing to map the peers' mailboxes (here: the stub raising on rank 1) must not leave the other rank alone in a
    collective: every fallible phase of pf_strong_record ends in an agreement, the row reports the error, the rest of the record and
    the JSON line are produced as usual."""
    env = _clean_env()
    env["MTFHIP_BENCH_STUB_PEER_FAIL"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "1"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    for row in d["pf_strong"]["sizes"]:
        assert "error" in row["sharded_peer"] and "value" not in row["sharded_peer"]
        assert row["sharded"]["value"] > 0 and row["estimates_equal_across_forms"]


def test_bench_single_rank_stub_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--repeats", "2", "--pf-strong", "1"],
                       env=_clean_env(), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["pf_strong"]["n_ranks"] == 1 and "sharded" not in d["pf_strong"]["sizes"][0]


def test_pf_strong_peer_children_row_from_rank_results():
    """the sharded_peer row is assembled from the child processes' results (PfDeviceEngine.peer_row): the slowest rank sets the time,
    a failed rank turns the row into an error that names it"""
    sys.path.insert(0, ROOT)
    import bench

    class Eng:
        rank = 0

        def __init__(self, res):
            self.res = res

        def peer_row(self, n, steps, iters, scratch):
            assert os.path.isdir(scratch)
            return self.res
    ok = {"seconds": 0.5, "score_kernel_ms": 0.01, "scan_select_ms": 0.02, "allgather_ms": 0.0, "peer_exchanges_timed": 30, "estimate": [1.0, 2.0]}
    row = bench.pf_strong_peer_children(Eng(ok), None, 1, 10000, 5, 10)
    assert row["value"] == 10000 * 10 * 5 / 0.5 and row["estimate"] == [1.0, 2.0] and row["estimates_equal_across_ranks"]
    row = bench.pf_strong_peer_children(Eng({"error": "exit code 1: boom"}), None, 1, 10000, 5, 10)
    assert "rank 0" in row["error"] and "boom" in row["error"]


def test_headline_line_survives_a_fault_in_native_code_inside_the_pf_record():
    """r06: rank 0 dying of a signal inside the sharded-filter record (simulated: SIGSEGV raised there, MTFHIP_BENCH_STUB_CRASH=1) -- what a
    fault inside RCCL or a peer mapping would be on the first real multi-GPU node -- still leaves exactly ONE JSON line on stdout: the
    guardian child prints the headline with pf_strong = {"error": ...}.  And when nothing dies the guardian stays silent (the tests above)."""
    env = _clean_env()
    env["MTFHIP_BENCH_STUB_CRASH"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "1"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (p.stdout[-1000:], p.stderr[-1000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "died" in d["pf_strong"]["error"]
