import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def frame():
    from mtf_amd import synth
    return synth.make_frame(512, 512)


@pytest.fixture(scope="session")
def frame2(frame):
    """the next frame: `frame` seen through a small known homography about the image centre"""
    import numpy as np
    from mtf_amd import synth
    p_true = synth.random_small_homography(np.random.default_rng(2026)) * 0.5
    return synth.warp_frame(frame, p_true, (256.0, 256.0))


@pytest.fixture(scope="session")
def gpu_ctx():
    import mtf_amd
    if not os.path.exists(mtf_amd._lib.LIB_PATH):
        pytest.fail("libmtfhip.so is missing on a GPU box: run __graft_entry__.build()")
    ctx = mtf_amd.Context(0)
    yield ctx
    ctx.close()


# Measured parity errors (not just pass / fail): tests append dicts to PARITY_RECORD; with MTFHIP_PARITY_RECORD=<path> the session
# writes them out as JSON lines (tools/r03_parity_record.sh copies the file into profiles/).
PARITY_RECORD = []


@pytest.fixture(scope="session")
def parity_record():
    return PARITY_RECORD


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("MTFHIP_PARITY_RECORD")
    if path and PARITY_RECORD:
        import json
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            for r in PARITY_RECORD:
                f.write(json.dumps(r) + "\n")
