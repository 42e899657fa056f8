"""CPU: the C-ABI library loads without a GPU and exports every symbol include/mtfhip.h declares;
argument validation that needs no device works; there is no CPU fallback behind the API."""
import ctypes
import os
import re

import pytest

import mtf_amd
from mtf_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    L.build()
    return ctypes.CDLL(L.LIB_PATH)


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mtfhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mtfhip_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    syms = header_symbols()
    assert len(syms) >= 50
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(L.SYMBOLS) == syms, set(L.SYMBOLS) ^ set(syms)


def test_header_cites_reference_lines():
    text = open(os.path.join(ROOT, "include", "mtfhip.h")).read()
    cites = re.findall(r"[A-Za-z]+\.(?:cc|h):\d+", text)
    assert len(cites) >= 40


def test_no_device_behaviour(lib):
    """Without a GPU every device entry point fails loudly (no silent CPU path)."""
    lib.mtfhip_last_error.restype = ctypes.c_char_p
    if lib.mtfhip_device_count() > 0:
        pytest.skip("a GPU is visible")
    h = ctypes.c_void_p()
    rc = lib.mtfhip_ctx_create(0, None, ctypes.byref(h))
    assert rc == -5
    assert b"no HIP device" in lib.mtfhip_last_error()
    with pytest.raises(mtf_amd.MtfHipError):
        mtf_amd.Context(0)


def test_product_does_not_reach_into_the_oracle():
    """Nothing under mtf_amd/ (nor bench.py's timed path) imports, links or executes oracle/."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "mtf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".cc")) or f == "Makefile":
                s = open(os.path.join(dp, f)).read()
                if re.search(r"oracle_py|mtf_oracle|numpy_ref|libmtf_oracle", s):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
