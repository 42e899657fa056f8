"""mtf_amd -- MI355X (gfx950) implementation of MTF's Lucas-Kanade inner loop.

Only what the hot path needs: the C-ABI library (csrc/ -> libmtfhip.so, contract in include/mtfhip.h),
its Python mirror of the reference's AppearanceModel / StateSpaceModel interface (api.py), the
search-method loops that drive it (sm.py), candidate sharding over GPUs (dist.py) and synthetic
frames (synth.py).
"""
import os as _os

# Kernel arguments in device memory (see the constructor in csrc/api_core.hip): read by the HIP runtime at its first
# call, so it is set here as well in case the library is loaded after some other module initialised HIP.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import _lib  # noqa: E402
from ._lib import (AM_MI, AM_NCC, AM_SSD, MATH_FAST, MATH_REPLAY, SM_ESM, SM_FCLK, SM_ICLK, SSM_AFFINE, SSM_HOMOGRAPHY,  # noqa: F401
                   FunctionNotImplemented, InvalidArgument, LogicError, MtfHipError)
from .api import (Batch, Context, sm_desc, identity_warp, compose_warps, estimate_warp_from_corners,  # noqa: F401
                  apply_warp_to_pts)

__all__ = ["Batch", "Context", "sm_desc", "_lib"]
