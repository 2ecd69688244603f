"""Runs a few gconv layer shapes once each (for rocprofv3 --pmc passes).  GPU only."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd._lib import check
from tests import debug_lib
lib = debug_lib.use_for_process()   # dv_debug_* live in libdebvader_hip_debug.so (include/debvader_hip_debug.h)
ctx = E.Context()
B = 256
ms = C.c_float()
iters = int(os.environ.get("OL_ITERS", "3"))
# (Hs, Cs, Ht, Ct, s, pb, dgrad, nmajor, epi)
cases = {
    "convt3_fwd_s1": (16, 128, 16, 128, 1, 1, 1, 1, 2),
    "convt2_fwd_s2": (8, 256, 16, 128, 2, 0, 1, 1, 2),
    "conv2_fwd_s1": (30, 32, 30, 64, 1, 1, 0, 0, 2),
    "convt7_fwd_s1": (64, 32, 64, 32, 1, 1, 1, 1, 2),
    "convt6_fwd_s2": (32, 64, 64, 32, 2, 0, 1, 1, 2),
}
for name in os.environ.get("OL_CASES", ",".join(cases)).split(","):
    a = cases[name]
    check(lib.dv_debug_gconv(ctx._h, B, *a, 0, -1, iters, C.byref(ms)))
    print(name, ms.value * 1e3, "us")
