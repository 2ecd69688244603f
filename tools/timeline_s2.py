import ctypes as C, sys, os
sys.path.insert(0, "/root/repo")
from debvader_amd import engine as E
from debvader_amd._lib import check
from tests import debug_lib
lib = debug_lib.use_for_process()   # dv_debug_* live in libdebvader_hip_debug.so (include/debvader_hip_debug.h)
ctx = E.Context()
ms = C.c_float()
for name, a in {"convt6_fwd_s2": (32, 64, 64, 32, 2, 0, 1, 1, 2), "conv1_dgrad_s2": (30, 32, 59, 32, 2, 1, 1, 1, 0)}.items():
    check(lib.dv_debug_gconv(ctx._h, 256, *a, 0, -1, 1000, C.byref(ms)))
    print(name, "plain", ms.value * 1e3, "us", flush=True)
    check(lib.dv_debug_gconv(ctx._h, 256, *a, 0, 7099, 100, C.byref(ms)))
    print(name, "with stamps", ms.value * 1e3, "us", flush=True)
