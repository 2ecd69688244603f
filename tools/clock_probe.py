"""In-kernel clock of the gather-GEMM main loop (stamped build path), after a warm period of back-to-back launches."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd._lib import check
from tests import debug_lib
lib = debug_lib.use_for_process()   # dv_debug_* live in libdebvader_hip_debug.so (include/debvader_hip_debug.h)
ctx = E.Context()
ms = C.c_float()
cases = {"convt3_fwd_s1": (16, 128, 16, 128, 1, 1, 1, 1, 2), "convt1_fwd_s1": (8, 256, 8, 256, 1, 1, 1, 1, 2), "convt7_fwd_s1": (64, 32, 64, 32, 1, 1, 1, 1, 2)}
for name, a in cases.items():
    check(lib.dv_debug_gconv(ctx._h, 256, *a, 0, -1, 3000, C.byref(ms)))   # ~0.6 s warm
    print(name, "plain", ms.value * 1e3, "us", flush=True)
    check(lib.dv_debug_gconv(ctx._h, 256, *a, 0, 599, 500, C.byref(ms)))
    print(name, "stamped", ms.value * 1e3, "us", flush=True)
