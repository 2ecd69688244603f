"""Per-workgroup timeline of one gather-GEMM launch (debug stamps).  GPU only."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd._lib import check
from tests import debug_lib
lib = debug_lib.use_for_process()   # dv_debug_* live in libdebvader_hip_debug.so (include/debvader_hip_debug.h)
ctx = E.Context()
ms = C.c_float()
cases = {"convt3_fwd_s1": (16, 128, 16, 128, 1, 1, 1, 1, 2), "convt7_fwd_s1": (64, 32, 64, 32, 1, 1, 1, 1, 2),
         "conv4_fwd_s1": (15, 64, 15, 128, 1, 1, 0, 0, 2), "convt5_fwd_s1": (32, 64, 32, 64, 1, 1, 1, 1, 2),
         "convt6_fwd_s2": (32, 64, 64, 32, 2, 0, 1, 1, 2), "convt2_fwd_s2": (8, 256, 16, 128, 2, 0, 1, 1, 2),
         "conv3_dgrad_s2": (15, 64, 30, 64, 2, 0, 1, 1, 0)}
for name, a in cases.items():
    NBATCH = int(os.environ.get("TL_BATCH", "256"))
    check(lib.dv_debug_gconv(ctx._h, NBATCH, *a, 0, -1, 2000, C.byref(ms)))
    print(name, "plain", ms.value * 1e3, "us", flush=True)
    check(lib.dv_debug_gconv(ctx._h, NBATCH, *a, 0, int(os.environ.get("TL_CODE", "7099")), 200, C.byref(ms)))
    print(name, "with stamps", ms.value * 1e3, "us", flush=True)
