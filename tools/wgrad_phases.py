import ctypes as C, sys, os
sys.path.insert(0, "/root/repo")
from debvader_amd import engine as E
from debvader_amd._lib import check
from tests import debug_lib
lib = debug_lib.use_for_process()   # dv_debug_* live in libdebvader_hip_debug.so (include/debvader_hip_debug.h)
ctx = E.Context()
ms = C.c_float()
for name, a in {"convt7 wgrad (32,32 s1 64x64)": (64, 32, 64, 32, 1, 1), "convt5 wgrad (64,64 s1 32x32)": (32, 64, 32, 64, 1, 1),
                "head wgrad (32,16)": (64, 32, 64, 16, 1, 1), "conv2 wgrad (32,64 s1 30x30)": (30, 32, 30, 64, 1, 1)}.items():
    for dbg in (0, 1, 2, 4):
        check(lib.dv_debug_wgrad(ctx._h, 256, *a, dbg << 2, 200, C.byref(ms)))
        print(f"{name} dbg={dbg}: {ms.value*1e3:.1f} us", flush=True)
