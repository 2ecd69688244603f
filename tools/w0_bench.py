"""First-layer weight gradient (Cx=8 strip form): plain vs fused-PReLU-backward, with timing ablations (GPU only).
dbg bits: 1 no refill DMA, 2 no MFMA loop, 16 no transform pass (results wrong with any of them)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd._lib import check
from tests import debug_lib
lib = debug_lib.use_for_process()   # dv_debug_* live in libdebvader_hip_debug.so (include/debvader_hip_debug.h)

B = int(os.environ.get("LB_BATCH", "256"))
ctx = E.Context()


def run(fused, dbg, iters=300):
    ms = C.c_float()
    check(lib.dv_debug_wgrad(ctx._h, B, 59, 8, 59, 32, 1, 1, (0x100 if fused else 0) | (dbg << 2), iters, C.byref(ms)))
    return ms.value * 1e3


for fused in (0, 1):
    for dbg in (0, 1, 2, 3) + ((16, 18) if fused else ()):
        print(f"fused={fused} dbg={dbg:2d}: {run(fused, dbg):7.1f} us (kernel + slab reductions)")
