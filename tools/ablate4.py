import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd._lib import lib, check
ctx = E.Context()
ms = C.c_float()
a = (64, 32, 64, 32, 1, 1)
for flags in (4, 5, 12, 13, 0):
    print("flags", flags, file=sys.stderr); sys.stderr.flush()
    check(lib.dv_debug_wgrad(ctx._h, 256, *a, 4 * flags, 3, C.byref(ms)))
    print("flags", flags, ms.value * 1e3, "us", file=sys.stderr)
