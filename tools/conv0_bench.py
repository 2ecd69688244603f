import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
from debvader_amd import engine as E
from debvader_amd._lib import check
from tests import debug_lib
lib = debug_lib.use_for_process()   # dv_debug_* live in libdebvader_hip_debug.so (include/debvader_hip_debug.h)
ctx = E.Context()
def gconv(B, Hs, Cs, Ht, Ct, s, pb, dgrad, nmajor, epi=2, iters=300):
    ms = C.c_float()
    check(lib.dv_debug_gconv(ctx._h, B, Hs, Cs, Ht, Ct, s, pb, dgrad, nmajor, epi, 0, -1, iters, C.byref(ms)))
    return ms.value * 1e3
print("conv0 fwd B=256: %.1f us" % gconv(256, 59, 8, 59, 32, 1, 1, 0, 0))
print("conv0 fwd B=128: %.1f us" % gconv(128, 59, 8, 59, 32, 1, 1, 0, 0))
