import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd._lib import lib, check
ctx = E.Context()
B = 256
def g(Hs, Cs, Ht, Ct, s, pb, dgrad, nmajor, epi, tile):
    ms = C.c_float()
    check(lib.dv_debug_gconv(ctx._h, B, Hs, Cs, Ht, Ct, s, pb, dgrad, nmajor, epi, 0, tile, 10, C.byref(ms)))
    return ms.value
for name, a in {"conv0 fwd 59x8->32": (59, 8, 59, 32, 1, 1, 0, 0, 2), "head dgrad 64x16->32": (64, 16, 64, 32, 1, 1, 1, 1, 0),
                "head dgrad v1 64x12->32": (64, 12, 64, 32, 1, 1, 1, 1, 0), "head fwd 64x32->16": (64, 32, 64, 16, 1, 1, 0, 0, 1)}.items():
    for rep in range(2):
        v1 = g(*a, 1099); v2 = g(*a, -1)
        print(f"{name:26s} v1 {v1*1e3:7.1f} us   v2 {v2*1e3:7.1f} us")
