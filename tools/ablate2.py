import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd._lib import lib, check
ctx = E.Context()
B = 256
def g(Hs, Cs, Ht, Ct, s, pb, dgrad, nmajor, epi, tile, iters=10):
    ms = C.c_float()
    check(lib.dv_debug_gconv(ctx._h, B, Hs, Cs, Ht, Ct, s, pb, dgrad, nmajor, epi, 0, tile, iters, C.byref(ms)))
    return ms.value
for tile in (0, 2):
  for dbg in (0, 1):
    prev = None
    for cs in (64, 128, 256, 512, 1024):
        ms = g(8, cs, 8, 256, 1, 1, 1, 1, 2, tile + 100 * dbg)
        fl = 2.0 * B * 64 * 9 * cs * 256
        extra = ""
        if prev:
            dms = ms - prev[0]; dfl = fl - prev[1]
            extra = f"  slope {dfl/dms/1e9:6.1f} TF"
        print(f"tile{tile} dbg{dbg} 8x8 Cin={cs:4d}->256: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF{extra}")
        prev = (ms, fl)
