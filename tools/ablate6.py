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
for (H, N, nm, epi) in ((16, 128, 0, 0), (16, 128, 1, 2), (8, 256, 1, 2), (32, 64, 1, 2), (64, 32, 1, 2)):
  for tile in (-1,):
    prev = None
    for cs in (32, 64, 128, 256, 512):
        ms = g(H, cs, H, N, 1, 1, nm, nm, epi, tile)
        fl = 2.0 * B * H * H * 9 * cs * N
        extra = ""
        if prev:
            dms = ms - prev[0]; dfl = fl - prev[1]
            extra = f"  slope {dfl/dms/1e9:6.1f} TF  fixed {(ms - fl/(dfl/dms))*1e3:6.1f} us"
        print(f"H={H} N={N} nmajor={nm} epi={epi} Cin={cs:4d}: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF{extra}")
        prev = (ms, fl)
