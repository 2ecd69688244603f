import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd._lib import lib, check
ctx = E.Context()
B = 256
t = C.c_float()
for blocks in (256, 512, 1024, 2048):
    check(lib.dv_debug_mfma_peak(ctx._h, blocks, 4000, C.byref(t)))
    print(f"mfma peak probe blocks={blocks}: {t.value:.1f} TF")
def g(Hs, Cs, Ht, Ct, s, pb, dgrad, nmajor, epi, tile):
    ms = C.c_float()
    check(lib.dv_debug_gconv(ctx._h, B, Hs, Cs, Ht, Ct, s, pb, dgrad, nmajor, epi, 0, tile, 10, C.byref(ms)))
    return ms.value
cases = {"convt1 fwd 8x256->8x256": (8, 256, 8, 256, 1, 1, 1, 1, 2), "convt3 dgrad 16x128 kmajor": (16, 128, 16, 128, 1, 1, 0, 0, 0),
         "convt7 fwd 64x32": (64, 32, 64, 32, 1, 1, 1, 1, 2), "convt5 fwd 32x64": (32, 64, 32, 64, 1, 1, 1, 1, 2)}
for name, a in cases.items():
    fl = 2.0 * B * a[2] * a[2] * 9 * a[1] * a[3]
    for tile in (0, 1, 2, 3):
        row = []
        for dbg in (0, 1, 2, 3):
            ms = g(*a, tile + 100 * dbg)
            row.append(f"{ms*1e3:7.1f}us/{fl/ms/1e9:5.1f}TF")
        print(f"{name:28s} tile{tile} normal|mfma+ldsread|+ldsstore+barrier|gload+mfma: " + " ".join(row))
