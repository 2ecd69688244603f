import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd._lib import lib, check
ctx = E.Context()
B = 256
def w(Hx, Cx, Hy, Cy, sx, pb, code):
    ms = C.c_float()
    check(lib.dv_debug_wgrad(ctx._h, B, Hx, Cx, Hy, Cy, sx, pb, code, 10, C.byref(ms)))
    return ms.value
for name, a in {"convt7 (32,32,s1,64)": (64, 32, 64, 32, 1, 1), "convt5 (64,64,s1,32)": (32, 64, 32, 64, 1, 1), "head (32,12)": (64, 32, 64, 12, 1, 1), "conv1 (32,32,s2)": (59, 32, 30, 32, 2, 1)}.items():
    fl = 2.0 * B * a[2] * a[2] * 9 * a[1] * a[3]
    r = [w(*a, 4 * d) for d in (0, 1, 2, 3)]
    print(f"{name:24s} normal {r[0]*1e3:7.1f} us ({fl/r[0]/1e9:5.1f} TF) | no refill DMA {r[1]*1e3:7.1f} | no MFMA {r[2]*1e3:7.1f} | MFMA only {r[3]*1e3:7.1f}")
