import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd._lib import lib, check
ctx = E.Context()
ms = C.c_float()
cases = {"convt3 dgrad 16x128 kmajor (128x128)": (16, 128, 16, 128, 1, 1, 0, 0, 0), "convt1 fwd 8x256 nmajor": (8, 256, 8, 256, 1, 1, 1, 1, 2),
         "convt5 fwd 32x64 nmajor": (32, 64, 32, 64, 1, 1, 1, 1, 2), "convt7 fwd 64x32": (64, 32, 64, 32, 1, 1, 1, 1, 2)}
for name, a in cases.items():
    print(name, file=sys.stderr); sys.stderr.flush()
    check(lib.dv_debug_gconv(ctx._h, 256, *a, 0, 599, 3, C.byref(ms)))
    fl = 2.0 * 256 * a[2] * a[2] * 9 * a[1] * a[3]
    print(f"   {ms.value*1e3:.1f} us  {fl/ms.value/1e9:.1f} TF (with stamps)", file=sys.stderr)
