"""Fused stride-2 kernel against the per-class general path at a given batch size (GPU): s2_check.py [NB]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd._lib import check
from tests import debug_lib
lib = debug_lib.use_for_process()   # dv_debug_* live in libdebvader_hip_debug.so (include/debvader_hip_debug.h)
ctx = E.default_context()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
out = (C.c_float * 2)()
for (hs, cs, ht, ct, pb) in ((4, 256, 8, 256, 0), (8, 256, 16, 128, 0), (16, 128, 32, 64, 0), (32, 64, 64, 32, 0),
                             (30, 32, 59, 32, 1), (15, 64, 30, 64, 0), (8, 128, 15, 128, 1), (4, 256, 8, 256, 0)):
    for epi in (0, 2):
        check(lib.dv_debug_gconv_check(ctx._h, NB, hs, cs, ht, ct, 2, pb, 1, 1, epi, out))
        print(f"NB={NB} {hs}x{cs}->{ht}x{ct} pb={pb} epi={epi}: max diff {out[0]:.3e} of {out[1]:.3e}  {'BAD' if out[0] > 2e-5 * out[1] else 'ok'}")
