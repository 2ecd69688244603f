import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd._lib import check
from tests import debug_lib
lib = debug_lib.use_for_process()   # dv_debug_* live in libdebvader_hip_debug.so (include/debvader_hip_debug.h)
ctx = E.Context()
o = (C.c_float * 3)()
for blocks in (256, 512, 1024):
    for nacc in (16, 36):
        for rnd in (0, 1):
            for iters in (4000, 40000):
                check(lib.dv_debug_mfma_peak(ctx._h, blocks, iters // (1 if nacc == 16 else 2), nacc, rnd, o))
                print(f"blocks {blocks:5d} nacc {nacc} random {rnd} iters {iters:6d}: {o[0]:6.1f} TF  clock {o[1]:6.0f} MHz  {o[2]:5.1f} cyc/MFMA")
# sustained: ~2 s of back-to-back launches on random operands, clock of the last launch
for rnd in (0, 1):
    for _ in range(200):
        check(lib.dv_debug_mfma_peak(ctx._h, 1024, 40000, 16, rnd, o))
    print(f"sustained random {rnd}: {o[0]:6.1f} TF  clock {o[1]:6.0f} MHz  {o[2]:5.1f} cyc/MFMA")
