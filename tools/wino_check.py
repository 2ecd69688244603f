"""Winograd F(2x2,3x3) kernel against the general gather-GEMM on random operands (dv_debug_gconv_check), and its speed
against the direct kernels (dv_debug_gconv) for the stride-1 layers of the 59 x 59 x 6 net at batch 256."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd._lib import check
from tests import debug_lib
lib = debug_lib.use_for_process()   # dv_debug_* live in libdebvader_hip_debug.so (include/debvader_hip_debug.h)
ctx = E.default_context()
out = (C.c_float * 2)()
bad = 0
for H in () if (len(sys.argv) > 1 and sys.argv[1] == "benchonly") else (64, 59, 40, 32, 30, 17, 16, 15, 8, 5):
    for (cs, ct) in ((32, 32), (32, 16), (16, 32), (64, 64), (32, 64), (64, 128), (128, 128), (128, 256), (256, 256)):
        if H > 32 and cs * ct > 64 * 64:
            continue
        for dgrad, nmajor in ((0, 0), (1, 1)):
            for epi in (0, 1, 2):
                c = (3, H, cs, H, ct, 1, 1, dgrad, nmajor, epi)
                check(lib.dv_debug_gconv_check(ctx._h, *c, out))
                rel = out[0] / max(out[1], 1e-30)
                flag = "" if rel <= 2e-5 else "   <-- BAD"
                bad += rel > 2e-5
                if flag or epi == 2:
                    print(f"H={H:3d} {cs:3d}->{ct:3d} dgrad={dgrad} epi={epi}: maxdiff {out[0]:.3e} / max {out[1]:.3e} = {rel:.2e}{flag}", flush=True)
# several items per workgroup (the ring runs across item boundaries), ragged block groups
for (NB, H, cs, ct) in () if (len(sys.argv) > 1 and sys.argv[1] == "benchonly") else (
        (37, 30, 32, 64), (21, 64, 32, 32), (150, 8, 128, 256), (150, 8, 256, 256), (61, 16, 128, 128), (45, 15, 64, 128), (29, 59, 32, 48), (3, 10, 96, 96), (40, 10, 96, 32), (40, 10, 96, 96), (5, 13, 48, 96)):
    for dgrad, nmajor in ((0, 0), (1, 1)):
        for epi in (0, 2):
            check(lib.dv_debug_gconv_check(ctx._h, NB, H, cs, H, ct, 1, 1, dgrad, nmajor, epi, out))
            rel = out[0] / max(out[1], 1e-30)
            flag = "" if rel <= 2e-5 else "   <-- BAD"
            bad += rel > 2e-5
            print(f"NB={NB:3d} H={H:3d} {cs:3d}->{ct:3d} dgrad={dgrad} epi={epi}: maxdiff {out[0]:.3e} / max {out[1]:.3e} = {rel:.2e}{flag}", flush=True)
print("bad cases:", bad)
if len(sys.argv) > 1 and sys.argv[1] in ("bench", "benchonly"):
    ms = C.c_float()
    NB = 256
    layers = [("enc conv2 fwd", 30, 32, 30, 64, 0, 0), ("enc conv2 dgrad", 30, 64, 30, 32, 1, 1),
              ("enc conv4 fwd", 15, 64, 15, 128, 0, 0), ("enc conv4 dgrad", 15, 128, 15, 64, 1, 1),
              ("enc conv6 fwd", 8, 128, 8, 256, 0, 0), ("enc conv6 dgrad", 8, 256, 8, 128, 1, 1),
              ("dec convt1 fwd", 8, 256, 8, 256, 1, 1), ("dec convt3 fwd", 16, 128, 16, 128, 1, 1),
              ("dec convt5 fwd", 32, 64, 32, 64, 1, 1), ("dec convt7 fwd", 64, 32, 64, 32, 1, 1),
              ("dec convt7 bwd", 64, 32, 64, 32, 0, 0), ("head fwd", 64, 32, 64, 16, 0, 0), ("head dgrad", 64, 16, 64, 32, 1, 1)]
    for name, hs, cs, ht, ct, dgrad, nmajor in layers:
        res = []
        for wino in (1, 3, 0):                         # four-wave kernel, eight-wave kernel, direct
            check(lib.dv_debug_winograd(wino))
            check(lib.dv_debug_gconv(ctx._h, NB, hs, cs, ht, ct, 1, 1, dgrad, nmajor, 2, 0, -1, 50, C.byref(ms)))
            res.append(ms.value)
        check(lib.dv_debug_winograd(1))
        fl = 2.0 * NB * ht * ht * 9 * cs * ct
        print(f"{name:18s} winograd4 {res[0]*1e3:7.1f} us ({fl/res[0]/1e9:6.1f} TF algorithmic)   winograd8 {res[1]*1e3:7.1f} us ({fl/res[1]/1e9:6.1f} TF)   direct {res[2]*1e3:7.1f} us ({fl/res[2]/1e9:6.1f} TF)", flush=True)

# ---- weight gradient in the Winograd domain ----
if len(sys.argv) > 1 and sys.argv[1] in ("wgrad", "bench"):
    bad = 0
    for H in (32, 30, 17, 16, 15, 8, 5):
        for (cx, cy) in ((64, 64), (64, 128), (128, 64), (128, 128), (128, 256), (256, 256)):
            if H > 16 and cx * cy > 128 * 128:
                continue
            for NB in (3, 7):
                check(lib.dv_debug_wgrad_check(ctx._h, NB, H, cx, cy, out))
                rel = out[0] / max(out[1], 1e-30)
                flag = "" if rel <= 2e-5 else "   <-- BAD"
                bad += rel > 2e-5
                if flag or NB == 3:
                    print(f"wgrad H={H:3d} {cx:3d}x{cy:3d} NB={NB}: maxdiff {out[0]:.3e} / max {out[1]:.3e} = {rel:.2e}{flag}", flush=True)
    print("wgrad bad cases:", bad)
    ms = C.c_float()
    NB = 256
    for name, H, cx, cy in (("enc conv4", 15, 64, 128), ("enc conv6", 8, 128, 256), ("dec convt1", 8, 256, 256),
                            ("dec convt3", 16, 128, 128), ("dec convt5", 32, 64, 64)):
        res = []
        for wino in (1, 0):
            check(lib.dv_debug_winograd(wino))
            check(lib.dv_debug_wgrad(ctx._h, NB, H, cx, H, cy, 1, 1, 0, 30, C.byref(ms)))
            res.append(ms.value)
        check(lib.dv_debug_winograd(1))
        fl = 2.0 * NB * H * H * 9 * cx * cy
        print(f"wgrad {name:12s} winograd {res[0]*1e3:7.1f} us ({fl/res[0]/1e9:6.1f} TF algorithmic)   direct {res[1]*1e3:7.1f} us ({fl/res[1]/1e9:6.1f} TF)", flush=True)
