"""Per-layer micro-benchmark of the gather-GEMM / wgrad kernels at the BASELINE batch (tuning aid, GPU only)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from debvader_amd import engine as E
from debvader_amd._lib import check
from tests import debug_lib
lib = debug_lib.use_for_process()   # dv_debug_* live in libdebvader_hip_debug.so (include/debvader_hip_debug.h)

B = int(os.environ.get("LB_BATCH", "256"))
ITERS = int(os.environ.get("LB_ITERS", "300"))   # long enough that clocks have ramped (10 iterations read ~12 % slow)
ctx = E.Context()


def gconv(Hs, Cs, Ht, Ct, s, pb, dgrad, nmajor, epi=2, single=0, tile=-1, iters=None):
    iters = iters or ITERS
    ms = C.c_float()
    check(lib.dv_debug_gconv(ctx._h, B, Hs, Cs, Ht, Ct, s, pb, dgrad, nmajor, epi, single, tile, iters, C.byref(ms)))
    return ms.value


def wgrad(Hx, Cx, Hy, Cy, sx, pb, single=0, iters=None):
    iters = iters or ITERS
    ms = C.c_float()
    check(lib.dv_debug_wgrad(ctx._h, B, Hx, Cx, Hy, Cy, sx, pb, single, iters, C.byref(ms)))
    return ms.value


def flops_conv(Hout, Cin, Cout, taps=9):
    return 2.0 * B * Hout * Hout * taps * Cin * Cout


# (name, kind, args, flops)
enc = [(59, 8, 59, 32, 1, 1), (59, 32, 30, 32, 2, 1), (30, 32, 30, 64, 1, 1), (30, 64, 15, 64, 2, 0),
       (15, 64, 15, 128, 1, 1), (15, 128, 8, 128, 2, 1), (8, 128, 8, 256, 1, 1), (8, 256, 4, 256, 2, 0)]
dec = [(4, 256, 8, 256, 2, 0), (8, 256, 8, 256, 1, 1), (8, 256, 16, 128, 2, 0), (16, 128, 16, 128, 1, 1),
       (16, 128, 32, 64, 2, 0), (32, 64, 32, 64, 1, 1), (32, 64, 64, 32, 2, 0), (64, 32, 64, 32, 1, 1)]

tiles = [int(t) for t in os.environ.get("LB_TILES", "-1").split(",")]
which = os.environ.get("LB_WHICH", "all")
rows = []
for tile in tiles:
    tot_ms = tot_fl = 0
    print(f"== tile override {tile}")
    if which in ("all", "gconv"):
        for i, (hi, ci, ho, co, s, pb) in enumerate(enc):
            fl = flops_conv(ho, ci if i else 6, co)
            ms = gconv(hi, ci, ho, co, s, pb, 0, 0, tile=tile)
            print(f"enc conv{i} fwd   {hi:3d}x{ci:3d}->{ho:3d}x{co:3d} s{s}: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF")
            tot_ms += ms; tot_fl += fl
            ms = gconv(ho, co, hi, ci, s, pb, 1, 1, epi=0, tile=tile)
            print(f"enc conv{i} dgrad {ho:3d}x{co:3d}->{hi:3d}x{ci:3d} s{s}: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF")
            tot_ms += ms; tot_fl += fl
        for i, (hi, ci, ho, co, s, pb) in enumerate(dec):
            fl = 2.0 * B * hi * hi * 9 * ci * co
            ms = gconv(hi, ci, ho, co, s, pb, 1, 1, tile=tile)
            print(f"dec convt{i} fwd   {hi:3d}x{ci:3d}->{ho:3d}x{co:3d} s{s}: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF")
            tot_ms += ms; tot_fl += fl
            ms = gconv(ho, co, hi, ci, s, pb, 0, 0, epi=0, tile=tile)
            print(f"dec convt{i} dgrad {ho:3d}x{co:3d}->{hi:3d}x{ci:3d} s{s}: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF")
            tot_ms += ms; tot_fl += fl
        fl = flops_conv(64, 32, 12)
        ms = gconv(64, 32, 64, 16, 1, 1, 0, 0, epi=1, tile=tile); print(f"head fwd: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF"); tot_ms += ms; tot_fl += fl
        ms = gconv(64, 16, 64, 32, 1, 1, 1, 1, epi=0, tile=tile); print(f"head dgrad: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF"); tot_ms += ms; tot_fl += fl
        for (k, n) in ((4096, 560), (32, 560), (560, 4096)):
            fl = 2.0 * B * k * n
            ms = gconv(1, k, 1, n, 1, 0, 0, 0, epi=1, single=1, tile=tile); print(f"dense {k}->{n} fwd: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF"); tot_ms += ms; tot_fl += fl
            ms = gconv(1, n, 1, k, 1, 0, 0, 1, epi=0, single=1, tile=tile); print(f"dense {k}->{n} dgrad: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF"); tot_ms += ms; tot_fl += fl
        print(f"gconv total {tot_ms:.3f} ms  {tot_fl/tot_ms/1e9:.1f} TF")
    if which in ("all", "wgrad") and tile == tiles[0]:
        tw = tf = 0
        for i, (hi, ci, ho, co, s, pb) in enumerate(enc):
            fl = flops_conv(ho, ci if i else 6, co)
            ms = wgrad(hi, ci, ho, co, s, pb); ms1 = wgrad(hi, ci, ho, co, s, pb, single=2); print(f"enc conv{i} wgrad: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF   (v1 {ms1*1e3:8.1f} us)"); tw += ms; tf += fl
        for i, (hi, ci, ho, co, s, pb) in enumerate(dec):
            fl = 2.0 * B * hi * hi * 9 * ci * co
            ms = wgrad(ho, co, hi, ci, s, pb); ms1 = wgrad(ho, co, hi, ci, s, pb, single=2); print(f"dec convt{i} wgrad: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF   (v1 {ms1*1e3:8.1f} us)"); tw += ms; tf += fl
        fl = flops_conv(64, 32, 12)
        ms = wgrad(64, 32, 64, 12, 1, 1); ms1 = wgrad(64, 32, 64, 12, 1, 1, single=2); print(f"head wgrad: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF   (v1 {ms1*1e3:8.1f} us)"); tw += ms; tf += fl
        for (k, n) in ((4096, 560), (32, 560), (560, 4096)):
            fl = 2.0 * B * k * n
            ms = wgrad(1, k, 1, n, 1, 0, single=1); print(f"dense {k}->{n} wgrad: {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF"); tw += ms; tf += fl
        print(f"wgrad total {tw:.3f} ms  {tf/tw/1e9:.1f} TF")
