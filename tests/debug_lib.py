"""The DEVELOPMENT build of the engine, libdebvader_hip_debug.so (TEST / TOOL INFRASTRUCTURE, not product).

The product library (debvader_amd/lib/libdebvader_hip.so) exports exactly what include/debvader_hip.h declares.  The
kernel micro-benchmarks, HIP-vs-HIP cross-checks and process-wide kernel-family switches of include/debvader_hip_debug.h
live in a second library built from the same objects plus engine.hip compiled with -DDV_DEBUG_EXPORTS
(debvader_amd/csrc/Makefile).  Tests and tools that need them load it HERE, explicitly:

    with debug_lib.debug_build() as dlib:          # debvader_amd.engine drives the debug build inside the block
        dlib.dv_debug_winograd(0)
        eng = E.Engine(cfg)                         # an engine of the debug build

Inside the block `debvader_amd.engine.lib` / `debvader_amd._lib.lib` point at the debug build and the default context is
one of its own; contexts created inside are closed on exit and the product handles restored.  The two libraries are two
copies of the engine in one process: a handle of one must never reach the other - hence the swap instead of mixing calls.
"""
import contextlib
import ctypes as C
import os

from debvader_amd import _lib
from debvader_amd import engine as E

DEBUG_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "lib", "libdebvader_hip_debug.so")

_p, _f = C.c_void_p, C.POINTER(C.c_float)
DEBUG_SIGNATURES = {
    "dv_debug_gconv": (C.c_int, [_p] + [C.c_int32] * 13 + [_f]),
    "dv_debug_gconv_check": (C.c_int, [_p] + [C.c_int32] * 10 + [_f]),
    "dv_debug_mfma_peak": (C.c_int, [_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f]),
    "dv_debug_general_kernels": (C.c_int, [C.c_int32]),
    "dv_debug_winograd": (C.c_int, [C.c_int32]),
    "dv_debug_fuse_prelu_bwd": (C.c_int, [C.c_int32]),
    "dv_debug_wgrad_check": (C.c_int, [_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f]),
    "dv_debug_wgrad": (C.c_int, [_p] + [C.c_int32] * 9 + [_f]),
}

_dlib = None


def load():
    """The debug build as a ctypes handle with every product and debug signature bound."""
    global _dlib
    if _dlib is None:
        if not os.path.exists(DEBUG_LIB_PATH):
            raise ImportError(f"{DEBUG_LIB_PATH} not found: make -C debvader_amd/csrc builds it beside the product library")
        h = C.CDLL(DEBUG_LIB_PATH, mode=C.RTLD_LOCAL)
        _lib.bind(h, _lib.SIGNATURES)
        _lib.bind(h, DEBUG_SIGNATURES)
        _dlib = h
    return _dlib


@contextlib.contextmanager
def debug_build():
    dlib = load()
    saved = (_lib.lib, E.lib, E._default_ctx, set(E.Context._live))
    _lib.lib = E.lib = dlib
    E._default_ctx = None
    try:
        yield dlib
    finally:
        for ctx in list(E.Context._live):
            if ctx not in saved[3]:
                ctx.close()                      # contexts (and their engines) of the debug build die with the block
        _lib.lib, E.lib, E._default_ctx = saved[0], saved[1], saved[2]


def use_for_process():
    """Tools: make the debug build THE library of this process (call before anything creates a context)."""
    dlib = load()
    _lib.lib = E.lib = dlib
    return dlib
