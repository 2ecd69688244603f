"""ctypes binding of libdebvader_hip.so (include/debvader_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If it is missing or fails to load,
importing this module raises with the build command to run.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DEBVADER_AMD_LIB selects another build of the same library (the host-side AddressSanitizer build, `make asan`)
LIB_PATH = os.environ.get("DEBVADER_AMD_LIB") or os.path.join(_HERE, "lib", "libdebvader_hip.so")

DV_MAX_LEVELS = 8
DV_UNIQUE_ID_BYTES = 128
DV_N_SCALARS = 4
SCALAR_NAMES = ("loss", "nll_mean", "kl_reg", "mse")


class DvConfig(C.Structure):
    _fields_ = [
        ("height", C.c_int32), ("width", C.c_int32), ("bands", C.c_int32),
        ("latent_dim", C.c_int32), ("n_levels", C.c_int32),
        ("filters", C.c_int32 * DV_MAX_LEVELS), ("kernels", C.c_int32 * DV_MAX_LEVELS),
        ("max_batch", C.c_int32),
        ("kl_weight", C.c_float), ("kl_multiplicity", C.c_int32),
        ("bn_eps", C.c_float), ("bn_momentum", C.c_float), ("bn_moving_var_unbiased", C.c_int32),
        ("sigma_floor", C.c_float), ("diag_shift", C.c_float),
        ("dtype", C.c_int32),
        ("infer_graph", C.c_int32),
    ]


DV_DTYPE_F32, DV_DTYPE_BF16 = 0, 1


class DvError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"libdebvader_hip status {status}: {msg}")
        self.status = status


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP engine first "
            "(python -c 'import __graft_entry__ as g; g.build()'  or  make -C debvader_amd/csrc). "
            "debvader_amd has no CPU fallback.")
    try:
        return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise ImportError(f"could not load {LIB_PATH}: {e}") from e


lib = _load()

_p = C.c_void_p
_f = C.POINTER(C.c_float)
_d = C.POINTER(C.c_double)
_i32 = C.POINTER(C.c_int32)
_i64 = C.POINTER(C.c_int64)

# name -> (restype, argtypes); every symbol declared in include/debvader_hip.h
# dv_chunk_fn of dv_infer_cutouts_stream: int (*)(void* user, int64 first, int32 count, const float* mean, const float* stddev)
CHUNK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float))

SIGNATURES = {
    "dv_version": (C.c_int, []),
    "dv_build_kind": (C.c_int, []),
    "dv_crc32c": (C.c_uint32, [C.c_uint32, C.c_void_p, C.c_size_t]),
    "dv_last_error": (C.c_int, [C.c_char_p, C.c_size_t]),
    "dv_config_default": (C.c_int, [C.POINTER(DvConfig)]),
    "dv_arch_counts": (C.c_int, [C.POINTER(DvConfig), _i32, _i64, _i64, _i64]),
    "dv_arch_describe": (C.c_int, [C.POINTER(DvConfig), C.c_int32, C.c_char_p, C.c_size_t, _i64, _i32, _i32]),
    "dv_arch_macs": (C.c_int, [C.POINTER(DvConfig), _i64, _i64]),
    "dv_arch_buckets": (C.c_int, [C.POINTER(DvConfig), C.POINTER(C.c_int64)]),
    "dv_arch_offset": (C.c_int, [C.POINTER(DvConfig), C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dv_device_count": (C.c_int, [_i32]),
    "dv_device_bus_id": (C.c_int, [C.c_int32, C.c_char_p, C.c_size_t]),
    "dv_comm_unique_id": (C.c_int, [_p]),
    "dv_ctx_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _p, C.POINTER(_p)]),
    "dv_ctx_destroy": (C.c_int, [_p]),
    "dv_ctx_sync": (C.c_int, [_p]),
    "dv_ctx_allreduce_host": (C.c_int, [_p, _f, C.c_int32]),
    "dv_ctx_comm_info": (C.c_int, [_p, _i32, _i32, _i32, C.c_char_p, C.c_size_t, _i32]),
    "dv_comm_prof_enable": (C.c_int, [_p, C.c_int32]),
    "dv_comm_prof_read": (C.c_int, [_p, _i64, C.POINTER(C.c_double), _i64, C.POINTER(C.c_double)]),
    "dv_model_create": (C.c_int, [_p, C.POINTER(DvConfig), C.POINTER(_p)]),
    "dv_model_destroy": (C.c_int, [_p]),
    "dv_model_init": (C.c_int, [_p, C.c_uint64]),
    "dv_model_get_param": (C.c_int, [_p, C.c_int32, _f, C.c_size_t]),
    "dv_model_set_param": (C.c_int, [_p, C.c_int32, _f, C.c_size_t]),
    "dv_model_get_grad": (C.c_int, [_p, C.c_int32, _f, C.c_size_t]),
    "dv_model_get_slot": (C.c_int, [_p, C.c_int32, C.c_int32, _f, C.c_size_t]),
    "dv_model_set_slot": (C.c_int, [_p, C.c_int32, C.c_int32, _f, C.c_size_t]),
    "dv_model_set_trainable": (C.c_int, [_p, C.c_int32, C.c_int32]),
    "dv_optimizer_reset": (C.c_int, [_p, C.c_float, C.c_float, C.c_float, C.c_float]),
    "dv_optimizer_get_iter": (C.c_int, [_p, _i64]),
    "dv_optimizer_set_iter": (C.c_int, [_p, C.c_int64]),
    "dv_data_upload": (C.c_int, [_p, C.c_int32, _f, _f, C.c_int64]),
    "dv_data_free": (C.c_int, [_p, C.c_int32]),
    "dv_train_step": (C.c_int, [_p, C.c_int32, _i32, C.c_int64, C.c_int32, C.c_int32, _f, C.c_uint64, _f]),
    "dv_eval_step": (C.c_int, [_p, C.c_int32, _i32, C.c_int64, C.c_int32, C.c_int32, _f, C.c_uint64, _f]),
    "dv_grad_step": (C.c_int, [_p, C.c_int32, _i32, C.c_int64, C.c_int32, C.c_int32, _f, C.c_uint64, _f]),
    "dv_train_step_async": (C.c_int, [_p, C.c_int32, C.POINTER(C.c_int32), C.c_int64, C.c_int32, C.c_int32, C.c_uint64, C.c_int32]),
    "dv_step_result": (C.c_int, [_p, C.c_int32, _f]),
    "dv_train_steps": (C.c_int, [_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, _f]),
    "dv_model_set_normalise": (C.c_int, [_p, C.c_int32]),
    "dv_model_set_mse_sample": (C.c_int, [_p, C.c_int32]),
    "dv_model_set_keep_outputs": (C.c_int, [_p, C.c_int32]),
    "dv_infer": (C.c_int, [_p, _f, C.c_int64, _f, C.c_uint64, _f, _f, _f, _f, _f]),
    "dv_infer_f64": (C.c_int, [_p, _d, C.c_int64, _f, C.c_uint64, _f, _f, _f, _f, _f]),
    "dv_infer_cutouts": (C.c_int, [_p, _d, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int64, C.c_uint64, _f, _f, _f, _f, _f]),
    "dv_infer_cutouts_keep": (C.c_int, [_p, _d, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int64, C.c_uint64, _f, _f, _d]),
    "dv_infer_cutouts_stream": (C.c_int, [_p, _d, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int64, C.c_uint64,
                                          C.c_void_p, C.c_void_p]),
    "dv_infer_cutouts_composite": (C.c_int, [_p, _d, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int64,
                                             C.c_uint64, _d, _d, _d, _d]),
    "dv_scene_extract": (C.c_int, [_p, _d, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, _d]),
    "dv_scene_composite": (C.c_int, [_p, _d, C.c_int32, C.c_int32, _d, _d, C.c_int32, C.c_int32, C.c_double]),
    "dv_infer_mc": (C.c_int, [_p, _f, C.c_int64, C.c_int32, C.c_uint64, _f, _f]),
    "dv_encode": (C.c_int, [_p, _f, C.c_int64, _f]),
    "dv_decode": (C.c_int, [_p, _f, C.c_int64, _f, _f]),
    "dv_model_get_activation": (C.c_int, [_p, C.c_char_p, _f, C.c_size_t]),
    "dv_prof_enable": (C.c_int, [_p, C.c_int32]),
    "dv_prof_read": (C.c_int, [_p, C.c_int32, _i64, C.POINTER(C.c_double)]),
    "dv_prof_reset": (C.c_int, [_p]),
    "dv_prof_read_family": (C.c_int, [_p, C.c_int32, C.c_char_p, C.c_size_t, _i64, C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

def bind(handle, signatures):
    """restype / argtypes of every listed symbol on a loaded library (raises AttributeError for a missing export)."""
    for _name, (_res, _args) in signatures.items():
        _fn = getattr(handle, _name)
        _fn.restype = _res
        _fn.argtypes = _args
    return handle


bind(lib, SIGNATURES)
# dv_build_kind() == 1: a DEVELOPMENT build of the engine was selected with DEBVADER_AMD_LIB (it carries measurement
# switches that give wrong results and the one-GPU rehearsal hooks).  The package honours DV_DEBUG_SAME_GPU /
# DV_DEBUG_FAKE_PEERS only then; with the product library those variables do nothing.
IS_DEBUG_LIB = lib.dv_build_kind() == 1


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib.dv_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(status: int):
    if status != 0:
        raise DvError(status, last_error())
