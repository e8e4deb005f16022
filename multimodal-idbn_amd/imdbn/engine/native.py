"""ctypes binding of the C ABI in ``include/imdbn_engine.h`` (one-to-one; no logic here).

The shared library is built in-tree by ``__graft_entry__.build()`` into
``multimodal-idbn_amd/lib/libimdbn_hip.so``.  There is NO CPU fallback: if the library is
missing or fails to load, :func:`lib` raises and every engine call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

MAX_GROUPS = 4
ABI_VERSION = 4
PARITY_F32, FAST_BF16 = 0, 1
RNG_PHILOX, RNG_REPLAY = 0, 1
DATA_UNKNOWN, DATA_BINARY, DATA_REAL = 0, 1, 2      # imdbn_cd_opts.data_binary / next_binary (IMDBN_DATA_*)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libimdbn_hip.so"))


class RbmDesc(C.Structure):
    _fields_ = [
        ("W", C.c_void_p), ("ldw", C.c_int64),
        ("hid_bias", C.c_void_p), ("vis_bias", C.c_void_p),
        ("W_m", C.c_void_p), ("hb_m", C.c_void_p), ("vb_m", C.c_void_p),
        ("V", C.c_int32), ("H", C.c_int32), ("mode", C.c_int32), ("n_groups", C.c_int32),
        ("group_start", C.c_int32 * MAX_GROUPS), ("group_end", C.c_int32 * MAX_GROUPS),
    ]


class Rng(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("_pad", C.c_int32),
        ("seed", C.c_uint64), ("offset", C.c_uint64), ("row0", C.c_int64),
        ("tape", C.c_void_p), ("tape_len", C.c_int64),
        ("cat_tape", C.c_void_p), ("cat_len", C.c_int64),
        ("tape_used", C.c_int64), ("cat_used", C.c_int64), ("draws_used", C.c_uint64),
        ("dev_offset", C.c_void_p),
    ]


class ChainStep(C.Structure):
    _fields_ = [("T", C.c_float), ("sigma", C.c_float), ("eta", C.c_float),
                ("sample_h", C.c_int32), ("vmode", C.c_int32), ("clamp", C.c_int32)]


class ChainSpec(C.Structure):
    _fields_ = [("v_known", C.c_void_p), ("mask", C.c_void_p), ("ldk", C.c_int64),
                ("init_uniform", C.c_int32), ("n_steps", C.c_int32), ("steps", C.POINTER(ChainStep)),
                ("mu", C.c_void_p), ("ldmu", C.c_int64), ("Dz", C.c_int32), ("_pad", C.c_int32),
                ("out_v", C.c_void_p), ("ldo", C.c_int64)]


class CdOpts(C.Structure):
    _fields_ = [("cd_k", C.c_int32), ("lr", C.c_float), ("momentum", C.c_float), ("weight_decay", C.c_float),
                ("sparsity", C.c_int32), ("sparsity_target", C.c_float),
                ("sample_h", C.c_int32), ("sample_v", C.c_int32), ("reclamp_negative", C.c_int32),
                ("next_data", C.c_void_p), ("ld_next", C.c_int64), ("next_slot", C.c_int32), ("data_slot", C.c_int32),
                ("data_binary", C.c_int32), ("next_binary", C.c_int32),
                ("fwd_out", C.c_void_p), ("ld_fwd", C.c_int64)]


_P = C.c_void_p
_I64 = C.c_int64
_INT = C.c_int
_F = C.c_float
_SZ = C.c_size_t

# name -> (restype, argtypes); must list every symbol the header declares
SIGNATURES = {
    "imdbn_version": (_INT, []),
    "imdbn_last_error": (_INT, [C.c_char_p, _SZ]),
    "imdbn_device_info": (_INT, [C.POINTER(_INT), C.c_char_p, _SZ]),
    "imdbn_ws_bytes": (_SZ, [_INT, _INT, _INT]),
    "imdbn_set_tuning": (_INT, [_INT, _INT]),
    "imdbn_set_option": (_INT, [C.c_char_p, _INT]),
    "imdbn_options_create": (_P, []),
    "imdbn_options_destroy": (None, [_P]),
    "imdbn_options_set": (_INT, [_P, C.c_char_p, _INT]),
    "imdbn_use_options": (_INT, [_P]),
    "imdbn_profile_enable": (_INT, [_INT]),
    "imdbn_profile_read": (_INT, [C.POINTER(C.c_double), C.POINTER(_INT)]),
    "imdbn_debug_stamps": (_INT, [C.POINTER(C.c_longlong), _INT]),
    "imdbn_debug_ws_offset": (_INT, [_INT, _INT, _INT, C.c_char_p, C.POINTER(_SZ)]),
    "imdbn_rng_advance": (_INT, [_P, C.c_uint64, _P]),
    "imdbn_rbm_prop_up": (_INT, [C.POINTER(RbmDesc), _P, _I64, _INT, _F, C.POINTER(Rng), _P, _I64, _P, _I64, _P, _SZ, _P]),
    "imdbn_rbm_forward": (_INT, [C.POINTER(RbmDesc), _P, _I64, _INT, _INT, _P, _I64, _P, _SZ, _P]),
    "imdbn_rbm_free_energy": (_INT, [C.POINTER(RbmDesc), _P, _I64, _INT, _P, _P, _SZ, _P]),
    "imdbn_rbm_prop_down": (_INT, [C.POINTER(RbmDesc), _P, _I64, _INT, _F, _INT, _P, _I64, _P, _SZ, _P]),
    "imdbn_rbm_sample_visible": (_INT, [C.POINTER(RbmDesc), _P, _I64, _INT, C.POINTER(Rng), _P, _I64, _P]),
    "imdbn_rbm_gibbs_step": (_INT, [C.POINTER(RbmDesc), _P, _I64, _INT, _INT, _INT, C.POINTER(Rng), _P, _P, _P, _P, _P, _SZ, _P]),
    "imdbn_rbm_cd_step": (_INT, [C.POINTER(RbmDesc), _P, _I64, _INT, C.POINTER(CdOpts), C.POINTER(Rng), _P, _P, _SZ, _P]),
    "imdbn_packed_delta_floats": (_SZ, [_INT, _INT]),
    "imdbn_rbm_prefetch_ok": (_INT, [C.POINTER(RbmDesc), _INT]),
    "imdbn_factor_compact_bytes": (_INT, [_INT, _INT, _INT, _INT, C.POINTER(_SZ)]),
    "imdbn_rbm_pack_factors": (_INT, [_INT, _INT, _INT, _INT, _P, _P, _P]),
    "imdbn_rbm_unpack_factors": (_INT, [_INT, _INT, _INT, _INT, _P, _SZ, _INT, _P, _SZ, _INT, _P]),
    "imdbn_rbm_cd_factors_wire": (_INT, [C.POINTER(RbmDesc), _P, _I64, _INT, C.POINTER(CdOpts), C.POINTER(Rng), _INT, _P, _P, _SZ, _P]),
    "imdbn_rbm_apply_wire": (_INT, [C.POINTER(RbmDesc), _P, _SZ, _INT, _INT, _INT, _INT, _P, _SZ, C.POINTER(CdOpts), _P, _P]),
    "imdbn_rbm_apply_factors_wire": (_INT, [C.POINTER(RbmDesc), _P, _SZ, _P, _SZ, _INT, _INT, _INT, C.POINTER(CdOpts), _P, _P]),
    "imdbn_rbm_cd_stats": (_INT, [C.POINTER(RbmDesc), _P, _I64, _INT, C.POINTER(CdOpts), C.POINTER(Rng), _P, _P, _SZ, _P]),
    "imdbn_rbm_apply_delta": (_INT, [C.POINTER(RbmDesc), _P, _INT, C.POINTER(CdOpts), _P, _P]),
    "imdbn_factor_block": (_INT, [_INT, _INT, _INT, C.POINTER(_SZ), C.POINTER(_SZ)]),
    "imdbn_rbm_cd_factors": (_INT, [C.POINTER(RbmDesc), _P, _I64, _INT, C.POINTER(CdOpts), C.POINTER(Rng), _P, _SZ, _P]),
    "imdbn_rbm_apply_factors": (_INT, [C.POINTER(RbmDesc), _P, _INT, _SZ, _INT, _INT, C.POINTER(CdOpts), _P, _P]),
    "imdbn_rbm_chain": (_INT, [C.POINTER(RbmDesc), _P, _P, _I64, _INT, _INT, _INT, C.POINTER(ChainStep), _P, _I64, _INT,
                               C.POINTER(Rng), _P, _I64, _P, _SZ, _P]),
    "imdbn_rbm_chain_pair": (_INT, [C.POINTER(RbmDesc), _INT, C.POINTER(ChainSpec), C.POINTER(ChainSpec), C.POINTER(Rng), _P, _SZ, _P]),
    "imdbn_rbm_clamped_step": (_INT, [C.POINTER(RbmDesc), _P, _P, _I64, _INT, _INT, C.POINTER(ChainStep), _P, _I64, _INT,
                                      C.POINTER(CdOpts), C.POINTER(Rng), _P, _P, _SZ, _P]),
    "imdbn_rbm_assoc_update": (_INT, [C.POINTER(RbmDesc), _P, _I64, _P, _I64, _P, _I64, _P, _I64, _INT, C.POINTER(CdOpts), _P, _SZ, _P]),
    "imdbn_comm_unique_id": (_INT, [_P]),
    "imdbn_comm_init": (_INT, [C.POINTER(_P), _INT, _INT, _P]),
    "imdbn_comm_destroy": (_INT, [_P]),
    "imdbn_allreduce_sum_f32": (_INT, [_P, _P, _SZ, _P]),
    "imdbn_allgather_bytes": (_INT, [_P, _P, _P, _SZ, _P]),
    "imdbn_rbm_clamped_stats": (_INT, [C.POINTER(RbmDesc), _P, _P, _I64, _INT, _INT, C.POINTER(ChainStep), _P, _I64, _INT,
                                      C.POINTER(CdOpts), C.POINTER(Rng), _P, _P, _SZ, _P]),
}

_lib = None


class EngineError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the native library; raise if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(
                f"native HIP engine not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` at the repo root. There is no CPU fallback for this path.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        buf = C.create_string_buffer(512)
        lib().imdbn_last_error(buf, 512)
        raise EngineError(f"{what} failed (rc={rc}): {buf.value.decode(errors='replace')}")
