"""ctypes loader for lcpc_amd/lib/liblcpc_hip.so (the product: HIP kernels + C ABI of include/lcpc_hip.h).

There is deliberately NO fallback: if the shared library is missing or no HIP device is usable, every
entry point raises.  Nothing in this package touches the CPU restatement the tests check it against."""
import ctypes as C
import os
import subprocess

ABI_VERSION = 5
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblcpc_hip.so")
CSRC = os.path.join(_HERE, "csrc")


class LcpcParams(C.Structure):
    _fields_ = [("field", C.c_uint32), ("encoding", C.c_uint32), ("hash", C.c_uint32),
                ("rho_num", C.c_uint32), ("rho_den", C.c_uint32), ("sdig_code", C.c_uint32),
                ("seed", C.c_uint64), ("n_coeffs", C.c_uint64), ("n_per_row", C.c_uint64), ("n_cols", C.c_uint64),
                ("device", C.c_int32), ("shard_rank", C.c_uint32), ("shard_count", C.c_uint32)]


class LcpcTimings(C.Structure):
    _fields_ = [("encode_ms", C.c_float), ("hash_ms", C.c_float), ("merkle_ms", C.c_float), ("total_ms", C.c_float),
                ("encode_launches", C.c_uint32), ("hash_launches", C.c_uint32), ("merkle_launches", C.c_uint32),
                ("exchange_exposed_ms", C.c_float), ("staged_slices", C.c_uint32), ("exchange_wire_ms", C.c_float)]


# every symbol include/lcpc_hip.h declares: name -> (restype, argtypes)
_vp, _u64, _u32, _i32, _sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_size_t
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64)      # lcpc_allgather_fn (include/lcpc_hip.h)
WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)   # lcpc_write_fn / lcpc_read_fn

SYMBOLS = {
    "lcpc_abi_version": (_i32, []),
    "lcpc_ctx_create": (_i32, [C.POINTER(LcpcParams), C.POINTER(_vp)]),
    "lcpc_ctx_destroy": (None, [_vp]),
    "lcpc_strerror": (C.c_char_p, [_i32]),
    "lcpc_last_error": (C.c_char_p, [_vp]),
    "lcpc_get_dims": (_i32, [_vp, _u64, _vp, _vp, _vp]),
    "lcpc_dims_ok": (_i32, [_vp, _u64, _u64]),
    "lcpc_get_n_col_opens": (_u64, [_vp]),
    "lcpc_get_n_degree_tests": (_u64, [_vp]),
    "lcpc_field_limbs": (_u32, [_vp]),
    "lcpc_static_get_dims": (_i32, [C.POINTER(LcpcParams), _vp, _vp, _vp]),
    "lcpc_static_get_dims_ml": (_i32, [_vp, _u32, _vp, _vp, _vp]),
    "lcpc_random_coeffs_device": (_i32, [_vp, _vp, _u64, _u64, _vp, _vp]),
    "lcpc_encode_rows": (_i32, [_vp, _vp, _u64]),
    "lcpc_commit_create": (_i32, [_vp, C.POINTER(_vp)]),
    "lcpc_commit_destroy": (None, [_vp]),
    "lcpc_commit_last_error": (C.c_char_p, [_vp]),
    "lcpc_commit": (_i32, [_vp, _vp, _u64, _vp]),
    "lcpc_commit_device": (_i32, [_vp, _vp, _u64, _vp, _u32, _vp]),
    "lcpc_commit_from_parts": (_i32, [_vp, _vp, _vp, _u64, _vp]),
    "lcpc_commit_bincode_size": (_u64, [_vp]),
    "lcpc_commit_bincode_write": (_i32, [_vp, WRITE_FN, _vp]),
    "lcpc_commit_from_bincode": (_i32, [_vp, WRITE_FN, _vp, _vp]),
    "lcpc_get_root": (_i32, [_vp, _vp]),
    "lcpc_commit_dims": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "lcpc_get_hashes": (_i32, [_vp, _vp]),
    "lcpc_get_comm": (_i32, [_vp, _u64, _u64, _vp]),
    "lcpc_get_coeffs": (_i32, [_vp, _u64, _u64, _vp]),
    "lcpc_collapse": (_i32, [_vp, _vp, _u32, _vp]),
    "lcpc_open_columns": (_i32, [_vp, _vp, _u32, _vp, _vp]),
    "lcpc_transcript_new": (_vp, [C.c_char_p, _sz]),
    "lcpc_transcript_clone": (_vp, [_vp]),
    "lcpc_transcript_append_message": (None, [_vp, C.c_char_p, _sz, C.c_char_p, _sz]),
    "lcpc_transcript_append_messages": (None, [_vp, C.c_char_p, _sz, C.c_char_p, _sz, _sz]),
    "lcpc_transcript_challenge_bytes": (None, [_vp, C.c_char_p, _sz, _vp, _sz]),
    "lcpc_transcript_free": (None, [_vp]),
    "lcpc_prove": (_i32, [_vp, _vp, _u64, _vp, _vp, _vp, _vp]),
    "lcpc_verify": (_i32, [_vp, _vp, _vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp]),
    "lcpc_root_bincode": (None, [_vp, _vp]),
    "lcpc_free": (None, [_vp]),
    "lcpc_shard_layout": (_i32, [_vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    "lcpc_shard_nodes": (_i32, [_u64, _u32, _u32, _vp, _vp, _vp]),
    "lcpc_shard_nodes_field": (_i32, [_u32, _u64, _u32, _u32, _vp, _vp, _vp]),
    "lcpc_comm_unique_id": (_i32, [_vp]),
    "lcpc_comm_init": (_i32, [_vp, _vp, _u32, _u32]),
    "lcpc_comm_destroy": (_i32, [_vp]),
    "lcpc_commit_sharded_device": (_i32, [_vp, _vp, _u64, _vp, _u32, _vp]),
    "lcpc_prove_sharded_rccl": (_i32, [_vp, _vp, _u64, _vp, _vp, _vp, _vp]),
    "lcpc_commit_shard_device": (_i32, [_vp, _vp, _u64, _vp, _u32, _vp]),
    "lcpc_commit_finish_device": (_i32, [_vp, _vp, _u64, _u32, _vp, _vp]),
    "lcpc_collapse_device": (_i32, [_vp, _vp, _u32, _vp, _vp]),
    "lcpc_field_sum_device": (_i32, [_vp, _vp, _u32, _u64, _vp, _vp]),
    "lcpc_prove_sharded_bytes": (_u64, [_vp, _u64]),
    "lcpc_prove_sharded": (_i32, [_vp, _vp, _u64, _vp, _vp, _vp, _u64, ALLGATHER_FN, _vp, _vp, _vp, _vp]),
    "lcpc_shard_exchange_probe": (_i32, [_vp, _vp, C.POINTER(_u64)]),
    "lcpc_comm_rccl_version": (_i32, [C.POINTER(_i32)]),
    "lcpc_set_timing": (_i32, [_vp, _i32]),
    "lcpc_get_timings": (_i32, [_vp, C.POINTER(LcpcTimings)]),
}


def build(force=False):
    """compile lcpc_amd/lib/liblcpc_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("lcpc_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)" % LIB_PATH)
        try:
            # when torch shares the process (tests, bench.py) its bundled HIP runtime must be the one that gets
            # loaded first; loading ROCm's copy first leaves torch unable to see the GPU ("No HIP GPUs are available")
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if L.lcpc_abi_version() != ABI_VERSION:
            raise RuntimeError("lcpc_amd: ABI version mismatch")
        _lib = L
    return _lib
