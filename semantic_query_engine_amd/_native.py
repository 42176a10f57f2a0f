"""ctypes binding of libsqe.so (include/sqe.h).

The HIP library is the product; there is no CPU fallback.  If ``libsqe.so`` is missing
or does not export a symbol the header declares, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SQE_LIB", os.path.join(_HERE, "libsqe.so"))

c_float_p = C.POINTER(C.c_float)
c_i64_p = C.POINTER(C.c_int64)
c_i32_p = C.POINTER(C.c_int32)


class BertCfg(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32),
                ("heads", C.c_int32), ("inter", C.c_int32), ("max_pos", C.c_int32),
                ("type_vocab", C.c_int32), ("ln_eps", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("scan_ms", C.c_double), ("prep_ms", C.c_double), ("select_ms", C.c_double),
                ("add_ms", C.c_double), ("encode_ms", C.c_double), ("cache_ms", C.c_double),
                ("scan_calls", C.c_int64), ("search_calls", C.c_int64), ("scan_rows", C.c_int64),
                ("scan_flops", C.c_int64), ("scan_bytes", C.c_int64), ("uncertified", C.c_int64),
                ("sample_ms", C.c_double), ("i8_collected", C.c_int64), ("i8_rescored", C.c_int64),
                ("i8_overflows", C.c_int64)]


class I8Launch(C.Structure):
    _fields_ = [("rows", C.c_int64), ("tile_stride", C.c_int64), ("dim", C.c_int32), ("B", C.c_int32), ("b_pad", C.c_int32),
                ("k", C.c_int32), ("tile_rows", C.c_int32), ("q_pitch", C.c_int32), ("query_block", C.c_int32),
                ("n_chunks", C.c_int32), ("list_cap", C.c_int32), ("sample_int8", C.c_int32), ("sample_step", C.c_int32),
                ("sample_tiles", C.c_int32), ("sample_chunks", C.c_int32), ("sample_b_pad", C.c_int32), ("sample_m", C.c_int32),
                ("uncertified", C.c_int32), ("pool_cap", C.c_int32)]


# name -> (restype, argtypes): every symbol include/sqe.h declares
SIGNATURES = {
    "sqe_version": (C.c_int, []),
    "sqe_last_error": (C.c_char_p, []),
    "sqe_create": (C.c_int, [c_i32_p, C.c_int, C.POINTER(C.c_void_p)]),
    "sqe_create_sharded": (C.c_int, [c_i32_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "sqe_group_info": (C.c_int, [C.c_void_p, c_i32_p, c_i32_p, c_i32_p, C.c_int]),
    "sqe_destroy": (None, [C.c_void_p]),
    "sqe_synchronize": (C.c_int, [C.c_void_p]),
    "sqe_stream": (C.c_void_p, [C.c_void_p]),
    "sqe_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sqe_device_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, c_i32_p, c_i64_p]),
    "sqe_index_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "sqe_index_destroy": (None, [C.c_void_p]),
    "sqe_index_reserve": (C.c_int, [C.c_void_p, C.c_int64]),
    "sqe_index_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "sqe_index_add_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "sqe_index_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "sqe_index_count": (C.c_int, [C.c_void_p, c_i64_p]),
    "sqe_index_get_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "sqe_index_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "sqe_index_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sqe_index_search_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sqe_index_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_uint64]),
    "sqe_index_train_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_uint64]),
    "sqe_index_ivf_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "sqe_index_i8_last": (C.c_int, [C.c_void_p, C.POINTER(I8Launch)]),
    "sqe_index_i8_read": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64]),
    "sqe_index_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "sqe_index_load": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "sqe_merge_topk_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sqe_cosine_best": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, c_float_p, c_i32_p]),
    "sqe_cosine_all": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sqe_cache_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "sqe_cache_destroy": (None, [C.c_void_p]),
    "sqe_cache_set_slot": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "sqe_cache_best": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, c_float_p, c_i32_p]),
    "sqe_encoder_create": (C.c_int, [C.c_void_p, C.POINTER(BertCfg), C.POINTER(C.c_void_p)]),
    "sqe_encoder_destroy": (None, [C.c_void_p]),
    "sqe_encoder_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, c_i64_p, C.c_int]),
    "sqe_encoder_finalize": (C.c_int, [C.c_void_p]),
    "sqe_tokenizer_create": (C.c_int, [C.c_char_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "sqe_tokenizer_destroy": (None, [C.c_void_p]),
    "sqe_tokenize": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64, C.c_int, c_i32_p, c_i32_p]),
    "sqe_tokenize_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), c_i64_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sqe_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sqe_encode_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sqe_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "sqe_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "sqe_stats_reset": (C.c_int, [C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


class SqeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libsqe error {code}: {msg}")
        self.code = code


def load() -> C.CDLL:
    """Load libsqe.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C semantic_query_engine_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    # A PyTorch-ROCm wheel bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1.  If libsqe.so pulls in the system
    # copies first and torch is imported later, the process holds two HIP runtimes and torch's finds no device
    # (hipErrorNoDevice at its first CUDA call).  With torch loaded first the dynamic linker binds libsqe.so to the
    # copies already in the process (same SONAMEs).  Nothing of torch is used here.
    # A caller that never imports torch (a plain ctypes / NumPy host) can skip the multi-second import: SQE_NO_TORCH_PRELOAD=1.
    import sys
    if "torch" not in sys.modules and os.environ.get("SQE_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise SqeError(rc, load().sqe_last_error().decode("utf-8", "replace"))
