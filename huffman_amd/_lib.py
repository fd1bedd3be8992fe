"""ctypes binding of libhuffhip.so (C ABI: include/huffhip.h).

The library is built in-tree by ``huffman_amd/csrc/Makefile`` (``__graft_entry__.build()``).
There is no CPU fallback: if the shared object is missing, loading raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HUFFHIP_LIB selects another build of the same library (kernel experiments)
LIB_PATH = os.environ.get("HUFFHIP_LIB") or os.path.join(_HERE, "lib", "libhuffhip.so")

# ---- constants mirrored from include/huffhip.h ------------------------------------------------
HF_ABI_VERSION = 3
HF_OK = 0
HF_ERR_ARG, HF_ERR_ALIGN, HF_ERR_RANGE, HF_ERR_HIP = -1, -2, -3, -4
HF_ERR_OVERFLOW, HF_ERR_TIMEOUT, HF_ERR_CORRUPT, HF_ERR_NOMEM, HF_ERR_RCCL = -5, -6, -7, -8, -9
HF_LAYOUT_U8, HF_LAYOUT_BF16_EXP, HF_LAYOUT_FP8 = 0, 1, 2
HF_LAYOUT_FP16_EXP, HF_LAYOUT_FP8_E4M3_EXP = 3, 4
HF_LAYOUT_FP4_E2M1_EXP, HF_LAYOUT_BITPLANE8 = 5, 6
HF_BITPLANE_BLOCK = 131072
HF_CHUNK_SYMS = 16384
HF_BLOCK_MAX_SYMS = 65536
HF_MAX_CODE_LEN = 16
HF_MODE_BYPASS, HF_MODE_HUFFMAN, HF_MODE_AUTO, HF_MODE_BEST = 0, 1, 2, 3
HF_CODEBOOK_DEV_BYTES = 36864
HF_CHUNK_CODEBOOK_BYTES = 1552
HF_STATIC_CB_MAX_CHUNKS = 4096
HF_STATIC_CB_SLOTS = 4
HF_CONTAINER_VERSION = 2
HF_CONTAINER_HEADER_BYTES = 128
HF_CONTAINER_CHUNK_CODEBOOK_BYTES = 296
HF_CODEBOOK_STREAM, HF_CODEBOOK_PER_CHUNK = 0, 1
HF_PLACEMENT_DENSE, HF_PLACEMENT_SLOTTED, HF_PLACEMENT_PAGED = 0, 1, 2
HF_RESERVE_ENCODE, HF_RESERVE_CODEBOOKS, HF_RESERVE_PLACED, HF_RESERVE_ALL = 1, 2, 4, 7
HF_PAGE_HDR_WORDS, HF_PAGE_GAP_WORDS = 10, 48


def HF_SLOT_WORDS(bits: int) -> int:
    """Words of a chunk's slot in a slotted stream, from the bits of its codes."""
    return (bits + 31) // 32 + 126


class hf_entropy_config(C.Structure):
    _fields_ = [
        ("encoding_mode", C.c_uint32),
        ("tree_depth_limit", C.c_uint32),
        ("min_freq_threshold", C.c_uint64),
    ]


class hf_adaptive_config(C.Structure):
    _fields_ = [
        ("encoding_mode", C.c_uint32),
        ("compression_threshold_q16", C.c_uint32),
        ("tree_depth_limit", C.c_uint32),
        ("reserved", C.c_uint32),
        ("min_freq_threshold", C.c_uint64),
    ]


class hf_container_info(C.Structure):
    _fields_ = [
        ("version", C.c_uint32), ("layout", C.c_uint32), ("codebook_kind", C.c_uint32),
        ("chunk_syms", C.c_uint32), ("max_code_len", C.c_uint32), ("mode", C.c_uint32),
        ("entropy_q16", C.c_uint32), ("placement", C.c_uint32),
        ("n", C.c_uint64), ("nchunks", C.c_uint64), ("payload_words", C.c_uint64),
        ("off_codebook", C.c_uint64), ("off_chunk_off", C.c_uint64), ("off_side", C.c_uint64),
        ("off_payload", C.c_uint64), ("total_bytes", C.c_uint64), ("off_chunk_words", C.c_uint64),
    ]


_vp = C.c_void_p
_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); every symbol include/huffhip.h declares
SIGNATURES = {
    "hf_abi_version": (C.c_int, []),
    "hf_ctx_create": (C.c_int, [C.POINTER(_vp), C.c_int, _vp]),
    "hf_ctx_destroy": (C.c_int, [_vp]),
    "hf_flush": (C.c_int, [_vp]),
    "hf_sync": (C.c_int, [_vp]),
    "hf_error_string": (C.c_char_p, [C.c_int]),
    "hf_last_hip_error": (C.c_int, [_vp]),
    "hf_ctx_reserve": (C.c_int, [_vp, C.c_uint64, C.c_uint]),
    "hf_symbol_stat": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.c_int, C.c_int, _u64p]),
    "hf_symbol_stat_dev": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, _vp]),
    "hf_symbol_stat_allreduce": (C.c_int, [_vp, _vp, _vp]),
    "hf_zero_dev": (C.c_int, [_vp, _vp, C.c_uint64]),
    "hf_fields_side_bytes": (C.c_uint64, [C.c_uint64, C.c_int]),
    "hf_fields_split": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, _vp, _vp]),
    "hf_fields_merge": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_int, _vp]),
    "hf_symbol_sort": (C.c_int, [_vp, _u64p, C.c_int, _u64p, _u16p]),
    "hf_entropy": (C.c_int, [_vp, _u64p, C.c_int, _u32p, _ip]),
    "hf_adaptive_threshold": (C.c_int, [C.c_uint32, C.c_int, C.c_int,
                                        C.POINTER(hf_adaptive_config)]),
    "hf_tree_build": (C.c_int, [_vp, _u64p, C.c_int, C.c_int, _u64p, _u16p, _u16p, _u16p, _u16p]),
    "hf_tree_adjust": (C.c_int, [_vp, _u16p, C.c_int, C.c_int, _u16p]),
    "hf_codebook": (C.c_int, [_vp, _u64p, C.c_int, C.POINTER(hf_entropy_config), _u16p, _u8p]),
    "hf_system_codebook_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_uint64, _vp]),
    "hf_system_codebook_block_dev": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.c_int, C.c_uint64,
                                               _vp, _vp]),
    "hf_codebook_upload": (C.c_int, [_vp, _u16p, _u8p, C.c_int, _vp]),
    "hf_codebook_download": (C.c_int, [_vp, _vp, _u16p, _u8p, _ip, _u32p]),
    "hf_encode_pairs": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, _vp, _vp]),
    "hf_encode": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, _vp, _vp, _vp, C.c_uint64, _vp, _u64p]),
    "hf_num_chunks": (C.c_uint64, [C.c_uint64]),
    "hf_payload_bound_words": (C.c_uint64, [C.c_uint64]),
    "hf_decode": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64, C.c_int, _vp, _vp, C.c_int]),
    "hf_symbol_stat_pages_dev": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, _vp, _vp]),
    "hf_symbol_stat_chunks_dev": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, _vp]),
    "hf_system_codebook_chunks_dev": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.c_uint64, _vp]),
    "hf_system_codebook_chunks_static": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.c_uint64, _vp]),
    "hf_encode_chunks": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, _vp, _vp, _vp, C.c_uint64, _vp,
                                   _u64p]),
    "hf_decode_chunks": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64, C.c_int, _vp, _vp,
                                   C.c_uint64, _vp, C.c_int]),
    "hf_decode_chunks_to": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64, C.c_int, _vp, _vp,
                                      _vp, C.c_uint64, _vp, C.c_uint64, C.c_int]),
    "hf_slotted_bound_words": (C.c_uint64, [C.c_uint64]),
    "hf_encode_chunks_slotted": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, _vp, _vp, _vp, _vp,
                                           C.c_uint64, _vp, _vp, _u64p]),
    "hf_decode_chunks_slotted": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp, C.c_uint64, C.c_int,
                                           _vp, _vp, _vp, C.c_uint64, _vp, C.c_uint64, C.c_int]),
    "hf_symbol_stat_fold_dev": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "hf_encode_slotted": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, _vp, _vp, _vp, _vp, C.c_uint64,
                                    _vp, _vp, _u64p]),
    "hf_decode_slotted": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp, C.c_uint64, C.c_int, _vp,
                                    _vp, C.c_int]),
    "hf_pages_bound_words": (C.c_uint64, [C.c_uint64]),
    "hf_encode_pages": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, _vp, _vp, _vp, _vp, C.c_uint64, _vp,
                                  _u64p]),
    "hf_decode_pages": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64, C.c_int, _vp, _vp,
                                  C.c_uint64, _vp, C.c_uint64, C.c_int]),
    "hf_encode_pages_shared": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, _vp, _vp, _vp, _vp, C.c_uint64,
                                         _vp, _u64p]),
    "hf_decode_pages_shared": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64, C.c_int, _vp, _vp,
                                         _vp, C.c_uint64, _vp, C.c_uint64, C.c_int]),
    "hf_page_codebook_download": (C.c_int, [_vp, _vp, C.c_uint64, _vp, C.c_uint64, _u16p, _u8p, _ip,
                                            _u32p]),
    "hf_chunk_codebook_download": (C.c_int, [_vp, _vp, C.c_uint64, _u16p, _u8p, _ip, _u32p]),
    "hf_container_plan": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_uint64,
                                    C.POINTER(hf_container_info)]),
    "hf_container_plan_placed": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                           C.POINTER(hf_container_info)]),
    "hf_container_pack": (C.c_int, [_vp, C.POINTER(hf_container_info), _vp, _vp, _vp, _vp, _vp,
                                    C.c_uint64]),
    "hf_container_pack_placed": (C.c_int, [_vp, C.POINTER(hf_container_info), _vp, _vp, _vp, _vp,
                                           _vp, _vp, C.c_uint64]),
    "hf_container_open": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(hf_container_info)]),
    "hf_container_codebook": (C.c_int, [_vp, _vp, C.POINTER(hf_container_info), _vp]),
    "hf_quality": (C.c_int, [_vp, _u64p, _u8p, C.c_int, C.c_int, _u16p, _u16p, _u16p]),
    "hf_last_kernel_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "hf_set_timing": (C.c_int, [_vp, C.c_int]),
}

_lib = None


class HuffhipError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        msg = lib().hf_error_string(code).decode() if _lib is not None else str(code)
        super().__init__(f"{where}: {msg} ({code})")


def lib() -> C.CDLL:
    """Load libhuffhip.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C huffman_amd/csrc` (there is no CPU fallback)"
            )
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.hf_abi_version() != HF_ABI_VERSION:
            raise RuntimeError("libhuffhip.so ABI version mismatch: rebuild")
        _lib = l
    return _lib


def check(rc: int, where: str) -> None:
    if rc != HF_OK:
        raise HuffhipError(rc, where)
