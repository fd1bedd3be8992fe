"""ctypes front end of the CPU oracle (oracle/hf_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py; the shipped package ``huffman_amd`` never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

CHUNK_SYMS = 16384
LAYOUT_U8, LAYOUT_BF16_EXP, LAYOUT_FP8 = 0, 1, 2
LAYOUT_FP16_EXP, LAYOUT_FP8_E4M3_EXP = 3, 4
LAYOUT_FP4_E2M1_EXP, LAYOUT_BITPLANE8 = 5, 6
BITPLANE_BLOCK = 8 * 16384


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc if it is missing or older than its sources."""
    srcs = [os.path.join(_HERE, f) for f in ("hf_oracle.c", "hf_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


def _declare(l: C.CDLL) -> None:
    l.orc_side_bytes.argtypes = [C.c_uint64, C.c_int]
    l.orc_side_bytes.restype = C.c_uint64
    l.orc_symbolize.argtypes = [C.c_void_p, C.c_size_t, C.c_int, _u8p, _u8p]
    l.orc_symbolize.restype = None
    l.orc_desymbolize.argtypes = [_u8p, _u8p, C.c_size_t, C.c_int, C.c_void_p]
    l.orc_desymbolize.restype = None
    l.orc_aw_symbol_stat.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, _u64p]
    l.orc_aw_symbol_stat.restype = None
    l.orc_aw_symbol_sort.argtypes = [_u64p, C.c_int, _u64p, _u16p]
    l.orc_aw_symbol_sort.restype = None
    l.orc_aw_tree_build.argtypes = [_u64p, C.c_int, C.c_int, _u64p, _u16p, _u16p, _u16p, _u16p]
    l.orc_aw_tree_build.restype = None
    l.orc_aw_code_start.argtypes = [_u16p, C.c_int, _u16p, _u16p]
    l.orc_aw_code_start.restype = None
    l.orc_aw_codes_as_written.argtypes = [_u16p, C.c_int, _u16p]
    l.orc_aw_codes_as_written.restype = None
    l.orc_aw_entropy_q16.argtypes = [_u64p, C.c_int, C.c_int]
    l.orc_aw_entropy_q16.restype = C.c_uint32
    l.orc_aw_adaptive_mode.argtypes = [C.c_uint32]
    l.orc_aw_adaptive_mode.restype = C.c_int
    l.orc_aw_adaptive_config.argtypes = [C.c_uint32, C.c_int, C.c_int, _u32p, _u32p, _u32p, _u64p]
    l.orc_aw_adaptive_config.restype = C.c_int
    l.orc_aw_filter.argtypes = [_u64p, C.c_int, C.c_uint64, _u64p]
    l.orc_aw_filter.restype = None
    l.orc_aw_encode_pairs.argtypes = [_u8p, C.c_size_t, _u16p, _u8p, C.c_int, C.c_int, _u8p]
    l.orc_aw_encode_pairs.restype = None
    l.orc_aw_quality.argtypes = [_u64p, _u8p, C.c_int, C.c_int, _u16p, _u16p, _u16p]
    l.orc_aw_quality.restype = C.c_int
    l.orc_rp_leaf_depths.argtypes = [_u16p, _u16p, C.c_int, C.c_int, _u16p]
    l.orc_rp_leaf_depths.restype = None
    l.orc_rp_tree_adjust.argtypes = [_u16p, C.c_int, C.c_int, _u16p]
    l.orc_rp_tree_adjust.restype = C.c_int
    l.orc_rp_canonical_codes.argtypes = [_u16p, C.c_int, _u16p]
    l.orc_rp_canonical_codes.restype = None
    l.orc_rp_codebook.argtypes = [_u64p, C.c_int, C.c_int, C.c_uint64, C.c_int, _u16p, _u8p]
    l.orc_rp_codebook.restype = C.c_int
    l.orc_rp_system_codebook.argtypes = [_u64p, C.c_int, C.c_int, C.c_uint64, C.c_int, _u16p, _u8p]
    l.orc_rp_system_codebook.restype = C.c_int
    l.orc_rp_entropy_q16_u64.argtypes = [_u64p, C.c_int]
    l.orc_rp_entropy_q16_u64.restype = C.c_uint32
    l.orc_rp_encode.argtypes = [_u8p, C.c_size_t, _u16p, _u8p, _u32p, _u32p]
    l.orc_rp_encode.restype = C.c_uint64
    l.orc_rp_decode.argtypes = [_u32p, _u32p, C.c_size_t, _u16p, _u8p, _u8p]
    l.orc_rp_decode.restype = C.c_int
    l.orc_rp_encode_blocks.argtypes = [_u8p, C.c_size_t, C.c_uint64, _u16p, _u8p, _u32p, _u32p]
    l.orc_rp_encode_blocks.restype = C.c_uint64
    l.orc_rp_slot_blocks.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, _u32p, _u32p, _u32p, _u32p,
                                     _u32p]
    l.orc_rp_slot_blocks.restype = C.c_uint64
    l.orc_rp_decode_blocks.argtypes = [_u32p, _u32p, C.c_size_t, _u16p, _u8p, _u8p]
    l.orc_rp_decode_blocks.restype = C.c_int
    l.orc_rp_page_head.argtypes = [_u8p]
    l.orc_rp_page_head.restype = C.c_uint32
    l.orc_rp_page_words.argtypes = [C.c_uint32, C.c_uint64]
    l.orc_rp_page_words.restype = C.c_uint64
    l.orc_rp_pages_bound_words.argtypes = [C.c_size_t]
    l.orc_rp_pages_bound_words.restype = C.c_uint64
    l.orc_rp_encode_pages.argtypes = [_u8p, C.c_size_t, C.c_uint64, C.c_int, _u16p, _u8p, _u32p,
                                      _u32p]
    l.orc_rp_encode_pages.restype = C.c_uint64
    l.orc_rp_page_codebook.argtypes = [_u32p, C.c_uint64, _u16p, _u8p, _u32p, _u32p]
    l.orc_rp_page_codebook.restype = C.c_int
    l.orc_rp_decode_pages.argtypes = [_u32p, _u32p, C.c_size_t, _u8p]
    l.orc_rp_decode_pages.restype = C.c_int
    l.orc_rp_shared_page_words.argtypes = [C.c_uint64]
    l.orc_rp_shared_page_words.restype = C.c_uint64
    l.orc_rp_encode_pages_shared.argtypes = [_u8p, C.c_size_t, _u16p, _u8p, _u32p, _u32p]
    l.orc_rp_encode_pages_shared.restype = C.c_uint64
    l.orc_rp_decode_pages_shared.argtypes = [_u32p, _u32p, C.c_size_t, _u16p, _u8p, _u8p]
    l.orc_rp_decode_pages_shared.restype = C.c_int
    l.orc_num_chunks.argtypes = [C.c_size_t]
    l.orc_num_chunks.restype = C.c_size_t
    l.orc_payload_bound_words.argtypes = [C.c_size_t]
    l.orc_payload_bound_words.restype = C.c_uint64


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(t)


def _arr(x, dt) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=dt))


# ---- front end -------------------------------------------------------------------------------
def side_bytes(n: int, layout: int) -> int:
    return int(lib().orc_side_bytes(n, layout))


def symbolize(data: np.ndarray, layout: int):
    data = np.ascontiguousarray(data)
    if layout == LAYOUT_BITPLANE8:
        assert data.dtype == np.uint8 and data.size % 8 == 0
        sym = np.empty(data.size, np.uint8)
        lib().orc_symbolize(data.ctypes.data, data.size, layout, _p(sym, _u8p), None)
        return sym, None
    if layout in (LAYOUT_FP16_EXP, LAYOUT_FP8_E4M3_EXP, LAYOUT_FP4_E2M1_EXP):
        assert data.dtype == (np.uint16 if layout == LAYOUT_FP16_EXP else np.uint8)
        n = data.size
        sym = np.empty(n, np.uint8)
        side = np.zeros(max(1, side_bytes(n, layout)), np.uint8)
        lib().orc_symbolize(data.ctypes.data, n, layout, _p(sym, _u8p), _p(side, _u8p))
        return sym, side[:side_bytes(n, layout)]
    if layout == LAYOUT_BF16_EXP:
        assert data.dtype == np.uint16
        n = data.size
        sym = np.empty(n, np.uint8)
        side = np.empty(n, np.uint8)
        lib().orc_symbolize(data.ctypes.data, n, layout, _p(sym, _u8p), _p(side, _u8p))
        return sym, side
    assert data.dtype == np.uint8
    return data.copy(), None


def desymbolize(sym: np.ndarray, side, layout: int) -> np.ndarray:
    sym = _arr(sym, np.uint8)
    if layout == LAYOUT_BITPLANE8:
        assert sym.size % 8 == 0
        out = np.empty(sym.size, np.uint8)
        lib().orc_desymbolize(_p(sym, _u8p), None, sym.size, layout, out.ctypes.data)
        return out
    if layout in (LAYOUT_FP8_E4M3_EXP, LAYOUT_FP4_E2M1_EXP):
        side = _arr(side, np.uint8)
        out = np.empty(sym.size, np.uint8)
        lib().orc_desymbolize(_p(sym, _u8p), _p(side, _u8p), sym.size, layout, out.ctypes.data)
        return out
    if layout in (LAYOUT_BF16_EXP, LAYOUT_FP16_EXP):
        side = _arr(side, np.uint8)
        out = np.empty(sym.size, np.uint16)
        lib().orc_desymbolize(_p(sym, _u8p), _p(side, _u8p), sym.size, layout, out.ctypes.data)
        return out
    return sym.copy()


# ---- as written -------------------------------------------------------------------------------
def aw_symbol_stat(sym, N: int = 256, W: int = 32) -> np.ndarray:
    sym = _arr(sym, np.uint8)
    freq = np.zeros(N, np.uint64)
    lib().orc_aw_symbol_stat(_p(sym, _u8p), sym.size, N, W, _p(freq, _u64p))
    return freq


def aw_symbol_sort(freq):
    freq = _arr(freq, np.uint64)
    N = freq.size
    out = np.zeros(N, np.uint64)
    ids = np.zeros(N, np.uint16)
    lib().orc_aw_symbol_sort(_p(freq, _u64p), N, _p(out, _u64p), _p(ids, _u16p))
    return out, ids


def aw_tree_build(freqs, W: int = 32):
    freqs = _arr(freqs, np.uint64)
    N = freqs.size
    M = 2 * N - 1
    weights = np.zeros(M, np.uint64)
    lefts = np.zeros(M, np.uint16)
    rights = np.zeros(M, np.uint16)
    heights = np.zeros(M, np.uint16)
    root = C.c_uint16(0)
    lib().orc_aw_tree_build(_p(freqs, _u64p), N, W, _p(weights, _u64p), _p(lefts, _u16p),
                            _p(rights, _u16p), C.byref(root), _p(heights, _u16p))
    return dict(weights=weights, lefts=lefts, rights=rights, root=int(root.value), heights=heights)


def aw_code_start(lens):
    lens = _arr(lens, np.uint16)
    count = np.zeros(17, np.uint16)
    cs = np.zeros(17, np.uint16)
    lib().orc_aw_code_start(_p(lens, _u16p), lens.size, _p(count, _u16p), _p(cs, _u16p))
    return count, cs


def aw_codes_as_written(lens) -> np.ndarray:
    lens = _arr(lens, np.uint16)
    code = np.zeros(lens.size, np.uint16)
    lib().orc_aw_codes_as_written(_p(lens, _u16p), lens.size, _p(code, _u16p))
    return code


def aw_entropy_q16(freq, W: int = 32) -> int:
    freq = _arr(freq, np.uint64)
    return int(lib().orc_aw_entropy_q16(_p(freq, _u64p), freq.size, W))


def aw_adaptive_mode(entropy_q16: int) -> int:
    return int(lib().orc_aw_adaptive_mode(entropy_q16))


def aw_adaptive_config(entropy_q16: int, depth: int = 256, W: int = 32):
    """(mode, compression_threshold_q16, tree_depth_limit, min_freq_threshold), or None where the
    reference's literal does not elaborate (adaptive_threshold.scala:41-79)."""
    m, t, d = C.c_uint32(), C.c_uint32(), C.c_uint32()
    f = C.c_uint64()
    rc = lib().orc_aw_adaptive_config(entropy_q16, depth, W, C.byref(m), C.byref(t), C.byref(d),
                                      C.byref(f))
    return None if rc else (m.value, t.value, d.value, f.value)


def aw_filter(freq, thr: int) -> np.ndarray:
    freq = _arr(freq, np.uint64)
    out = np.zeros_like(freq)
    lib().orc_aw_filter(_p(freq, _u64p), freq.size, thr, _p(out, _u64p))
    return out


def aw_encode_pairs(sym, code, lens, mode: int, sym_width: int = 8) -> np.ndarray:
    sym = _arr(sym, np.uint8)
    code = _arr(code, np.uint16)
    lens = _arr(lens, np.uint8)
    out = np.zeros((sym.size, 4), np.uint8)
    lib().orc_aw_encode_pairs(_p(sym, _u8p), sym.size, _p(code, _u16p), _p(lens, _u8p), mode,
                              sym_width, _p(out, _u8p))
    return out


def aw_quality(freq, lens, W: int = 32):
    freq = _arr(freq, np.uint64)
    lens = _arr(lens, np.uint8)
    a, e, r = C.c_uint16(0), C.c_uint16(0), C.c_uint16(0)
    rc = lib().orc_aw_quality(_p(freq, _u64p), _p(lens, _u8p), freq.size, W, C.byref(a),
                              C.byref(e), C.byref(r))
    if rc != 0:
        return None
    return int(a.value), int(e.value), int(r.value)


# ---- repaired ---------------------------------------------------------------------------------
def rp_leaf_depths(lefts, rights, root: int, N: int) -> np.ndarray:
    lefts = _arr(lefts, np.uint16)
    rights = _arr(rights, np.uint16)
    d = np.zeros(N, np.uint16)
    lib().orc_rp_leaf_depths(_p(lefts, _u16p), _p(rights, _u16p), root, N, _p(d, _u16p))
    return d


def rp_tree_adjust(depths, lmax: int = 16):
    depths = _arr(depths, np.uint16)
    out = np.zeros_like(depths)
    steps = lib().orc_rp_tree_adjust(_p(depths, _u16p), depths.size, lmax, _p(out, _u16p))
    if steps < 0:
        raise ValueError("2^lmax < N")
    return out, int(steps)


def rp_canonical_codes(lens) -> np.ndarray:
    lens = _arr(lens, np.uint16)
    code = np.zeros(lens.size, np.uint16)
    lib().orc_rp_canonical_codes(_p(lens, _u16p), lens.size, _p(code, _u16p))
    return code


def rp_codebook(freq, W: int = 64, thr: int = 0, lmax: int = 16):
    freq = _arr(freq, np.uint64)
    code = np.zeros(freq.size, np.uint16)
    lens = np.zeros(freq.size, np.uint8)
    rc = lib().orc_rp_codebook(_p(freq, _u64p), freq.size, W, thr, lmax, _p(code, _u16p),
                               _p(lens, _u8p))
    if rc != 0:
        raise ValueError("codebook failed")
    return code, lens


def rp_system_codebook(freq_by_symbol, W: int = 64, thr: int = 0, lmax: int = 16):
    freq = _arr(freq_by_symbol, np.uint64)
    code = np.zeros(freq.size, np.uint16)
    lens = np.zeros(freq.size, np.uint8)
    rc = lib().orc_rp_system_codebook(_p(freq, _u64p), freq.size, W, thr, lmax, _p(code, _u16p),
                                      _p(lens, _u8p))
    if rc != 0:
        raise ValueError("codebook failed")
    return code, lens


def rp_entropy_q16_u64(freq) -> int:
    freq = _arr(freq, np.uint64)
    return int(lib().orc_rp_entropy_q16_u64(_p(freq, _u64p), freq.size))


def num_chunks(n: int) -> int:
    return (n + CHUNK_SYMS - 1) // CHUNK_SYMS


def container_bytes(n, layout, kind, payload_words) -> int:
    """Size of a version-1 container (kind 0: one codebook, 1: one per chunk)."""
    l = lib()
    l.orc_container_bytes.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_uint64]
    l.orc_container_bytes.restype = C.c_uint64
    return int(l.orc_container_bytes(n, layout, kind, payload_words))


def container_pack(n, layout, payload, chunk_off, side=None, code=None, lens=None, mode=1,
                   entropy_q16=0, rank2sym=None, len_by_rank=None, modes=None, entropies=None,
                   placement=0, chunk_words=None):
    """The container as bytes.  One codebook (code, lens, mode, entropy_q16) or per-chunk canonical
    orders (rank2sym, len_by_rank: nchunks x 256; modes, entropies).  placement 0: dense, wire
    format version 1; 1: slotted (with chunk_words); 2: paged (no codebook section, or the one codebook of pages
    without records when code / lens are given) -- version 2."""
    l = lib()
    l.orc_container2_bytes.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint64]
    l.orc_container2_bytes.restype = C.c_uint64
    l.orc_container2_pack.argtypes = [_u8p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                      C.c_uint32, C.c_uint32, _u16p, _u8p, _u8p, _u8p, _u32p, _u32p,
                                      _u32p, _u32p, _u8p, _u32p]
    l.orc_container2_pack.restype = C.c_uint64
    kind = 1 if (rank2sym is not None or (placement == 2 and code is None)) else 0
    payload = _arr(payload, np.uint32)
    chunk_off = _arr(chunk_off, np.uint32)
    total = int(l.orc_container2_bytes(n, layout, kind, placement, payload.size))
    if placement == 0:
        assert total == container_bytes(n, layout, kind, payload.size)
    out = np.zeros(total, np.uint8)
    z8, z16, z32 = np.zeros(1, np.uint8), np.zeros(1, np.uint16), np.zeros(1, np.uint32)
    code = _arr(code, np.uint16) if code is not None else z16
    lens = _arr(lens, np.uint8) if lens is not None else z8
    side = _arr(side, np.uint8) if side is not None else z8
    r2s = _arr(rank2sym, np.uint8).reshape(-1) if rank2sym is not None else z8
    lbr = _arr(len_by_rank, np.uint8).reshape(-1) if len_by_rank is not None else z8
    modes = _arr(modes, np.uint32) if modes is not None else z32
    ents = _arr(entropies, np.uint32) if entropies is not None else z32
    words = _arr(chunk_words, np.uint32) if chunk_words is not None else z32
    pay = payload if payload.size else z32
    got = l.orc_container2_pack(_p(out, _u8p), n, layout, kind, placement, payload.size, mode,
                                entropy_q16, _p(code, _u16p), _p(lens, _u8p), _p(r2s, _u8p),
                                _p(lbr, _u8p), _p(modes, _u32p), _p(ents, _u32p),
                                _p(chunk_off, _u32p), _p(words, _u32p), _p(side, _u8p),
                                _p(pay, _u32p))
    assert int(got) == total
    return out


def rp_best_mode(freq, lens) -> int:
    freq = _arr(freq, np.uint64)
    lens = _arr(lens, np.uint8)
    l = lib()
    l.orc_rp_best_mode.argtypes = [_u64p, _u8p, C.c_int]
    l.orc_rp_best_mode.restype = C.c_int
    return int(l.orc_rp_best_mode(_p(freq, _u64p), _p(lens, _u8p), freq.size))


def payload_bound_words(n: int) -> int:
    return int(lib().orc_payload_bound_words(n))


def rp_encode(sym, code, lens):
    """Returns (payload words u32, chunk_off u32[nchunks+1])."""
    sym = _arr(sym, np.uint8)
    code = _arr(code, np.uint16)
    lens = _arr(lens, np.uint8)
    nch = num_chunks(sym.size)
    off = np.zeros(nch + 1, np.uint32)
    total = int(lib().orc_rp_encode(_p(sym, _u8p), sym.size, _p(code, _u16p), _p(lens, _u8p),
                                    None, _p(off, _u32p)))
    payload = np.zeros(max(total, 1), np.uint32)
    lib().orc_rp_encode(_p(sym, _u8p), sym.size, _p(code, _u16p), _p(lens, _u8p),
                        _p(payload, _u32p), _p(off, _u32p))
    return payload[:total], off


def rp_decode(payload, chunk_off, n: int, code, lens) -> np.ndarray:
    payload = _arr(payload, np.uint32)
    if payload.size == 0:
        payload = np.zeros(1, np.uint32)
    chunk_off = _arr(chunk_off, np.uint32)
    code = _arr(code, np.uint16)
    lens = _arr(lens, np.uint8)
    out = np.zeros(max(n, 1), np.uint8)
    rc = lib().orc_rp_decode(_p(payload, _u32p), _p(chunk_off, _u32p), n, _p(code, _u16p),
                             _p(lens, _u8p), _p(out, _u8p))
    if rc != 0:
        raise ValueError("payload does not decode")
    return out[:n]


def rp_encode_blocks(sym, thr: int = 0, payload_words: int = 0):
    """Per-block codebooks, Huffman forced: (payload u32, chunk_off u32[nchunks+1],
    code u16[nchunks, 256], len u8[nchunks, 256]).  `payload_words` > 0 skips the sizing pass
    (the buffer must be large enough)."""
    sym = _arr(sym, np.uint8)
    nch = num_chunks(sym.size)
    off = np.zeros(nch + 1, np.uint32)
    code = np.zeros((max(nch, 1), 256), np.uint16)
    lens = np.zeros((max(nch, 1), 256), np.uint8)
    if payload_words <= 0:
        payload_words = int(lib().orc_rp_encode_blocks(_p(sym, _u8p), sym.size, thr,
                                                       _p(code, _u16p), _p(lens, _u8p), None,
                                                       _p(off, _u32p)))
    payload = np.zeros(max(payload_words, 1), np.uint32)
    total = int(lib().orc_rp_encode_blocks(_p(sym, _u8p), sym.size, thr, _p(code, _u16p),
                                           _p(lens, _u8p), _p(payload, _u32p), _p(off, _u32p)))
    return payload[:total], off, code[:nch], lens[:nch]


def slot_words(bits: int) -> int:
    """ORC_SLOT_WORDS: words of the slot of a chunk whose codes take `bits` bits."""
    return (bits + 31) // 32 + 126


def rp_encode_blocks_slotted(sym, thr: int = 0):
    """The per-block stream with every chunk in a slot of ceil(bits / 32) + 126 words:
    (payload u32, slot_off u32[nchunks+1], words u32[nchunks], code, len)."""
    sym = _arr(sym, np.uint8)
    payload, off, code, lens = rp_encode_blocks(sym, thr)
    nch = num_chunks(sym.size)
    lens_c = _arr(lens if nch else np.zeros((1, 256), np.uint8), np.uint8)
    pay = payload if payload.size else np.zeros(1, np.uint32)
    out, slot_off, words = _slot(sym, lens_c, 256, pay, off)
    return out, slot_off, words[:nch], code, lens


def _slot(sym, lens, stride, pay, off):
    nch = num_chunks(sym.size)
    slot_off = np.zeros(nch + 1, np.uint32)
    words = np.zeros(max(nch, 1), np.uint32)
    args = (_p(sym, _u8p), sym.size, _p(lens, _u8p), stride, _p(pay, _u32p), _p(off, _u32p))
    total = int(lib().orc_rp_slot_blocks(*args, None, _p(slot_off, _u32p), _p(words, _u32p)))
    assert total != 2**64 - 1, "a chunk outgrew its slot"
    out = np.zeros(max(total, 1), np.uint32)
    lib().orc_rp_slot_blocks(*args, _p(out, _u32p), _p(slot_off, _u32p), _p(words, _u32p))
    return out[:total], slot_off, words


def rp_encode_slotted(sym, code, lens):
    """The stream of rp_encode (one codebook) with every chunk in its slot:
    (payload u32, slot_off u32[nchunks+1], words u32[nchunks])."""
    sym = _arr(sym, np.uint8)
    lens = _arr(lens, np.uint8)
    payload, off = rp_encode(sym, code, lens)
    pay = payload if payload.size else np.zeros(1, np.uint32)
    out, slot_off, words = _slot(sym, lens, 0, pay, off)
    return out, slot_off, words[:num_chunks(sym.size)]


def rp_decode_blocks(payload, chunk_off, n: int, code, lens) -> np.ndarray:
    payload = _arr(payload, np.uint32)
    if payload.size == 0:
        payload = np.zeros(1, np.uint32)
    chunk_off = _arr(chunk_off, np.uint32)
    code = _arr(code, np.uint16)
    lens = _arr(lens, np.uint8)
    out = np.zeros(max(n, 1), np.uint8)
    rc = lib().orc_rp_decode_blocks(_p(payload, _u32p), _p(chunk_off, _u32p), n,
                                    _p(code, _u16p), _p(lens, _u8p), _p(out, _u8p))
    if rc != 0:
        raise ValueError("payload does not decode")
    return out[:n]


# ---- paged stream (format version 2) -----------------------------------------------------------
PAGE_HDR_WORDS, PAGE_GAP_WORDS = 10, 48


def page_head(rank2sym) -> int:
    r = _arr(rank2sym, np.uint8)
    assert r.size == 256
    return int(lib().orc_rp_page_head(_p(r, _u8p)))


def page_words(head: int, bits: int) -> int:
    return int(lib().orc_rp_page_words(head, bits))


def pages_bound_words(n: int) -> int:
    return int(lib().orc_rp_pages_bound_words(n))


def rp_encode_pages(sym, thr: int = 0, mode: int = 1):
    """Per-block codebooks, every page self-contained: (payload u32, page_off u32[nchunks+1],
    code u16[nchunks, 256], len u8[nchunks, 256]).  mode: 0 bypass, 1 Huffman, 2 entropy decision,
    3 best."""
    sym = _arr(sym, np.uint8)
    nch = num_chunks(sym.size)
    off = np.zeros(nch + 1, np.uint32)
    code = np.zeros((max(nch, 1), 256), np.uint16)
    lens = np.zeros((max(nch, 1), 256), np.uint8)
    total = int(lib().orc_rp_encode_pages(_p(sym, _u8p), sym.size, thr, mode, _p(code, _u16p),
                                          _p(lens, _u8p), None, _p(off, _u32p)))
    payload = np.zeros(max(total, 1), np.uint32)
    lib().orc_rp_encode_pages(_p(sym, _u8p), sym.size, thr, mode, _p(code, _u16p), _p(lens, _u8p),
                              _p(payload, _u32p), _p(off, _u32p))
    return payload[:total], off, code[:nch], lens[:nch]


def rp_page_codebook(payload, page_off, page: int):
    """(code u16[256], len u8[256], mode, entropy_q16) of a stored page, None if it is damaged."""
    payload = _arr(payload, np.uint32)
    page_off = _arr(page_off, np.uint32)
    a, b = int(page_off[page]), int(page_off[page + 1])
    pg = np.ascontiguousarray(payload[a:b]) if b > a else np.zeros(1, np.uint32)
    code = np.zeros(256, np.uint16)
    lens = np.zeros(256, np.uint8)
    mode, ent = C.c_uint32(0), C.c_uint32(0)
    rc = lib().orc_rp_page_codebook(_p(pg, _u32p), max(b - a, 0), _p(code, _u16p), _p(lens, _u8p),
                                    C.byref(mode), C.byref(ent))
    return None if rc else (code, lens, int(mode.value), int(ent.value))


def shared_page_words(bits: int) -> int:
    return int(lib().orc_rp_shared_page_words(bits))


def rp_encode_pages_shared(sym, code, lens):
    """Pages without records, all coded with one codebook: (payload u32, page_off u32[nchunks+1])."""
    sym = _arr(sym, np.uint8)
    code = _arr(code, np.uint16)
    lens = _arr(lens, np.uint8)
    nch = num_chunks(sym.size)
    off = np.zeros(nch + 1, np.uint32)
    total = int(lib().orc_rp_encode_pages_shared(_p(sym, _u8p), sym.size, _p(code, _u16p),
                                                 _p(lens, _u8p), None, _p(off, _u32p)))
    payload = np.zeros(max(total, 1), np.uint32)
    lib().orc_rp_encode_pages_shared(_p(sym, _u8p), sym.size, _p(code, _u16p), _p(lens, _u8p),
                                     _p(payload, _u32p), _p(off, _u32p))
    return payload[:total], off


def rp_decode_pages_shared(payload, page_off, n: int, code, lens) -> np.ndarray:
    payload = _arr(payload, np.uint32)
    if payload.size == 0:
        payload = np.zeros(1, np.uint32)
    page_off = _arr(page_off, np.uint32)
    code = _arr(code, np.uint16)
    lens = _arr(lens, np.uint8)
    out = np.zeros(max(n, 1), np.uint8)
    rc = lib().orc_rp_decode_pages_shared(_p(payload, _u32p), _p(page_off, _u32p), n,
                                          _p(code, _u16p), _p(lens, _u8p), _p(out, _u8p))
    if rc != 0:
        raise ValueError("pages do not decode")
    return out[:n]


def rp_decode_pages(payload, page_off, n: int) -> np.ndarray:
    payload = _arr(payload, np.uint32)
    if payload.size == 0:
        payload = np.zeros(1, np.uint32)
    page_off = _arr(page_off, np.uint32)
    out = np.zeros(max(n, 1), np.uint8)
    rc = lib().orc_rp_decode_pages(_p(payload, _u32p), _p(page_off, _u32p), n, _p(out, _u8p))
    if rc != 0:
        raise ValueError("pages do not decode")
    return out[:n]
