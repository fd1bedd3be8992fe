"""CPU suite: container layout arithmetic of the library (host-only call) against the oracle's
restatement, and the oracle's container against a hand-built one."""
import ctypes as C
import struct

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib():
    import huffman_amd
    return huffman_amd.lib()


def test_plan_matches_oracle_sizes(lib, orc):
    from huffman_amd import _lib as L
    ol = orc.lib()
    ol.orc_container_bytes.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_uint64]
    ol.orc_container_bytes.restype = C.c_uint64
    info = L.hf_container_info()
    for n in (0, 1, 15, 16384, 16385, 10**6 + 7, 8_030_261_248):
        for layout in (0, 1, 2):
            for kind in (0, 1):
                for words in (0, 1, 3, 4, 5, 12345, n // 8):
                    assert lib.hf_container_plan(n, layout, kind, words, C.byref(info)) == 0
                    assert info.total_bytes == ol.orc_container_bytes(n, layout, kind, words)
                    for off in (info.off_codebook, info.off_chunk_off, info.off_side,
                                info.off_payload, info.total_bytes):
                        assert off % 16 == 0
                    assert (info.off_side != 0) == (layout == 1)
                    assert info.nchunks == orc.num_chunks(n)
    assert lib.hf_container_plan(10, 7, 0, 1, C.byref(info)) == L.HF_ERR_ARG
    assert lib.hf_container_plan(10, 0, 2, 1, C.byref(info)) == L.HF_ERR_ARG


def test_oracle_container_fields(orc):
    rng = np.random.default_rng(8)
    n = 40000
    sym = np.minimum(rng.geometric(0.3, n) - 1, 255).astype(np.uint8)
    freq = orc.aw_symbol_stat(sym, 256, 64)
    code, lens = orc.rp_system_codebook(freq, 64, 0, 16)
    payload, off = orc.rp_encode(sym, code, lens)
    blob = orc.container_pack(n, orc.LAYOUT_U8, payload, off, code=code, lens=lens, mode=1,
                              entropy_q16=0x1234)
    raw = blob.tobytes()
    assert raw[:8] == b"HUFHIP1\0"
    ver, hdr, layout, kind, chunk, lmax, mode, ent = struct.unpack_from("<8I", raw, 8)
    assert (ver, hdr, layout, kind, chunk, lmax, mode, ent) == (1, 128, 0, 0, 16384, 16, 1, 0x1234)
    n_, nch, words, o_cb, o_off, o_side, o_pay, total = struct.unpack_from("<8Q", raw, 40)
    assert (n_, nch, words, o_side, total) == (n, 3, payload.size, 0, len(raw))
    assert raw[104:128] == bytes(24)
    assert np.array_equal(np.frombuffer(raw, "<u2", 256, o_cb), code)
    assert np.array_equal(np.frombuffer(raw, "u1", 256, o_cb + 512), lens)
    assert np.array_equal(np.frombuffer(raw, "<u4", nch + 1, o_off), off)
    assert np.array_equal(np.frombuffer(raw, "<u4", words, o_pay), payload)
    # the stored stream decodes
    out = orc.rp_decode(np.frombuffer(raw, "<u4", words, o_pay),
                        np.frombuffer(raw, "<u4", nch + 1, o_off), n, code, lens)
    assert np.array_equal(out, sym)


def test_placed_plan_matches_oracle_sizes(lib, orc):
    """Wire format version 2 (slotted: a words array; paged: no codebook section when the pages
    carry records, the stream's one codebook when they do not)."""
    from huffman_amd import _lib as L
    ol = orc.lib()
    ol.orc_container2_bytes.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint64]
    ol.orc_container2_bytes.restype = C.c_uint64
    info = L.hf_container_info()
    for n in (0, 1, 16384, 16385, 10**6 + 7):
        for layout in (0, 1):
            for kind, placement in ((0, 0), (1, 0), (0, 1), (1, 1), (1, 2), (0, 2)):
                for words in (0, 3, 4, 12345):
                    assert lib.hf_container_plan_placed(n, layout, kind, placement, words,
                                                        C.byref(info)) == 0
                    assert info.total_bytes == ol.orc_container2_bytes(n, layout, kind, placement, words)
                    assert info.version == (1 if placement == 0 else 2)
                    assert (info.off_chunk_words != 0) == (placement == 1)
                    assert (info.off_codebook == 0) == (placement == 2 and kind == 1)
                    for off in (info.off_codebook, info.off_chunk_off, info.off_chunk_words,
                                info.off_side, info.off_payload, info.total_bytes):
                        assert off % 16 == 0
    # unknown placements are refused
    assert lib.hf_container_plan_placed(10, 0, 1, 3, 1, C.byref(info)) == L.HF_ERR_ARG


def test_oracle_container_v2_fields(orc):
    rng = np.random.default_rng(9)
    n = 3 * 16384 + 5
    sym = np.minimum(rng.geometric(0.2, n) - 1, 255).astype(np.uint8)
    payload, off, code, lens = orc.rp_encode_pages(sym, 0, 1)
    raw = orc.container_pack(n, orc.LAYOUT_U8, payload, off, placement=2).tobytes()
    ver, hdr, layout, kind = struct.unpack_from("<4I", raw, 8)
    assert (ver, hdr, layout, kind) == (2, 128, 0, 1)
    n_, nch, words, o_cb, o_off, o_side, o_pay, total = struct.unpack_from("<8Q", raw, 40)
    assert (n_, nch, words, o_cb, o_side, total) == (n, 4, payload.size, 0, 0, len(raw))
    assert struct.unpack_from("<IIQ", raw, 104) == (2, 0, 0) and raw[120:128] == bytes(8)
    assert np.array_equal(np.frombuffer(raw, "<u4", nch + 1, o_off), off)
    back = orc.rp_decode_pages(np.frombuffer(raw, "<u4", words, o_pay),
                               np.frombuffer(raw, "<u4", nch + 1, o_off), n)
    assert np.array_equal(back, sym)
    # slotted: the words array sits between the offsets and the payload
    p2, slot_off, w2, c2, l2 = orc.rp_encode_blocks_slotted(sym)
    r2s = np.zeros((nch, 256), np.uint8)
    raw = orc.container_pack(n, orc.LAYOUT_U8, p2, slot_off, rank2sym=r2s, len_by_rank=r2s,
                             modes=[1] * nch, entropies=[0] * nch, placement=1,
                             chunk_words=w2).tobytes()
    o_words = struct.unpack_from("<Q", raw, 112)[0]
    assert struct.unpack_from("<I", raw, 104)[0] == 1 and o_words % 16 == 0
    assert np.array_equal(np.frombuffer(raw, "<u4", nch, o_words), w2)


def test_oracle_pages_without_records(orc):
    """Pages under one codebook: 49 words plus the strings each, round trips at ragged sizes, the
    container with the stream's codebook section, and what damage the decoder reports."""
    rng = np.random.default_rng(10)
    for n in (1, 63, 16384, 16385, 3 * 16384 + 5):
        sym = np.minimum(rng.geometric(0.15, n) - 1, 255).astype(np.uint8)
        code, lens = orc.rp_system_codebook(orc.aw_symbol_stat(sym, 256, 64), 64, 0, 16)
        payload, off = orc.rp_encode_pages_shared(sym, code, lens)
        nch = (n + 16383) // 16384
        for c in range(nch):
            bits = int(lens[sym[c * 16384:(c + 1) * 16384]].astype(np.int64).sum())
            assert int(off[c + 1]) - int(off[c]) == orc.shared_page_words(bits) == 49 + (bits + 31) // 32
            assert int(payload[off[c]]) == (bits << 10) | (1 << 29)
        assert np.array_equal(orc.rp_decode_pages_shared(payload, off, n, code, lens), sym)
    raw = orc.container_pack(n, orc.LAYOUT_U8, payload, off, code=code, lens=lens, mode=1,
                             entropy_q16=7, placement=2).tobytes()
    ver, hdr, layout, kind = struct.unpack_from("<4I", raw, 8)
    assert (ver, layout, kind) == (2, 0, 0) and struct.unpack_from("<I", raw, 104)[0] == 2
    o_cb, o_off, o_side, o_pay = struct.unpack_from("<4Q", raw, 64)
    assert o_cb == 128 and o_off == 128 + 768 and o_side == 0
    assert np.array_equal(np.frombuffer(raw, "<u2", 256, o_cb), code)
    assert np.array_equal(np.frombuffer(raw, "u1", 256, o_cb + 512), lens)
    back = orc.rp_decode_pages_shared(np.frombuffer(raw, "<u4", payload.size, o_pay),
                                      np.frombuffer(raw, "<u4", nch + 1, o_off), n, code, lens)
    assert np.array_equal(back, sym)
    # a page that claims a record, a size that disagrees with the offsets, region lengths that do
    # not add up: all refused
    for word, flip in ((int(off[1]), 1 << 29), (int(off[1]), 1 << 12), (int(off[2]) + 3, 0x7)):
        hurt = payload.copy()
        hurt[word] ^= np.uint32(flip)
        with pytest.raises(ValueError):
            orc.rp_decode_pages_shared(hurt, off, n, code, lens)
