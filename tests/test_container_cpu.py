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
