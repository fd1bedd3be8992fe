"""Paged stream (format version 2: hf_encode_pages / hf_decode_pages) against the oracle's
restatement: page offsets, every payload word (records, region lengths, strings), the records read
back, round trips, decode-on-read into consecutive and named pages, refusals and damage.  The
reference has neither a packer nor a decoder (top_encoder.scala:17-21, decoder/top_decoder.scala:6-10);
the codebooks in the records are the per-block codebooks of the oracle (parity unpinned beyond it)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CH = 16384


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def make(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "geom":
        return np.minimum(rng.geometric(0.08, n) - 1, 255).astype(np.uint8)
    if kind == "uniform":
        return rng.integers(0, 256, n).astype(np.uint8)
    if kind == "const":
        return np.full(n, 7, np.uint8)
    if kind == "normal":
        return np.clip(rng.normal(128, 9, n), 0, 255).astype(np.uint8)
    if kind == "mixed":  # another distribution in every page
        out = np.empty(n, np.uint8)
        for c in range(0, n, CH):
            m = min(CH, n - c)
            p = [0.5, 0.08, 0.02, 0.9][(c // CH) % 4]
            out[c:c + m] = np.minimum(rng.geometric(p, m) - 1, 255) + (c // CH) * 3 % 200
        return out
    raise ValueError(kind)


def compare(s, payload, off, where=""):
    nch = s.nchunks
    g_off = s.chunk_off[: nch + 1].cpu().numpy().view(np.uint32)
    assert np.array_equal(g_off, off), f"page offsets differ {where}"
    assert s.total_words == payload.size
    got = s.payload[: payload.size].cpu().numpy().view(np.uint32)
    if not np.array_equal(got, payload):
        bad = int(np.flatnonzero(got != payload)[0])
        page = int(np.searchsorted(off, bad, side="right") - 1)
        raise AssertionError(f"payload differs {where}: word {bad} (page {page}, word "
                             f"{bad - int(off[page])} of {int(off[page + 1] - off[page])}): "
                             f"{got[bad]:#010x} != {payload[bad]:#010x}")


@pytest.mark.parametrize("kind,n", [
    ("geom", 1), ("geom", 511), ("geom", CH), ("geom", CH + 1), ("normal", 5 * CH),
    ("mixed", 9 * CH + 77), ("uniform", 4 * CH), ("const", 3 * CH + 5), ("mixed", 300 * CH + 4097),
])
def test_paged_stream_matches_the_oracle(hf, orc, kind, n):
    sym = make(kind, n, n % 1000 + 5)
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, paged=True)
    payload, off, code, lens = orc.rp_encode_pages(sym, 0, 1)
    compare(s, payload, off, f"({kind}, {n})")
    for c in {0, s.nchunks // 2, s.nchunks - 1}:
        gc, gl, mode, ent = sys_.block_codebook(s, c)
        assert mode == 1 and np.array_equal(gc, code[c]) and np.array_equal(gl, lens[c])
        assert ent == orc.rp_page_codebook(payload, off, c)[3]
    assert torch.equal(dec.decode(s), t)
    assert np.array_equal(orc.rp_decode_pages(payload, off, n), sym)


@pytest.mark.parametrize("mode", ["bypass", "auto", "best"])
def test_paged_other_modes_and_threshold(hf, orc, mode):
    n = 7 * CH + 1000
    sym = np.concatenate([make("uniform", 3 * CH, 1), make("geom", n - 3 * CH, 2)])
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    m = {"bypass": hf.HF_MODE_BYPASS, "auto": hf.HF_MODE_AUTO, "best": hf.HF_MODE_BEST}[mode]
    for thr in (0, 9):
        s = sys_.encode_blocks(t, mode=m, min_freq_threshold=thr, paged=True)
        payload, off, _, _ = orc.rp_encode_pages(sym, thr, m)
        compare(s, payload, off, f"({mode}, thr {thr})")
        assert torch.equal(hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s), t)


def test_paged_bf16_pages(hf, orc):
    n = 37 * CH + 1234
    rng = np.random.default_rng(3)
    w = (rng.standard_normal(n) * 0.02).astype(np.float32)
    bits = (w.view(np.uint32) >> 16).astype(np.uint16)
    t = to_dev(bits).view(torch.bfloat16)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, paged=True)
    esym, eside = orc.symbolize(bits, orc.LAYOUT_BF16_EXP)
    payload, off, _, _ = orc.rp_encode_pages(esym, 0, 1)
    compare(s, payload, off, "(bf16)")
    assert np.array_equal(s.side[:n].cpu().numpy(), eside)
    assert torch.equal(dec.decode(s).view(torch.int16), t.view(torch.int16))
    ids = to_dev(np.array([5, 0, 36, 5], np.int32))
    got = dec.decode_blocks(s, ids).view(torch.int16).cpu().numpy().view(np.uint16)
    want = np.concatenate([bits[i * CH:(i + 1) * CH] for i in (5, 0, 36, 5)])
    assert np.array_equal(got, want)


def test_paged_sub_field_layout_round_trip(hf, orc):
    n = 11 * CH + 9
    rng = np.random.default_rng(9)
    w = (rng.standard_normal(n) * 0.05).astype(np.float16)
    t = to_dev(w)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, layout=hf.HF_LAYOUT_FP16_EXP, paged=True)
    esym, eside = orc.symbolize(w.view(np.uint16), orc.LAYOUT_FP16_EXP)
    payload, off, _, _ = orc.rp_encode_pages(esym, 0, 1)
    compare(s, payload, off, "(fp16 exponent fields)")
    assert np.array_equal(s.side.cpu().numpy(), eside)
    assert torch.equal(dec.decode(s).view(torch.int16), t.view(torch.int16))


def test_paged_decode_on_read(hf, orc):
    pages = 40
    n = pages * CH
    sym = make("mixed", n, 17)
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode_blocks(t, paged=True)  # AUTO: some pages may be bypassed
    ids = np.array([39, 0, 17, 17, 5], np.int32)
    got = dec.decode_blocks(s, to_dev(ids))
    want = np.concatenate([sym[i * CH:(i + 1) * CH] for i in ids])
    assert np.array_equal(got.cpu().numpy(), want)
    pool = torch.zeros(8 * CH, dtype=torch.uint8, device=DEV)
    slots = np.array([7, 2, 0, 4, 5], np.int32)
    dec.decode_blocks_to(s, to_dev(ids), to_dev(slots), pool)
    p = pool.cpu().numpy()
    for i, sl in zip(ids, slots):
        assert np.array_equal(p[sl * CH:(sl + 1) * CH], sym[i * CH:(i + 1) * CH])
    assert not p[CH: 2 * CH].any()
    # only payload and offsets are needed: a stream object without its codebook records decodes
    import dataclasses
    bare = dataclasses.replace(s, chunk_codebooks=None, chunk_freq=None)
    assert torch.equal(dec.decode(bare), t)


def test_paged_pages_above_the_image_go_through_global_memory(hf, orc):
    """Huffman forced on pages it does not compress, with a threshold that throws most counts
    away: such pages take more than 8.1 bits per symbol and leave the LDS image."""
    n = 5 * CH + 300
    rng = np.random.default_rng(77)
    sym = rng.integers(0, 256, n).astype(np.uint8)
    heavy = rng.random(n) < 0.2  # 16 symbols keep their counts, 240 frequent ones lose theirs
    sym[heavy] = rng.integers(0, 16, int(heavy.sum())).astype(np.uint8)
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, min_freq_threshold=100, paged=True)
    payload, off, _, _ = orc.rp_encode_pages(sym, 100, 1)
    assert int(np.diff(off.astype(np.int64)).max()) > 4224, "the test needs an oversized page"
    compare(s, payload, off, "(oversized pages)")
    assert torch.equal(hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s), t)


def test_paged_refusals_and_damage(hf, orc):
    n = 6 * CH
    sym = make("geom", n, 23)
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    lib, h = hf.L.lib(), sys_.ctx.handle
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, paged=True)
    P = lambda x: C.c_void_p(x.data_ptr())
    total = C.c_uint64(0)
    off = torch.empty(8, dtype=torch.int32, device=DEV)
    # a payload buffer the pages do not fit: reported, nothing written
    small = torch.full((s.total_words - 1,), 0x5A5A5A5A, dtype=torch.int32, device=DEV)
    rc = lib.hf_encode_pages(h, P(t), n, 0, P(s.chunk_codebooks), P(s.chunk_freq), None, P(small),
                             small.numel(), P(off), C.byref(total))
    assert rc == hf.L.HF_ERR_OVERFLOW and bool((small == 0x5A5A5A5A).all())
    # histograms of other data: the pages do not have the sizes their places were cut for
    rare = int(np.bincount(sym, minlength=256).argmin())
    other = to_dev(np.full(n, rare, np.uint8))
    cap = s.total_words
    buf = torch.full((cap + 4096,), 0x5A5A5A5A, dtype=torch.int32, device=DEV)
    rc = lib.hf_encode_pages(h, P(other), n, 0, P(s.chunk_codebooks), P(s.chunk_freq), None, P(buf),
                             cap, P(off), C.byref(total))
    assert rc == hf.L.HF_ERR_CORRUPT
    assert bool((buf == 0x5A5A5A5A).all())
    for fill in (0x7FFFFFFF, 0):
        wild = torch.full_like(s.chunk_freq, fill)
        buf.fill_(0x5A5A5A5A)
        rc = lib.hf_encode_pages(h, P(t), n, 0, P(s.chunk_codebooks), P(wild), None, P(buf), cap,
                                 P(off), C.byref(total))
        assert rc in (hf.L.HF_ERR_CORRUPT, hf.L.HF_ERR_OVERFLOW)
        assert bool((buf[cap:] == 0x5A5A5A5A).all())
    # damage: a flipped bit in page 2's strings, record, region lengths and header in turn; the
    # other pages decode all the same and nothing is written outside the output
    o = s.chunk_off[:7].cpu().numpy().view(np.uint32)
    head = int(s.payload[int(o[2])].item()) & 0x1FF
    hdr = 10 + (head + 3) // 4
    out = torch.empty(n + CH, dtype=torch.uint8, device=DEV)
    for word, bit, must_report in ((hdr + 48 + 100, 5, False), (3, 1, True), (hdr + 7, 3, True),
                                   (0, 12, True), (10, 0, None)):
        pay = s.payload.clone()
        pay[int(o[2]) + word] ^= (1 << bit)
        out.fill_(0xEE)
        rc = lib.hf_decode_pages(h, P(pay), pay.numel(), P(s.chunk_off), None, n, 0, None, None, 0,
                                 P(out), 0, 1)
        if must_report:
            assert rc == hf.L.HF_ERR_CORRUPT, (word, bit, rc)
        else:
            assert rc in (0, hf.L.HF_ERR_CORRUPT)
        got = out.cpu().numpy()
        assert np.array_equal(got[: 2 * CH], sym[: 2 * CH]) and np.array_equal(got[3 * CH:n], sym[3 * CH:])
        assert (got[n:] == 0xEE).all()
    # offsets that run backwards or past the buffer, ids and slots out of range
    bad = s.chunk_off.clone()
    bad[3] = int(o[2]) - 1
    rc = lib.hf_decode_pages(h, P(s.payload), s.payload.numel(), P(bad), None, n, 0, None, None, 0,
                             P(out), 0, 1)
    assert rc == hf.L.HF_ERR_CORRUPT
    ids = to_dev(np.array([1, 6, 2], np.int32))
    slots = to_dev(np.array([0, 1, 9], np.int32))
    pool = torch.zeros(4 * CH, dtype=torch.uint8, device=DEV)
    rc = lib.hf_decode_pages(h, P(s.payload), s.payload.numel(), P(s.chunk_off), None, n, 0, P(ids),
                             P(slots), 3, P(pool), 4, 1)
    assert rc == hf.L.HF_ERR_CORRUPT
    assert np.array_equal(pool[:CH].cpu().numpy(), sym[CH:2 * CH]) and not bool(pool[CH:].any())
    assert lib.hf_decode_pages(h, P(s.payload), s.payload.numel(), P(s.chunk_off), None, n, 0, None,
                               P(slots), 3, P(pool), 4, 1) == hf.L.HF_ERR_ARG
    # the context is usable afterwards
    s2 = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, paged=True)
    assert s2.total_words == s.total_words
    assert torch.equal(hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s2), t)


def test_paged_empty_input(hf, orc):
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    t = torch.empty(0, dtype=torch.uint8, device=DEV)
    s = sys_.encode_blocks(t, paged=True)
    assert s.total_words == 0 and int(s.chunk_off[0].item()) == 0
    assert dec.decode(s).numel() == 0


@pytest.mark.parametrize("kind,n", [("geom", 0), ("geom", 7), ("mixed", 9 * CH + 77), ("normal", 64 * CH)])
def test_paged_container_round_trip(hf, orc, kind, n):
    """Wire format version 2: header, page offsets and payload, nothing else -- the pages carry their
    records -- byte for byte the oracle's; a stream opened from the blob decodes in place."""
    sym = make(kind, n, 41)
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_BEST, paged=True)
    payload, off, _, _ = orc.rp_encode_pages(sym, 0, 3)
    blob = s.to_container(sys_.ctx)
    assert blob.numel() == s.container_bytes()
    exp = orc.container_pack(n, orc.LAYOUT_U8, payload, off, placement=2)
    assert np.array_equal(blob.cpu().numpy(), exp)
    info = hf.L.hf_container_info()
    assert hf.L.lib().hf_container_open(sys_.ctx.handle, C.c_void_p(blob.data_ptr()), blob.numel(),
                                        C.byref(info)) == 0
    assert (info.version, info.placement, info.off_codebook, info.off_chunk_words) == (2, 2, 0, 0)
    s2 = hf.EncodedStream.from_container(blob, ctx=sys_.ctx)
    assert s2.paged and s2.chunk_codebooks is None and s2.total_words == payload.size
    assert torch.equal(dec.decode(s2), t)
    if n >= 2 * CH:
        ids = to_dev(np.array([s.nchunks - 2, 0], np.int32))
        got = dec.decode_blocks(s2, ids).cpu().numpy()
        assert np.array_equal(got[:CH], sym[(s.nchunks - 2) * CH:(s.nchunks - 1) * CH])
        assert np.array_equal(got[CH:], sym[:CH])
        # a version-2 header with a damaged placement or words offset is refused
        for pos, val in ((104, 3), (104, 0), (112, 16), (108, 1)):
            bad = blob.clone()
            bad[pos] = val
            assert hf.L.lib().hf_container_open(sys_.ctx.handle, C.c_void_p(bad.data_ptr()),
                                                bad.numel(), C.byref(info)) == hf.L.HF_ERR_CORRUPT


def test_paged_bf16_container(hf, orc):
    n = 5 * CH + 100
    rng = np.random.default_rng(5)
    bits = ((rng.standard_normal(n) * 0.02).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    t = to_dev(bits).view(torch.bfloat16)
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, paged=True)
    esym, eside = orc.symbolize(bits, orc.LAYOUT_BF16_EXP)
    payload, off, _, _ = orc.rp_encode_pages(esym, 0, 1)
    blob = s.to_container(sys_.ctx)
    exp = orc.container_pack(n, orc.LAYOUT_BF16_EXP, payload, off, side=eside, placement=2)
    assert np.array_equal(blob.cpu().numpy(), exp)
    s2 = hf.EncodedStream.from_container(blob, ctx=sys_.ctx)
    out = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s2)
    assert torch.equal(out.view(torch.int16), t.view(torch.int16))
