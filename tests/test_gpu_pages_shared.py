"""Paged stream with one codebook for the whole stream (hf_encode_pages_shared /
hf_decode_pages_shared: pages without records) against the oracle's restatement: page offsets,
every payload word, round trips, decode-on-read, the container, refusals.  The codebook is the
stream-wide one of the oracle (orc_rp_system_codebook on the stream's counts); the stored form is
this build's (the reference has none: top_encoder.scala:17-21), parity unpinned beyond the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import gaussian_bf16
from test_gpu_pages import CH, compare, make, to_dev

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def oracle_stream(orc, sym, mode=1, thr=0):
    freq = orc.aw_symbol_stat(sym, 256, 64)
    if mode == 0:
        code, lens = np.arange(256, dtype=np.uint16), np.full(256, 8, np.uint8)
    else:
        code, lens = orc.rp_system_codebook(freq, 64, thr, 16)
    payload, off = orc.rp_encode_pages_shared(sym, code, lens)
    return payload, off, code, lens


# counts: the pages placed beforehand from per-page counts, or while they are encoded
@pytest.mark.parametrize("counts", [False, True])
@pytest.mark.parametrize("kind,n", [
    ("geom", 1), ("geom", 511), ("geom", CH), ("geom", CH + 1), ("normal", 5 * CH),
    ("mixed", 9 * CH + 77), ("uniform", 4 * CH), ("const", 3 * CH + 5), ("mixed", 300 * CH + 4097),
])
def test_shared_pages_match_the_oracle(hf, orc, kind, n, counts):
    sym = make(kind, n, n % 1000 + 5)
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, paged=True, page_counts=counts)
    assert s.paged and s.codebook is not None and s.chunk_codebooks is None
    assert (s.chunk_freq is not None) == counts
    payload, off, code, lens = oracle_stream(orc, sym)
    compare(s, payload, off, f"({kind}, {n})")
    gc, gl, mode, _ = s.codebook.download()
    assert mode == 1 and np.array_equal(gc, code) and np.array_equal(gl, lens)
    assert torch.equal(dec.decode(s), t)
    assert np.array_equal(orc.rp_decode_pages_shared(payload, off, n, code, lens), sym)
    # every page is 49 words plus its strings: nothing else is stored
    bits = [int(lens[sym[c:c + CH]].astype(np.int64).sum()) for c in range(0, n, CH)]
    assert list(np.diff(off.astype(np.int64))) == [49 + (b + 31) // 32 for b in bits]


def test_every_page_count_up_to_nineteen(hf, orc):
    """The page encoders hand every eighth of the pages to one group of workgroups (the grid is
    rounded up to a multiple of eight): every page count from one to nineteen, below, at and above
    one and two multiples of eight, with a ragged last page, with and without records, word for
    word against the oracle."""
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    for pages in range(1, 20):
        n = pages * CH - (pages * 997) % CH // 2
        sym = make("mixed", n, 40 + pages)
        t = to_dev(sym)
        s = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, paged=True)
        assert s.nchunks == pages
        payload, off, _, _ = oracle_stream(orc, sym)
        compare(s, payload, off, f"({pages} pages without records)")
        assert torch.equal(dec.decode(s), t)
        s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, paged=True)
        payload, off, _, _ = orc.rp_encode_pages(sym, 0, 1)
        compare(s, payload, off, f"({pages} pages with records)")
        assert torch.equal(dec.decode(s), t)


def test_shared_pages_bypass_and_threshold(hf, orc):
    n = 6 * CH + 321
    sym = np.concatenate([make("uniform", 2 * CH, 1), make("geom", n - 2 * CH, 2)])
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode(t, mode=hf.HF_MODE_BYPASS, paged=True)
    payload, off, _, _ = oracle_stream(orc, sym, mode=0)
    compare(s, payload, off, "(bypass)")
    assert torch.equal(dec.decode(s), t)
    for thr in (3, 40):   # symbols below the threshold share the long codes
        s = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, min_freq_threshold=thr, paged=True)
        payload, off, _, _ = oracle_stream(orc, sym, thr=thr)
        compare(s, payload, off, f"(thr {thr})")
        assert torch.equal(dec.decode(s), t)


@pytest.mark.parametrize("counts", [False, True])
def test_shared_pages_bf16_and_decode_on_read(hf, orc, counts):
    n = 37 * CH + 1234
    bits = gaussian_bf16(n, seed=3)
    t = to_dev(bits.view(np.int16)).view(torch.bfloat16)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, paged=True, page_counts=counts)
    esym, eside = orc.symbolize(bits, orc.LAYOUT_BF16_EXP)
    payload, off, _, _ = oracle_stream(orc, esym)
    compare(s, payload, off, "(bf16)")
    assert np.array_equal(s.side[:n].cpu().numpy(), eside)
    assert torch.equal(dec.decode(s).view(torch.int16), t.view(torch.int16))
    ids = np.array([5, 0, 36, 5], np.int32)
    got = dec.decode_blocks(s, to_dev(ids)).view(torch.int16).cpu().numpy().view(np.uint16)
    assert np.array_equal(got, np.concatenate([bits[i * CH:(i + 1) * CH] for i in ids]))
    pool = torch.zeros(6 * CH, dtype=torch.bfloat16, device=DEV)
    slots = np.array([4, 1, 0, 5], np.int32)
    dec.decode_blocks_to(s, to_dev(ids), to_dev(slots), pool)
    p = pool.view(torch.int16).cpu().numpy().view(np.uint16)
    for i, sl in zip(ids, slots):
        assert np.array_equal(p[sl * CH:(sl + 1) * CH], bits[i * CH:(i + 1) * CH])
    assert not p[2 * CH:4 * CH].any()
    # the stored size: exponent payload without look-ahead words
    assert 32.0 * s.total_words / n < 32.0 * sys_.encode(t, mode=hf.HF_MODE_HUFFMAN).total_words / n


def test_shared_pages_with_a_foreign_codebook(hf, orc):
    # a codebook from other data (a global table, another shard's counts): pages of any size up to
    # 16 bits per symbol, the large ones taking the path through global memory
    rng = np.random.default_rng(8)
    other = np.minimum(rng.geometric(0.3, 40 * CH) - 1, 255).astype(np.uint8)
    n = 5 * CH + 100
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    cb = sys_.codebook(sys_.histogram(to_dev(other)), hf.HF_MODE_HUFFMAN)
    code, lens, _, _ = cb.download()
    rare = int(np.argmax(lens))
    assert lens[rare] >= 12
    sym = np.concatenate([make("geom", 2 * CH, 4), np.full(CH, rare, np.uint8),
                          rng.integers(128, 256, n - 3 * CH).astype(np.uint8)])
    t = to_dev(sym)
    payload, off = orc.rp_encode_pages_shared(sym, code, lens)
    assert int(np.diff(off.astype(np.int64)).max()) == 49 + CH * int(lens[rare]) // 32 > 4224
    for counts in (None, sys_.histogram_pages(t)[0]):
        s = sys_.encode_with(t, cb, paged=True, chunk_freq=counts)
        compare(s, payload, off, "(foreign codebook)")
        assert torch.equal(dec.decode(s), t)
    out = dec.decode_blocks(s, to_dev(np.array([2, 1], np.int32)))
    assert np.array_equal(out.cpu().numpy(), np.concatenate([sym[2 * CH:3 * CH], sym[CH:2 * CH]]))


def test_shared_pages_sub_field_layout(hf, orc):
    n = 11 * CH + 9
    w = (np.random.default_rng(9).standard_normal(n) * 0.05).astype(np.float16)
    t = to_dev(w)
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, layout=hf.HF_LAYOUT_FP16_EXP, paged=True)
    esym, eside = orc.symbolize(w.view(np.uint16), orc.LAYOUT_FP16_EXP)
    payload, off, _, _ = oracle_stream(orc, esym)
    compare(s, payload, off, "(fp16 exponent fields)")
    assert np.array_equal(s.side.cpu().numpy(), eside)
    out = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s)
    assert torch.equal(out.view(torch.int16), t.view(torch.int16))


@pytest.mark.parametrize("kind", ["u8", "bf16"])
def test_shared_pages_container_round_trip(hf, orc, kind):
    n = 13 * CH + 777
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    if kind == "bf16":
        bits = gaussian_bf16(n, seed=21)
        t = to_dev(bits.view(np.int16)).view(torch.bfloat16)
        sym, side = orc.symbolize(bits, orc.LAYOUT_BF16_EXP)
        lay = orc.LAYOUT_BF16_EXP
    else:
        sym, side, lay = make("mixed", n, 2), None, orc.LAYOUT_U8
        t = to_dev(sym)
    s = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, paged=True)
    blob = s.to_container(sys_.ctx)
    assert blob.numel() == s.container_bytes()
    payload, off, code, lens = oracle_stream(orc, sym)
    _, _, mode, ent = s.codebook.download()
    want = orc.container_pack(n, lay, payload, off, side=side, code=code, lens=lens, mode=mode,
                              entropy_q16=ent, placement=2)
    assert np.array_equal(blob.cpu().numpy(), want)
    back = hf.EncodedStream.from_container(blob, ctx=sys_.ctx)
    assert back.paged and back.codebook is not None and back.placement == hf.L.HF_PLACEMENT_PAGED
    assert torch.equal(dec.decode(back).view(torch.uint8), t.view(torch.uint8))
    ids = to_dev(np.array([12, 3], np.int32))
    got = dec.decode_blocks(back, ids).view(torch.uint8)
    el = t.element_size() * CH
    raw = t.view(torch.uint8)
    assert torch.equal(got, torch.cat([raw[12 * el:13 * el], raw[3 * el:4 * el]]))


def test_shared_pages_refusals_and_damage(hf, orc):
    L = hf.L
    lib = L.lib()
    n = 4 * CH
    sym = make("geom", n, 6)
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    h = sys_.ctx.handle
    s = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, paged=True)
    rec = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, paged=True)
    out = torch.zeros(n, dtype=torch.uint8, device=DEV)
    P = lambda x: C.c_void_p(x.data_ptr())

    def dec_shared(stream, cb_ptr):
        return lib.hf_decode_pages_shared(h, P(stream.payload), stream.payload.numel(),
                                          P(stream.chunk_off), None, n, hf.HF_LAYOUT_U8, cb_ptr, None,
                                          None, 0, P(out), 0, 1)
    assert dec_shared(s, None) == L.HF_ERR_ARG
    assert dec_shared(s, C.c_void_p(s.codebook.ptr)) == 0 and torch.equal(out, t)
    # pages with records are not pages without, and the other way round
    assert dec_shared(rec, C.c_void_p(s.codebook.ptr)) == L.HF_ERR_CORRUPT
    assert lib.hf_decode_pages(h, P(s.payload), s.payload.numel(), P(s.chunk_off), None, n,
                               hf.HF_LAYOUT_U8, None, None, 0, P(out), 0, 1) == L.HF_ERR_CORRUPT
    sys_.ctx.flush()
    # a table that is not the canonical code of its lengths (1 | 00 | 01 instead of 0 | 10 | 11)
    # cannot code pages of this kind: every page is reported
    mw = hf.MemWrapper(ctx=sys_.ctx)
    for i in range(256):
        c, l = ((1, 1), (0, 2), (1, 2))[i] if i < 3 else (0, 0)
        mw.write(i, c, l)
    assert dec_shared(s, C.c_void_p(mw.to_device().ptr)) == L.HF_ERR_CORRUPT
    sys_.ctx.flush()
    # damage: a page's size word, a region length, an offset
    for word, what in ((int(s.chunk_off[1].item()), "header"), (int(s.chunk_off[2].item()) + 7, "region")):
        hurt = hf.EncodedStream(**{**s.__dict__, "payload": s.payload.clone()})
        hurt.payload[word] ^= 0x00040000
        with pytest.raises(RuntimeError):
            dec.decode(hurt)
        sys_.ctx.flush()
    hurt = hf.EncodedStream(**{**s.__dict__, "chunk_off": s.chunk_off.clone()})
    hurt.chunk_off[2] += 1
    with pytest.raises(RuntimeError):
        dec.decode(hurt)
    sys_.ctx.flush()
    # a payload buffer that is too small is reported, nothing written behind it
    small = torch.full((s.total_words - 1 + 8,), -1, dtype=torch.int32, device=DEV)
    freq = sys_.histogram_pages(t)[0]
    rc = lib.hf_encode_pages_shared(h, P(t), n, hf.HF_LAYOUT_U8, C.c_void_p(s.codebook.ptr), P(freq),
                                    None, P(small), s.total_words - 1, P(torch.empty_like(s.chunk_off)),
                                    C.byref(C.c_uint64(0)))
    assert rc == L.HF_ERR_OVERFLOW and bool((small == -1).all())
    sys_.ctx.flush()
    assert lib.hf_encode_pages_shared(h, P(t), n, hf.HF_LAYOUT_U8, None, P(freq), None, P(small), 8,
                                      P(s.chunk_off), None) == L.HF_ERR_ARG
    # placed while they are encoded, the pages in front of the one that does not fit are written,
    # nothing behind the buffer's stated size is, and the call reports the same
    small.fill_(-1)
    rc = lib.hf_encode_pages_shared(h, P(t), n, hf.HF_LAYOUT_U8, C.c_void_p(s.codebook.ptr), None,
                                    None, P(small), s.total_words - 1, P(torch.empty_like(s.chunk_off)),
                                    C.byref(C.c_uint64(0)))
    assert rc == L.HF_ERR_OVERFLOW and bool((small[s.total_words - 1:] == -1).all())
    last = int(s.chunk_off[3].item())
    assert torch.equal(small[:last], s.payload[:last]) and bool((small[last:] == -1).all())
    sys_.ctx.flush()


@pytest.mark.parametrize("layout", ["u8", "bf16"])
def test_pages_placed_while_encoded_equal_pages_placed_from_counts(hf, orc, layout):
    # many more pages than the GPU holds workgroups (the walk back over the descriptors runs through
    # thousands of pages that finish in any order), twice on one context, the second call shorter
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    for n in (9000 * CH + 4321, 2500 * CH):
        if layout == "bf16":
            t = (torch.randn(n, device=DEV, generator=torch.Generator(DEV).manual_seed(n % 89)) * 0.02
                 ).to(torch.bfloat16)
        else:
            g = torch.Generator(DEV).manual_seed(n % 97)
            t = (torch.randn(n, device=DEV, generator=g) * 20 + 128).clamp(0, 255).to(torch.uint8)
        a = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, paged=True, page_counts=True)
        b = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, paged=True, page_counts=False)
        assert b.chunk_freq is None and a.total_words == b.total_words
        nch = a.nchunks
        assert torch.equal(a.chunk_off[:nch + 1], b.chunk_off[:nch + 1])
        assert torch.equal(a.payload[:a.total_words], b.payload[:b.total_words])
        if a.side is not None:
            assert torch.equal(a.side[:n], b.side[:n])
        assert torch.equal(dec.decode(b).view(torch.uint8), t.view(torch.uint8))


def test_shared_pages_empty_input(hf):
    sys_ = hf.EntropyAwareHuffmanSystem()
    t = torch.empty(0, dtype=torch.uint8, device=DEV)
    cb = sys_.codebook(torch.zeros(256, dtype=torch.int64, device=DEV), hf.HF_MODE_HUFFMAN)
    s = sys_.encode_with(t, cb, paged=True)
    assert s.total_words == 0 and int(s.chunk_off[0].item()) == 0
    assert hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s).numel() == 0
    blob = s.to_container(sys_.ctx)
    back = hf.EncodedStream.from_container(blob, ctx=sys_.ctx)
    assert back.n == 0 and back.paged


@pytest.mark.parametrize("layout", ["u8", "bf16"])
def test_page_counts_and_stream_totals_from_one_read(hf, orc, layout):
    # hf_symbol_stat_pages_dev against the oracle's SymbolStat per page and over the stream: sizes
    # around page boundaries, more pages than resident workgroups, totals ADDED into d_freq256,
    # and the call without the totals
    L = hf.L
    lib = L.lib()
    ctx = hf.Context(0)
    rng = np.random.default_rng(31)
    P = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
    for n in (0, 1, 7, CH - 1, CH, CH + 1, 5 * CH + 77, 3000 * CH + 12345):
        if layout == "bf16":
            bits = gaussian_bf16(max(n, 1), seed=n % 97)[:n]
            t = to_dev(bits.view(np.int16)) if n else torch.empty(0, dtype=torch.int16, device=DEV)
            sym, _ = orc.symbolize(bits, orc.LAYOUT_BF16_EXP) if n else (np.zeros(0, np.uint8), None)
            lay = hf.HF_LAYOUT_BF16_EXP
        else:
            sym = np.minimum(rng.geometric(0.05, n) - 1, 255).astype(np.uint8)
            t = to_dev(sym) if n else torch.empty(0, dtype=torch.uint8, device=DEV)
            lay = hf.HF_LAYOUT_U8
        nch = (n + CH - 1) // CH
        pages = torch.full((max(nch, 1) * 256,), -1, dtype=torch.int32, device=DEV)
        tot = torch.full((256,), 5, dtype=torch.int64, device=DEV)
        assert lib.hf_symbol_stat_pages_dev(ctx.handle, P(t) if n else None, n, lay, P(pages), P(tot)) == 0
        ctx.sync()
        want = np.zeros((max(nch, 1), 256), np.int64)
        for c in range(nch):
            want[c] = np.bincount(sym[c * CH:(c + 1) * CH], minlength=256)
        got = pages.cpu().numpy().reshape(-1, 256)
        assert np.array_equal(got[:nch], want[:nch]), (layout, n)
        assert np.array_equal(tot.cpu().numpy(), want[:nch].sum(0) + 5), (layout, n)
        if n:
            assert np.array_equal(want[:nch].sum(0).astype(np.uint64),
                                  np.asarray(orc.aw_symbol_stat(sym, 256, 64), np.uint64))
            again = torch.zeros_like(pages)
            assert lib.hf_symbol_stat_pages_dev(ctx.handle, P(t), n, lay, P(again), None) == 0
            ctx.sync()
            assert torch.equal(again[: nch * 256], pages[: nch * 256])
    assert lib.hf_symbol_stat_pages_dev(ctx.handle, None, 5, hf.HF_LAYOUT_U8, None, None) == L.HF_ERR_ARG
