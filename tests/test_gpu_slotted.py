"""Slotted placement of a per-block stream (hf_encode_chunks_slotted / hf_decode_chunks_slotted) and
of a stream with one codebook (hf_symbol_stat_fold_dev, hf_encode_slotted / hf_decode_slotted)
against the oracle's restatement: slot offsets, words per chunk, every payload word including the
zeroed slot tails, round trips, decode-on-read and the refusals.  The reference has no packer
(top.scala:60-67 stops at the codebook); the chunk words are those of the dense per-block stream."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


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
    if kind == "mixed":  # another distribution in every chunk
        out = np.empty(n, np.uint8)
        for c in range(0, n, 16384):
            m = min(16384, n - c)
            p = [0.5, 0.08, 0.02, 0.9][(c // 16384) % 4]
            out[c:c + m] = np.minimum(rng.geometric(p, m) - 1, 255) + (c // 16384) * 3 % 200
        return out
    raise ValueError(kind)


@pytest.mark.parametrize("kind,n", [
    ("geom", 1), ("geom", 511), ("geom", 16384), ("geom", 16385), ("mixed", 9 * 16384 + 77),
    ("uniform", 4 * 16384), ("const", 3 * 16384 + 5), ("mixed", 1100 * 16384 + 4097),
])
def test_slotted_stream_matches_the_oracle(hf, orc, kind, n):
    sym = make(kind, n, n % 1000 + 5)
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, slotted=True)
    payload, slot_off, words, code, lens = orc.rp_encode_blocks_slotted(sym)
    nch = s.nchunks
    assert s.total_words == payload.size == int(slot_off[nch])
    assert np.array_equal(s.chunk_off[: nch + 1].cpu().numpy().view(np.uint32), slot_off)
    assert np.array_equal(s.chunk_words[:nch].cpu().numpy().view(np.uint32), words)
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    # the chunks' words are the dense per-block stream's
    dense = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN)
    d_off = dense.chunk_off[: nch + 1].cpu().numpy().view(np.uint32)
    assert np.array_equal(np.diff(d_off.astype(np.int64)), words.astype(np.int64))
    dp = dense.payload.cpu().numpy().view(np.uint32)
    sp = s.payload.cpu().numpy().view(np.uint32)
    for c in {0, nch // 2, nch - 1}:
        assert np.array_equal(sp[slot_off[c]: slot_off[c] + words[c]], dp[d_off[c]: d_off[c + 1]])
    assert torch.equal(dec.decode(s), t)
    # stored form: wire format version 2 with the words array, byte for byte the oracle's
    blob = s.to_container(sys_.ctx)
    r2s = np.stack([orc.aw_symbol_sort(orc.aw_symbol_stat(sym[c * 16384:(c + 1) * 16384], 256, 64))[1]
                    for c in range(nch)]).astype(np.uint8) if nch <= 16 else None
    if r2s is not None:
        lbr = np.stack([lens[c][r2s[c]] for c in range(nch)])
        ents = [orc.rp_entropy_q16_u64(orc.aw_symbol_stat(sym[c * 16384:(c + 1) * 16384], 256, 64))
                for c in range(nch)]
        exp = orc.container_pack(n, orc.LAYOUT_U8, payload, slot_off, rank2sym=r2s, len_by_rank=lbr,
                                 modes=[1] * nch, entropies=ents, placement=1, chunk_words=words)
        assert np.array_equal(blob.cpu().numpy(), exp)
    s2 = hf.EncodedStream.from_container(blob, ctx=sys_.ctx)
    assert s2.chunk_words is not None and s2.total_words == s.total_words
    assert torch.equal(dec.decode(s2), t)


def test_slotted_bf16_blocks_round_trip(hf, orc):
    n = 37 * 16384 + 1234
    rng = np.random.default_rng(3)
    w = (rng.standard_normal(n) * 0.02).astype(np.float32)
    bits = (w.view(np.uint32) >> 16).astype(np.uint16)
    t = to_dev(bits).view(torch.bfloat16)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, slotted=True)
    esym, eside = orc.symbolize(bits, orc.LAYOUT_BF16_EXP)
    payload, slot_off, words, _, _ = orc.rp_encode_blocks_slotted(esym)
    assert s.total_words == payload.size
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    assert np.array_equal(s.side[:n].cpu().numpy(), eside)
    assert np.array_equal(s.chunk_words[: s.nchunks].cpu().numpy().view(np.uint32), words)
    out = dec.decode(s)
    assert torch.equal(out.view(torch.int16), t.view(torch.int16))


def test_slotted_sub_field_layout_round_trip(hf, orc):
    n = 11 * 16384 + 9
    rng = np.random.default_rng(9)
    w = (rng.standard_normal(n) * 0.05).astype(np.float16)
    t = to_dev(w)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, layout=hf.HF_LAYOUT_FP16_EXP, slotted=True)
    esym, eside = orc.symbolize(w.view(np.uint16), orc.LAYOUT_FP16_EXP)
    payload, _, _, _, _ = orc.rp_encode_blocks_slotted(esym)
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    assert np.array_equal(s.side.cpu().numpy(), eside)
    assert torch.equal(dec.decode(s).view(torch.int16), t.view(torch.int16))


def test_slotted_decode_on_read_and_paged(hf, orc):
    pages = 40
    n = pages * 16384
    sym = make("mixed", n, 17)
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode_blocks(t, slotted=True)  # AUTO: some blocks may be bypassed
    ids = np.array([39, 0, 17, 17, 5], np.int32)
    got = dec.decode_blocks(s, to_dev(ids))
    want = np.concatenate([sym[i * 16384:(i + 1) * 16384] for i in ids])
    assert np.array_equal(got.cpu().numpy(), want)
    pool = torch.zeros(8 * 16384, dtype=torch.uint8, device=DEV)
    slots = np.array([7, 2, 0, 4, 5], np.int32)
    dec.decode_blocks_to(s, to_dev(ids), to_dev(slots), pool)
    p = pool.cpu().numpy()
    for i, sl in zip(ids, slots):
        assert np.array_equal(p[sl * 16384:(sl + 1) * 16384], sym[i * 16384:(i + 1) * 16384])
    assert not p[16384: 2 * 16384].any()
    assert torch.equal(dec.decode(s), t)


def test_slotted_refusals(hf, orc):
    n = 6 * 16384
    sym = make("geom", n, 23)
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    ctx = sys_.ctx
    lib, h = hf.L.lib(), ctx.handle
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, slotted=True)
    P = lambda x: C.c_void_p(x.data_ptr())
    total = C.c_uint64(0)
    # a payload buffer the slots do not fit: reported, nothing written
    small = torch.full((s.total_words - 1,), 0x5A5A5A5A, dtype=torch.int32, device=DEV)
    off = torch.empty(8, dtype=torch.int32, device=DEV)
    cw = torch.empty(8, dtype=torch.int32, device=DEV)
    rc = lib.hf_encode_chunks_slotted(h, P(t), n, 0, P(s.chunk_codebooks), P(s.chunk_freq), None,
                                      P(small), small.numel(), P(off), P(cw), C.byref(total))
    assert rc == hf.L.HF_ERR_OVERFLOW
    assert bool((small == 0x5A5A5A5A).all())
    # histograms of other data: a chunk outgrows its slot, nothing lands outside the slots
    rare = int(np.bincount(sym, minlength=256).argmin())
    other = to_dev(np.full(n, rare, np.uint8))
    cap = s.total_words
    guard = 4096
    buf = torch.full((cap + guard,), 0x5A5A5A5A, dtype=torch.int32, device=DEV)
    rc = lib.hf_encode_chunks_slotted(h, P(other), n, 0, P(s.chunk_codebooks), P(s.chunk_freq),
                                      None, P(buf), cap, P(off), P(cw), C.byref(total))
    code, lens, _, _ = sys_.block_codebook(s, 0)
    if int(lens[rare]) * 16384 > int(s.chunk_words[0].item()) * 32 + 126 * 32:
        assert rc == hf.L.HF_ERR_CORRUPT
    assert bool((buf[cap:] == 0x5A5A5A5A).all())
    # counts that are nobody's (sums that wrap 32 bits, zeros): every slot is clamped or too small,
    # the call reports it and stays inside the buffer
    for fill in (0x7FFFFFFF, 0):
        wild = torch.full_like(s.chunk_freq, fill)
        buf.fill_(0x5A5A5A5A)
        rc = lib.hf_encode_chunks_slotted(h, P(t), n, 0, P(s.chunk_codebooks), P(wild), None,
                                          P(buf), cap, P(off), P(cw), C.byref(total))
        assert rc in (hf.L.HF_ERR_CORRUPT, hf.L.HF_ERR_OVERFLOW)
        assert bool((buf[cap:] == 0x5A5A5A5A).all())
    # a stream without its words array, a chunk whose words pass its slot
    out = torch.empty(n, dtype=torch.uint8, device=DEV)
    rc = lib.hf_decode_chunks_slotted(h, P(s.payload), s.payload.numel(), P(s.chunk_off), None,
                                      None, n, 0, P(s.chunk_codebooks), None, None, 0, P(out), 0, 1)
    assert rc == hf.L.HF_ERR_ARG
    bad = s.chunk_words.clone()
    bad[2] = int(s.chunk_off[3].item()) - int(s.chunk_off[2].item()) + 1
    rc = lib.hf_decode_chunks_slotted(h, P(s.payload), s.payload.numel(), P(s.chunk_off), P(bad),
                                      None, n, 0, P(s.chunk_codebooks), None, None, 0, P(out), 0, 1)
    assert rc == hf.L.HF_ERR_CORRUPT
    got = out.cpu().numpy()
    assert np.array_equal(got[: 2 * 16384], sym[: 2 * 16384])
    assert np.array_equal(got[3 * 16384:], sym[3 * 16384:])
    # the context is usable afterwards
    s2 = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, slotted=True)
    assert s2.total_words == s.total_words


def test_one_codebook_slotted_empty_input(hf, orc):
    sys_ = hf.EntropyAwareHuffmanSystem()
    t = torch.empty(0, dtype=torch.uint8, device=DEV)
    s = sys_.encode(t, slotted=True)
    assert s.total_words == 0 and int(s.chunk_off[0].item()) == 0
    assert hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s).numel() == 0


def test_slotted_empty_input(hf, orc):
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    t = torch.empty(0, dtype=torch.uint8, device=DEV)
    s = sys_.encode_blocks(t, slotted=True)
    assert s.total_words == 0 and int(s.chunk_off[0].item()) == 0
    assert dec.decode(s).numel() == 0


@pytest.mark.parametrize("kind,n", [
    ("geom", 1), ("geom", 16384), ("geom", 16385), ("mixed", 9 * 16384 + 77), ("uniform", 4 * 16384),
    ("const", 3 * 16384 + 5), ("mixed", 1100 * 16384 + 4097),
])
def test_one_codebook_slotted_stream_matches_the_oracle(hf, orc, kind, n):
    sym = make(kind, n, n % 1000 + 11)
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, slotted=True)
    # the folded per-chunk counts are the stream's counts, so the codebook is the stream's
    code, lens, mode, _ = s.codebook.download()
    ec, el = orc.rp_system_codebook(orc.aw_symbol_stat(sym, 256, 64), 64, 0, 16)
    assert mode == 1 and np.array_equal(code, ec) and np.array_equal(lens, el)
    payload, slot_off, words = orc.rp_encode_slotted(sym, ec, el)
    nch = s.nchunks
    assert s.total_words == payload.size == int(slot_off[nch])
    assert np.array_equal(s.chunk_off[: nch + 1].cpu().numpy().view(np.uint32), slot_off)
    assert np.array_equal(s.chunk_words[:nch].cpu().numpy().view(np.uint32), words)
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    assert torch.equal(dec.decode(s), t)
    blob = s.to_container(sys_.ctx)
    exp = orc.container_pack(n, orc.LAYOUT_U8, payload, slot_off, code=ec, lens=el, mode=1,
                             entropy_q16=orc.rp_entropy_q16_u64(orc.aw_symbol_stat(sym, 256, 64)),
                             placement=1, chunk_words=words)
    assert np.array_equal(blob.cpu().numpy(), exp)
    s2 = hf.EncodedStream.from_container(blob, ctx=sys_.ctx)
    assert s2.chunk_words is not None and s2.codebook is not None
    assert torch.equal(dec.decode(s2), t)


def test_one_codebook_slotted_bf16_and_sub_fields(hf, orc):
    n = 41 * 16384 + 321
    rng = np.random.default_rng(13)
    w = (rng.standard_normal(n) * 0.02).astype(np.float32)
    bits = (w.view(np.uint32) >> 16).astype(np.uint16)
    t = to_dev(bits).view(torch.bfloat16)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, slotted=True)
    esym, eside = orc.symbolize(bits, orc.LAYOUT_BF16_EXP)
    ec, el = orc.rp_system_codebook(orc.aw_symbol_stat(esym, 256, 64), 64, 0, 16)
    payload, slot_off, words = orc.rp_encode_slotted(esym, ec, el)
    assert s.total_words == payload.size
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    assert np.array_equal(s.side[:n].cpu().numpy(), eside)
    assert torch.equal(dec.decode(s).view(torch.int16), t.view(torch.int16))
    # fp8 E4M3 exponent fields in front of the same path
    b = (rng.standard_normal(n) * 0.3).astype(np.float32)
    t8 = torch.from_numpy(b).to(DEV).to(torch.float8_e4m3fn)
    s8 = sys_.encode(t8.view(torch.uint8), layout=hf.HF_LAYOUT_FP8_E4M3_EXP, slotted=True)
    assert s8.chunk_words is not None
    assert torch.equal(dec.decode(s8), t8.view(torch.uint8))


@pytest.mark.parametrize("mode", ["bypass", "auto", "best"])
def test_one_codebook_slotted_other_modes(hf, orc, mode):
    n = 6 * 16384 + 1000
    sym = make("uniform" if mode != "best" else "geom", n, 31)
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    m = {"bypass": hf.HF_MODE_BYPASS, "auto": hf.HF_MODE_AUTO, "best": hf.HF_MODE_BEST}[mode]
    s = sys_.encode(t, mode=m, slotted=True)
    code, lens, got_mode, _ = s.codebook.download()
    payload, slot_off, words = orc.rp_encode_slotted(sym, code, lens)
    if mode == "bypass":
        assert got_mode == 0 and (lens == 8).all()
    assert s.total_words == payload.size
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    assert np.array_equal(s.chunk_words[: s.nchunks].cpu().numpy().view(np.uint32), words)
    assert torch.equal(hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s), t)


def test_fold_adds_per_chunk_counts_into_the_streams(hf, orc):
    n = 700 * 16384 + 99
    sym = make("mixed", n, 5)
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    lib, h = hf.L.lib(), sys_.ctx.handle
    nch = (n + 16383) // 16384
    fc = torch.empty(nch * 256, dtype=torch.int32, device=DEV)
    f = torch.full((256,), 7, dtype=torch.int64, device=DEV)  # fold ADDS
    P = lambda x: C.c_void_p(x.data_ptr())
    assert lib.hf_symbol_stat_chunks_dev(h, P(t), n, 0, P(fc)) == 0
    assert lib.hf_symbol_stat_fold_dev(h, P(fc), nch, P(f)) == 0
    sys_.ctx.sync()
    want = np.bincount(sym, minlength=256).astype(np.int64) + 7
    assert np.array_equal(f.cpu().numpy(), want)
    assert lib.hf_symbol_stat_fold_dev(h, None, 0, None) == 0
    assert lib.hf_symbol_stat_fold_dev(h, None, 3, P(f)) == hf.L.HF_ERR_ARG


def test_one_codebook_slotted_refusals(hf, orc):
    n = 5 * 16384 + 100
    sym = make("geom", n, 3)
    t = to_dev(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    lib, h = hf.L.lib(), sys_.ctx.handle
    s = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, slotted=True)
    P = lambda x: C.c_void_p(x.data_ptr())
    total = C.c_uint64(0)
    small = torch.full((s.total_words - 1,), 0x5A5A5A5A, dtype=torch.int32, device=DEV)
    off = torch.empty(8, dtype=torch.int32, device=DEV)
    cw = torch.empty(8, dtype=torch.int32, device=DEV)
    rc = lib.hf_encode_slotted(h, P(t), n, 0, C.c_void_p(s.codebook.ptr), P(s.chunk_freq), None,
                               P(small), small.numel(), P(off), P(cw), C.byref(total))
    assert rc == hf.L.HF_ERR_OVERFLOW and bool((small == 0x5A5A5A5A).all())
    out = torch.empty(n, dtype=torch.uint8, device=DEV)
    rc = lib.hf_decode_slotted(h, P(s.payload), s.payload.numel(), P(s.chunk_off), None, None, n, 0,
                               C.c_void_p(s.codebook.ptr), P(out), 1)
    assert rc == hf.L.HF_ERR_ARG
    bad = s.chunk_words.clone()
    bad[1] = int(s.chunk_off[2].item()) - int(s.chunk_off[1].item()) + 1
    rc = lib.hf_decode_slotted(h, P(s.payload), s.payload.numel(), P(s.chunk_off), P(bad), None, n,
                               0, C.c_void_p(s.codebook.ptr), P(out), 1)
    assert rc == hf.L.HF_ERR_CORRUPT
    got = out.cpu().numpy()
    assert np.array_equal(got[:16384], sym[:16384]) and np.array_equal(got[2 * 16384:], sym[2 * 16384:])
    assert torch.equal(hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s), t)
