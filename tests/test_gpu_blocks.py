"""GPU parity of the per-block path: one codebook per 16384-symbol chunk (fp8 KV-cache pages
with per-block EntropyAwareTreeBuilder codebooks, decode-on-read), checked against the CPU
oracle run block by block."""
import os

import numpy as np
import pytest
import torch

from conftest import gaussian_bf16, skewed_symbols

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CH = 16384


def kv_pages_fp8(n_pages: int, seed: int = 7) -> np.ndarray:
    """16-token x 1024-byte pages: N(0,1) x per-channel scale -> fp8 E4M3 bytes."""
    g = torch.Generator().manual_seed(seed)
    scale = torch.exp(torch.randn(1024, generator=g) * 0.7)
    x = torch.randn(n_pages * 16, 1024, generator=g) * scale
    return x.to(torch.float8_e4m3fn).view(torch.uint8).reshape(-1).numpy()


def oracle_blocks(orc, sym, mode):
    """Block-by-block oracle: per-chunk codebook, payload concatenated, offsets accumulated."""
    payloads, offs, books, total = [], [0], [], 0
    for lo in range(0, sym.size, CH):
        blk = sym[lo:lo + CH]
        f = orc.aw_symbol_stat(blk, 256, 32)
        ent = orc.rp_entropy_q16_u64(f)
        m = orc.aw_adaptive_mode(ent) if mode == 2 else mode
        if m == 1:
            code, lens = orc.rp_system_codebook(f, 64, 0, 16)
        else:
            code, lens = np.arange(256, dtype=np.uint16), np.full(256, 8, np.uint8)
        p, _ = orc.rp_encode(blk, code, lens)
        payloads.append(p)
        total += p.size
        offs.append(total)
        books.append((code, lens, m, ent))
    payload = np.concatenate(payloads) if payloads else np.zeros(0, np.uint32)
    return payload, np.array(offs, np.uint32), books


@pytest.mark.parametrize("n_pages,mode", [(1, 1), (5, 1), (33, 2), (8, 0)])
def test_fp8_pages_vs_oracle(hf, orc, n_pages, mode):
    sym = kv_pages_fp8(n_pages, seed=7 + n_pages)
    t = torch.from_numpy(sym).to(DEV)
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode_blocks(t, mode=mode)
    payload, off, books = oracle_blocks(orc, sym, mode)
    # per-block histograms and codebooks
    freq = s.chunk_freq.cpu().numpy().view(np.uint32).reshape(-1, 256)
    for c in range(n_pages):
        assert np.array_equal(freq[c], orc.aw_symbol_stat(sym[c * CH:(c + 1) * CH], 256, 32))
        code, lens, m, ent = sys_.block_codebook(s, c)
        assert (m, ent) == (books[c][2], books[c][3])
        assert np.array_equal(code, books[c][0]) and np.array_equal(lens, books[c][1])
    assert s.total_words == payload.size
    assert np.array_equal(s.chunk_off[: n_pages + 1].cpu().numpy().view(np.uint32), off)
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    out = hf.ParrallelHuffmanDecoder().decode(s)
    assert torch.equal(out, t)


def test_decode_on_read_random_page_order(hf):
    n_pages = 64
    sym = kv_pages_fp8(n_pages, seed=3)
    t = torch.from_numpy(sym).to(DEV)
    s = hf.EntropyAwareHuffmanSystem().encode_blocks(t, mode=hf.HF_MODE_HUFFMAN)
    ids = torch.from_numpy(np.random.default_rng(0).permutation(n_pages)[:40].astype(np.int32)).to(DEV)
    out = hf.ParrallelHuffmanDecoder().decode_blocks(s, chunk_ids=ids)
    exp = t.view(n_pages, CH)[ids.long()].reshape(-1)
    assert torch.equal(out, exp)
    # an id past the stream is reported
    bad = torch.tensor([1, 10_000], dtype=torch.int32, device=DEV)
    with pytest.raises(hf.HuffhipError) as ei:
        hf.ParrallelHuffmanDecoder().decode_blocks(s, chunk_ids=bad)
    assert ei.value.code == -7


def test_decode_on_read_into_a_page_pool(hf):
    # listed pages land in caller-named slots of a larger pool; the other slots stay untouched
    n_pages, pool_pages = 96, 256
    sym = kv_pages_fp8(n_pages, seed=9)
    t = torch.from_numpy(sym).to(DEV)
    s = hf.EntropyAwareHuffmanSystem().encode_blocks(t, mode=hf.HF_MODE_HUFFMAN)
    rng = np.random.default_rng(1)
    ids = torch.from_numpy(rng.permutation(n_pages)[:50].astype(np.int32)).to(DEV)
    slots = torch.from_numpy(rng.permutation(pool_pages)[:50].astype(np.int32)).to(DEV)
    pool = torch.full((pool_pages * CH,), 0xEE, dtype=torch.uint8, device=DEV)
    dec = hf.ParrallelHuffmanDecoder()
    dec.decode_blocks_to(s, ids, slots, pool)
    exp = torch.full((pool_pages, CH), 0xEE, dtype=torch.uint8, device=DEV)
    exp[slots.long()] = t.view(n_pages, CH)[ids.long()]
    assert torch.equal(pool.view(pool_pages, CH), exp)
    # a slot past the pool is reported and nothing is written for it
    bad_slots = slots.clone()
    bad_slots[3] = pool_pages
    pool2 = torch.full((pool_pages * CH,), 0xEE, dtype=torch.uint8, device=DEV)
    with pytest.raises(hf.HuffhipError) as ei:
        dec.decode_blocks_to(s, ids, bad_slots, pool2)
    assert ei.value.code == -7
    exp[slots[3].long()] = 0xEE
    assert torch.equal(pool2.view(pool_pages, CH), exp)
    # bf16 pages go the same way
    w = gaussian_bf16(6 * CH, seed=21)
    tb = torch.from_numpy(w.view(np.int16)).to(DEV).view(torch.bfloat16)
    sb = hf.EntropyAwareHuffmanSystem().encode_blocks(tb, mode=hf.HF_MODE_HUFFMAN)
    idb = torch.tensor([5, 0, 3], dtype=torch.int32, device=DEV)
    slb = torch.tensor([1, 7, 2], dtype=torch.int32, device=DEV)
    poolb = torch.zeros(8 * CH, dtype=torch.bfloat16, device=DEV)
    dec.decode_blocks_to(sb, idb, slb, poolb)
    expb = torch.zeros(8, CH, dtype=torch.int16, device=DEV)
    expb[slb.long()] = tb.view(torch.int16).view(6, CH)[idb.long()]
    assert torch.equal(poolb.view(torch.int16).view(8, CH), expb)


@pytest.mark.parametrize("kind", ["one", "two", "fib", "geometric", "uniform"])
def test_degenerate_blocks(hf, orc, kind):
    # 1-2 symbol blocks, blocks that need the length limiter, ragged tail
    sym = skewed_symbols(3 * CH + 1234, 11, kind)
    t = torch.from_numpy(sym).to(DEV)
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN)
    payload, off, books = oracle_blocks(orc, sym, 1)
    assert s.total_words == payload.size
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    for c in range(4):
        code, lens, _, _ = sys_.block_codebook(s, c)
        assert np.array_equal(code, books[c][0]) and np.array_equal(lens, books[c][1])
        assert lens.max() <= 16
    assert torch.equal(hf.ParrallelHuffmanDecoder().decode(s), t)


def test_bf16_blocks(hf, orc):
    w = gaussian_bf16(5 * CH + 77, seed=4)
    t = torch.from_numpy(w.view(np.int16)).to(DEV).view(torch.bfloat16)
    s = hf.EntropyAwareHuffmanSystem().encode_blocks(t, mode=hf.HF_MODE_HUFFMAN)
    sym, side = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
    payload, off, _ = oracle_blocks(orc, sym, 1)
    assert s.total_words == payload.size
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    assert np.array_equal(s.side[: w.size].cpu().numpy(), side)
    out = hf.ParrallelHuffmanDecoder().decode(s)
    assert torch.equal(out.view(torch.int16), t.view(torch.int16))


def test_many_pages_round_trip(hf):
    # 2048 pages (32 MiB): round trip plus the size identity sum(freq * len) <= payload bits
    n_pages = 2048
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(n_pages * CH, generator=g, device=DEV)
    t = x.to(torch.float8_e4m3fn).view(torch.uint8)
    s = hf.EntropyAwareHuffmanSystem().encode_blocks(t, mode=hf.HF_MODE_HUFFMAN)
    out = hf.ParrallelHuffmanDecoder().decode(s)
    assert torch.equal(out, t)
    ids = torch.randperm(n_pages, device=DEV)[:512].to(torch.int32)
    part = hf.ParrallelHuffmanDecoder().decode_blocks(s, chunk_ids=ids)
    assert torch.equal(part, t.view(n_pages, CH)[ids.long()].reshape(-1))
    assert 1.0 < s.ratio() < 2.0


def test_block_codebooks_from_caller_counts(hf, orc):
    # counts that do not come from a 16384-symbol block: sums of 2^16 and more take the general
    # 64-bit stages, small ones the per-block fast path; both must agree with the oracle
    import ctypes as C
    from huffman_amd import _lib as L
    rng = np.random.default_rng(21)
    n_blocks = 70
    counts = np.zeros((n_blocks, 256), np.uint32)
    for c in range(n_blocks):
        kind = c % 5
        if kind == 0:    # a real block
            counts[c] = np.bincount(rng.integers(0, 256, CH), minlength=256)
        elif kind == 1:  # large, smooth
            counts[c] = rng.integers(1 << 20, 1 << 28, 256)
        elif kind == 2:  # large with many ties and zeros
            counts[c] = rng.integers(0, 4, 256) * (1 << 22)
        elif kind == 3:  # just below / above the fast-path limit
            counts[c, :8] = [30000, 20000, 10000, 5000, 300, 200, 30 + (c % 2) * 10, 5]
        else:            # geometric, needs the length limiter
            counts[c, :40] = (1 << np.arange(40).clip(max=31)).astype(np.uint32)
    ctx = hf.Context(0)
    lib = L.lib()
    d_freq = torch.from_numpy(counts.view(np.int32).reshape(-1)).to(DEV)
    d_cbs = torch.zeros(n_blocks * L.HF_CHUNK_CODEBOOK_BYTES, dtype=torch.uint8, device=DEV)
    L.check(lib.hf_system_codebook_chunks_dev(ctx.handle, C.c_void_p(d_freq.data_ptr()), n_blocks,
                                              hf.HF_MODE_HUFFMAN, 0, C.c_void_p(d_cbs.data_ptr())),
            "codebooks")
    code = np.zeros(256, np.uint16)
    lens = np.zeros(256, np.uint8)
    mode, ent = C.c_int(0), C.c_uint32(0)
    for c in range(n_blocks):
        L.check(lib.hf_chunk_codebook_download(
            ctx.handle, C.c_void_p(d_cbs.data_ptr()), c,
            code.ctypes.data_as(C.POINTER(C.c_uint16)), lens.ctypes.data_as(C.POINTER(C.c_uint8)),
            C.byref(mode), C.byref(ent)), "download")
        f = counts[c].astype(np.uint64)
        ecode, elens = orc.rp_system_codebook(f, 64, 0, 16)
        assert ent.value == orc.rp_entropy_q16_u64(f), c
        assert np.array_equal(lens, elens), c
        assert np.array_equal(code, ecode), c


def test_best_mode_keeps_huffman_only_where_it_pays(hf, orc):
    # blocks of uniform bytes (Huffman cannot win: bypass), skewed blocks (Huffman), an empty-ish
    # tail; the stream-wide codebook follows the same rule on the whole stream
    rng = np.random.default_rng(41)
    blocks = []
    for c in range(24):
        if c % 3 == 0:
            blocks.append(np.arange(CH, dtype=np.uint8))             # exactly uniform: 8.0 bits
        elif c % 3 == 1:
            blocks.append(rng.integers(0, 256, CH, dtype=np.uint8))  # ~7.99 bits: Huffman may win
        else:
            blocks.append(np.minimum(rng.geometric(0.2, CH) - 1, 255).astype(np.uint8))
    sym = np.concatenate(blocks + [np.full(100, 9, np.uint8)])
    t = torch.from_numpy(sym).to(DEV)
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_BEST)
    seen = set()
    for c in range(25):
        blk = sym[c * CH:(c + 1) * CH]
        f = orc.aw_symbol_stat(blk, 256, 32)
        ecode, elens = orc.rp_system_codebook(f, 64, 0, 16)
        emode = orc.rp_best_mode(f, elens)
        code, lens, mode, ent = sys_.block_codebook(s, c)
        assert mode == emode, c
        seen.add(mode)
        if mode == 1:
            assert np.array_equal(code, ecode) and np.array_equal(lens, elens)
        else:
            assert np.array_equal(code, np.arange(256)) and np.all(lens == 8)
    assert seen == {0, 1}
    assert torch.equal(hf.ParrallelHuffmanDecoder().decode(s), t)
    # stream-wide
    for data in (np.arange(1 << 16, dtype=np.uint8), sym):
        d = torch.from_numpy(data).to(DEV)
        st = sys_.encode(d, mode=hf.HF_MODE_BEST)
        f = orc.aw_symbol_stat(data, 256, 64)
        ecode, elens = orc.rp_system_codebook(f, 64, 0, 16)
        code, lens, mode, _ = st.codebook.download()
        assert mode == orc.rp_best_mode(f, elens)
        if mode == 1:
            assert np.array_equal(code, ecode) and np.array_equal(lens, elens)
        assert torch.equal(hf.ParrallelHuffmanDecoder().decode(st), d)


def test_config3_full_size(hf, orc):
    # 2^18 fp8 pages (4 GiB), one codebook per page: the codebook kernels go through their scratch
    # block in four batches.  Round trip and decode-on-read at full size; pages on either side of
    # every batch boundary against the oracle.
    n_pages = 1 << 18
    g = torch.Generator(device=DEV).manual_seed(7)
    scale = torch.exp(torch.randn(1024, generator=g, device=DEV) * 0.7)
    t = torch.empty(n_pages * CH, dtype=torch.uint8, device=DEV)
    step = 1 << 14
    for lo in range(0, n_pages, step):
        x = torch.randn(step * 16, 1024, generator=g, device=DEV) * scale
        t[lo * CH:(lo + step) * CH] = x.to(torch.float8_e4m3fn).view(torch.uint8).reshape(-1)
        del x
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN)
    off = s.chunk_off[: n_pages + 1].cpu().numpy().view(np.uint32).astype(np.int64)
    assert off[0] == 0 and off[-1] == s.total_words and np.all(np.diff(off) > 0)
    for c in (0, 1, 65535, 65536, 131071, 131072, 196607, 196608, n_pages - 1):
        blk = t[c * CH:(c + 1) * CH].cpu().numpy()
        f = orc.aw_symbol_stat(blk, 256, 32)
        code, lens = orc.rp_system_codebook(f, 64, 0, 16)
        gc, gl, m, ent = sys_.block_codebook(s, c)
        assert m == 1 and ent == orc.rp_entropy_q16_u64(f)
        assert np.array_equal(gc, code) and np.array_equal(gl, lens), c
        p, _ = orc.rp_encode(blk, code, lens)
        got = s.payload[int(off[c]):int(off[c + 1])].cpu().numpy().view(np.uint32)
        assert np.array_equal(got, p), c
    out = hf.ParrallelHuffmanDecoder().decode(s)
    assert torch.equal(out, t)
    del out
    ids = torch.randperm(n_pages, device=DEV)[:4096].to(torch.int32)
    part = hf.ParrallelHuffmanDecoder().decode_blocks(s, chunk_ids=ids)
    assert torch.equal(part, t.view(n_pages, CH)[ids.long()].reshape(-1))
    assert 1.05 < s.ratio() < 1.3


def test_seeded_random_pages_against_the_oracle(hf, orc):
    # 48 pages, each with its own random alphabet size and skew (from one symbol to all 256,
    # near-uniform to a few hot symbols), a ragged last page, all four modes
    # (HF_SOAK_SEED / HF_SOAK_SCALE: another seed, and that many times the pages, for a soak run)
    rng = np.random.default_rng(int(os.environ.get("HF_SOAK_SEED", "77")))
    P = 48 * int(os.environ.get("HF_SOAK_SCALE", "1"))
    pages = []
    for _ in range(P):
        k = int(rng.choice([1, 2, 3, rng.integers(4, 40), rng.integers(40, 257)]))
        alpha = float(10.0 ** rng.uniform(-2.5, 1.0))
        p = rng.dirichlet(np.full(k, alpha))
        syms = rng.permutation(256)[:k]
        pages.append(syms[rng.choice(k, size=CH, p=p)].astype(np.uint8))
    sym = np.concatenate(pages)[: (P - 1) * CH + 5000]
    t = torch.from_numpy(sym).to(DEV)
    sys_ = hf.EntropyAwareHuffmanSystem()
    for mode in (hf.HF_MODE_HUFFMAN, hf.HF_MODE_AUTO, hf.HF_MODE_BYPASS):
        s = sys_.encode_blocks(t, mode=mode)
        payload, off, books = oracle_blocks(orc, sym, mode)
        assert s.total_words == payload.size
        assert np.array_equal(s.chunk_off[: P + 1].cpu().numpy().view(np.uint32), off)
        assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
        for c in range(P):
            code, lens, m, ent = sys_.block_codebook(s, c)
            assert (m, ent) == (books[c][2], books[c][3])
            assert np.array_equal(code, books[c][0]) and np.array_equal(lens, books[c][1]), c
        assert torch.equal(hf.ParrallelHuffmanDecoder().decode(s), t)
        # the same pages in slots sized from each page's counts and code lengths
        ss = sys_.encode_blocks(t, mode=mode, slotted=True)
        words = np.diff(off.astype(np.int64))
        slots = np.array([orc.slot_words(int(books[c][1][sym[c * CH:(c + 1) * CH]].astype(np.int64).sum()))
                          for c in range(P)], np.int64)
        starts = np.concatenate([[0], np.cumsum(slots)])
        assert ss.total_words == int(starts[P])
        assert np.array_equal(ss.chunk_off[: P + 1].cpu().numpy().view(np.uint32), starts)
        assert np.array_equal(ss.chunk_words[:P].cpu().numpy().view(np.uint32), words)
        got = ss.payload[: ss.total_words].cpu().numpy().view(np.uint32)
        for c in range(P):
            a = int(starts[c])
            assert np.array_equal(got[a:a + int(words[c])], payload[off[c]:off[c + 1]]), c
            assert not got[a + int(words[c]):int(starts[c + 1])].any(), c
        assert torch.equal(hf.ParrallelHuffmanDecoder().decode(ss), t)
        # ... as pages with their codebook records (exact sizes, four sub-decoders per page) ...
        mreq = {hf.HF_MODE_HUFFMAN: 1, hf.HF_MODE_AUTO: 2, hf.HF_MODE_BYPASS: 0}[mode]
        sp = sys_.encode_blocks(t, mode=mode, paged=True)
        ppay, poff, _, _ = orc.rp_encode_pages(sym, 0, mreq)
        assert sp.total_words == ppay.size
        assert np.array_equal(sp.chunk_off[: P + 1].cpu().numpy().view(np.uint32), poff)
        assert np.array_equal(sp.payload[: ppay.size].cpu().numpy().view(np.uint32), ppay)
        assert torch.equal(hf.ParrallelHuffmanDecoder().decode(sp), t)
        # ... and as pages without records under the stream's one codebook
        if mode != hf.HF_MODE_AUTO:
            so = sys_.encode(t, mode=mode, paged=True)
            if mode == hf.HF_MODE_BYPASS:
                c1, l1 = np.arange(256, dtype=np.uint16), np.full(256, 8, np.uint8)
            else:
                c1, l1 = orc.rp_system_codebook(orc.aw_symbol_stat(sym, 256, 64), 64, 0, 16)
            spay, soff = orc.rp_encode_pages_shared(sym, c1, l1)
            assert so.total_words == spay.size
            assert np.array_equal(so.chunk_off[: P + 1].cpu().numpy().view(np.uint32), soff)
            assert np.array_equal(so.payload[: spay.size].cpu().numpy().view(np.uint32), spay)
            assert torch.equal(hf.ParrallelHuffmanDecoder().decode(so), t)
            # ... the same pages placed while they are encoded (no per-page counts)
            so = sys_.encode(t, mode=mode, paged=True, page_counts=False)
            assert so.total_words == spay.size
            assert np.array_equal(so.chunk_off[: P + 1].cpu().numpy().view(np.uint32), soff)
            assert np.array_equal(so.payload[: spay.size].cpu().numpy().view(np.uint32), spay)
    # HF_MODE_BEST: per page, Huffman only where it pays; the stream still round-trips
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_BEST)
    for c in range(P):
        blk = sym[c * CH:(c + 1) * CH]
        f = orc.aw_symbol_stat(blk, 256, 32)
        _, lens_h = orc.rp_system_codebook(f, 64, 0, 16)
        assert sys_.block_codebook(s, c)[2] == orc.rp_best_mode(f, lens_h)
    assert torch.equal(hf.ParrallelHuffmanDecoder().decode(s), t)


def _random_counts(rng, cap):
    """One 256-bin count vector with sum < cap: ties, zeros, skews and depths of every kind."""
    kind = int(rng.integers(0, 6))
    f = np.zeros(256, np.uint64)
    if kind == 0:      # a sample of a Dirichlet distribution over k symbols
        k = int(rng.integers(1, 257))
        m = int(rng.integers(1, min(cap, 1 << 16)))
        p = rng.dirichlet(np.full(k, float(10.0 ** rng.uniform(-2.5, 1.0))))
        f[rng.permutation(256)[:k]] = rng.multinomial(m, p)
    elif kind == 1:    # small integers: ties everywhere
        f[:] = rng.integers(0, int(rng.integers(2, 200)), 256)
    elif kind == 2:    # geometric with a random ratio: deep trees, the limiter works
        r = float(rng.uniform(1.05, 2.2))
        k = int(rng.integers(2, 257))
        v = np.floor(r ** np.arange(k)).astype(np.float64)
        v = v[np.cumsum(v) < cap / 2]
        f[rng.permutation(256)[:v.size]] = v.astype(np.uint64)
    elif kind == 3:    # Fibonacci weights: the deepest tree a sum allows
        a, b, v = 1, 1, []
        while sum(v) + a < cap / 2 and len(v) < 256:
            v.append(a)
            a, b = b, a + b
        f[rng.permutation(256)[:len(v)]] = np.array(v, np.uint64)
    elif kind == 4:    # a handful of symbols
        k = int(rng.integers(1, 6))
        f[rng.permutation(256)[:k]] = rng.integers(1, max(2, min(cap // 8, 1 << 30)), k)
    else:              # flat with a few outliers
        f[:] = int(rng.integers(1, 60))
        f[rng.permutation(256)[:int(rng.integers(0, 9))]] = int(rng.integers(60, 4000))
    assert int(f.sum()) < cap
    return f


def test_seeded_random_counts_against_the_oracle(hf, orc):
    # count vectors that no 16384-symbol block produces (ties, zeros, Fibonacci and geometric
    # weights, a random min_freq_threshold): the batched per-block codebooks (fast path below a sum
    # of 2^16, general stages above) and the stream-wide codebook kernel (64-bit counts, the same
    # vectors shifted up to 2^55) against the oracle
    # (HF_SOAK_SEED / HF_SOAK_SCALE: another seed, and that many times the vectors, for a soak run)
    import ctypes as C
    from huffman_amd import _lib as L
    rng = np.random.default_rng(int(os.environ.get("HF_SOAK_SEED", "4242")))
    n_blocks = 96 * int(os.environ.get("HF_SOAK_SCALE", "1"))
    ctx = hf.Context(0)
    lib = L.lib()
    sys_ = hf.EntropyAwareHuffmanSystem(ctx=ctx)
    code = np.zeros(256, np.uint16)
    lens = np.zeros(256, np.uint8)
    mode, ent = C.c_int(0), C.c_uint32(0)
    for thr in (0, int(rng.integers(1, 40))):
        counts = np.stack([_random_counts(rng, (1 << 16) if c % 3 else (1 << 31))
                           for c in range(n_blocks)])
        d_freq = torch.from_numpy(counts.astype(np.uint32).view(np.int32).reshape(-1)).to(DEV)
        d_cbs = torch.zeros(n_blocks * L.HF_CHUNK_CODEBOOK_BYTES, dtype=torch.uint8, device=DEV)
        L.check(lib.hf_system_codebook_chunks_dev(ctx.handle, C.c_void_p(d_freq.data_ptr()),
                                                  n_blocks, hf.HF_MODE_HUFFMAN, thr,
                                                  C.c_void_p(d_cbs.data_ptr())), "codebooks")
        for c in range(n_blocks):
            L.check(lib.hf_chunk_codebook_download(
                ctx.handle, C.c_void_p(d_cbs.data_ptr()), c,
                code.ctypes.data_as(C.POINTER(C.c_uint16)),
                lens.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(mode), C.byref(ent)), "download")
            ecode, elens = orc.rp_system_codebook(counts[c], 64, thr, 16)
            assert ent.value == orc.rp_entropy_q16_u64(counts[c]), (thr, c)
            assert np.array_equal(lens, elens), (thr, c)
            assert np.array_equal(code, ecode), (thr, c)
        # the same vectors, shifted up as far as 2^55 allows, through the stream-wide kernel
        for c in range(0, n_blocks, 4):
            f = counts[c]
            room = 55 - int(f.sum()).bit_length()
            f = f << np.uint64(rng.integers(0, room + 1))
            big_thr = thr << int(rng.integers(0, 20))
            cb = sys_.codebook(torch.from_numpy(f.view(np.int64)).to(DEV), hf.HF_MODE_HUFFMAN,
                               big_thr)
            gcode, glens, _, gent = cb.download()
            ecode, elens = orc.rp_system_codebook(f, 64, big_thr, 16)
            assert gent == orc.rp_entropy_q16_u64(f), (thr, c)
            assert np.array_equal(glens, elens) and np.array_equal(gcode, ecode), (thr, c)


@pytest.mark.parametrize("damage", ["decreasing", "overfull", "too_long"])
def test_damaged_page_record_is_reported_and_stays_contained(hf, damage):
    # a page record whose lengths are not a canonical order (they decrease with the rank, fill
    # more than the code space, or exceed 16) is refused by the decoder's table build: the call
    # reports a corrupt stream, the page is left alone and every other page still decodes
    rng = np.random.default_rng(3)
    n_pages = 12
    sym = np.minimum(rng.geometric(0.08, n_pages * CH) - 1, 255).astype(np.uint8)
    t = torch.from_numpy(sym).to(DEV)
    s = hf.EntropyAwareHuffmanSystem().encode_blocks(t, mode=hf.HF_MODE_HUFFMAN)
    from huffman_amd import _lib as L
    rec = s.chunk_codebooks.view(n_pages, L.HF_CHUNK_CODEBOOK_BYTES)
    lens = rec[5, 1280:1536]  # len_by_rank of page 5
    if damage == "decreasing":
        lens[10], lens[200] = lens[200].clone(), lens[10].clone()
    elif damage == "overfull":
        lens[:] = 3
    else:
        lens[255] = 40
    out = torch.full((n_pages * CH,), 0xEE, dtype=torch.uint8, device=DEV)
    with pytest.raises(hf.HuffhipError) as ei:
        hf.ParrallelHuffmanDecoder().decode_blocks(s, out=out)
    assert ei.value.code == -7
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(n_pages, CH)
    exp = sym.reshape(n_pages, CH)
    for c in range(n_pages):
        if c == 5:
            assert np.all(got[c] == 0xEE)
        else:
            assert np.array_equal(got[c], exp[c]), c
