"""GPU parity: every C-ABI entry point against the CPU oracle on the same inputs (bit exact).

Known-answer vectors taken from the reference's own specs are marked with their source lines.
"""
import os

import numpy as np
import pytest
import torch

from conftest import gaussian_bf16, skewed_symbols

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev_u8(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).to(DEV)


def dev_bf16(a_u16):
    return torch.from_numpy(np.ascontiguousarray(a_u16).view(np.int16)).to(DEV).view(torch.bfloat16)


# ---- SymbolStat ----------------------------------------------------------------------------
def test_symbol_stat_reference_kat(hf):
    # t_symbol_stat.scala:14, :28, :48-60 -- SymbolStat(4, 8, 16), inputs 3,3,5,5,5,15
    out = hf.SymbolStat(sym_width=4, wtWidth=8, depth=16)(dev_u8([3, 3, 5, 5, 5, 15]))
    exp = np.zeros(16, np.uint64)
    exp[3], exp[5], exp[15] = 2, 3, 1
    assert np.array_equal(out, exp)


def test_symbol_stat_flush(hf):
    # t_symbol_stat.scala:64-99 -- count a 1, flush, restart with no data: all zero
    st = hf.SymbolStat(sym_width=4, wtWidth=8, depth=16)
    st(dev_u8([1]))
    st.flush()
    assert not st(torch.empty(0, dtype=torch.uint8, device=DEV)).any()


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 4095, 4096, 65537, (1 << 20) + 3])
@pytest.mark.parametrize("kind", ["uniform", "geometric", "one"])
def test_symbol_stat_u8_vs_oracle(hf, orc, n, kind):
    sym = skewed_symbols(n, 11 + n % 7, kind)
    got = hf.SymbolStat(8, 64, 256)(dev_u8(sym) if n else torch.empty(0, dtype=torch.uint8, device=DEV))
    assert np.array_equal(got, orc.aw_symbol_stat(sym, 256, 64))


def test_symbol_stat_counter_wrap(hf, orc):
    sym = np.full(1000, 9, np.uint8)
    got = hf.SymbolStat(8, 8, 256)(dev_u8(sym))
    assert got[9] == 1000 % 256 and np.array_equal(got, orc.aw_symbol_stat(sym, 256, 8))
    # N = 16 keeps the low 4 bits of the symbol (symbol_stat.scala:36-37)
    sym = skewed_symbols(5000, 3, "uniform")
    assert np.array_equal(hf.SymbolStat(8, 32, 16)(dev_u8(sym)), orc.aw_symbol_stat(sym, 16, 32))


@pytest.mark.parametrize("n", [1, 7, 8, 9, 16384, 100003, (1 << 21) + 5])
def test_symbol_stat_bf16_vs_oracle(hf, orc, n):
    w = gaussian_bf16(n, seed=n)
    sym, _ = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
    got = hf.SymbolStat(8, 64, 256)(dev_bf16(w))
    assert np.array_equal(got, orc.aw_symbol_stat(sym, 256, 64))


# ---- SymbolSort ----------------------------------------------------------------------------
def test_symbol_sort_reference_kat(hf):
    # t_symbol_sort.scala:15-16, :46-49 -- SymbolSort(16, 8)
    inputs = [3, 1, 7, 0, 5, 2, 6, 4, 9, 8, 15, 12, 10, 11, 14, 13]
    out = hf.SymbolSort(16, 8)(inputs)
    assert list(out) == sorted(inputs, reverse=True)
    # t_symbol_sort.scala:92-94 -- all zeros stay all zeros
    assert not hf.SymbolSort(16, 8)([0] * 16).any()


@pytest.mark.parametrize("N", [2, 4, 16, 100, 256])
def test_symbol_sort_vs_oracle(hf, orc, N):
    rng = np.random.default_rng(N)
    for hi in (3, 50, 1 << 40):  # heavy ties, some ties, none
        f = rng.integers(0, hi, N).astype(np.uint64)
        srt = hf.SymbolSort(N, 64)
        out = srt(f)
        exp, ids = orc.aw_symbol_sort(f)
        assert np.array_equal(out, exp)
        assert np.array_equal(srt.rank2sym, ids)


# ---- ShannonEntropy / AdaptiveThreshold ----------------------------------------------------
def test_entropy_vs_oracle(hf, orc):
    rng = np.random.default_rng(5)
    cases = []
    f = np.zeros(256, np.uint64); f[7] = 4096; cases.append(f)          # one symbol
    cases.append(np.full(256, 16, np.uint64))                            # uniform
    f = np.zeros(256, np.uint64); f[1] = f[9] = 2048; cases.append(f)    # two symbols
    cases.append(np.zeros(256, np.uint64))                               # empty block
    for _ in range(20):
        cases.append(rng.integers(0, 1 << int(rng.integers(1, 24)), 256).astype(np.uint64))
    cases.append(rng.integers(0, 1 << 36, 256).astype(np.uint64))        # total above 32 bits
    for f in cases:
        e, m = hf.ShannonEntropy(256, 32)(f)
        exp = orc.rp_entropy_q16_u64(f)
        assert e == exp
        assert m == orc.aw_adaptive_mode(exp)
        if int(f.sum()) < (1 << 32):
            assert e == orc.aw_entropy_q16(f, 32)


# ---- HuffmanTreeBuilder ---------------------------------------------------------------------
TREE_CASES = [
    [5, 1, 3, 2],                 # input of t_tree_builder.scala:58
    [5, 3, 2, 1],
    [0, 0, 0, 0],                 # input of t_tree_builder.scala:22-24
    [8, 4, 2, 1, 1, 0, 0, 0],
    [1] * 8,
    [13, 8, 5, 3, 2, 1, 1, 0],
]


@pytest.mark.parametrize("freqs", TREE_CASES)
def test_tree_build_small(hf, orc, freqs):
    N = len(freqs)
    t = hf.HuffmanTreeBuilder(N, 32)(freqs)
    o = orc.aw_tree_build(freqs, 32)
    for k in ("weights", "lefts", "rights"):
        assert np.array_equal(t[k], o[k]), k
    assert t["root"] == o["root"]
    assert np.array_equal(t["depths"], orc.rp_leaf_depths(o["lefts"], o["rights"], o["root"], N))


@pytest.mark.parametrize("seed", range(6))
def test_tree_build_256_vs_oracle(hf, orc, seed):
    rng = np.random.default_rng(seed)
    hi = [2, 5, 100, 1 << 16, 1 << 30, 1 << 44][seed]
    f = rng.integers(0, hi, 256).astype(np.uint64)
    if seed % 2:
        f = np.sort(f)[::-1].copy()  # as the system feeds it
    W = 64 if hi > (1 << 20) else 32
    t = hf.HuffmanTreeBuilder(256, W)(f)
    o = orc.aw_tree_build(f, W)
    for k in ("weights", "lefts", "rights"):
        assert np.array_equal(t[k], o[k]), k
    assert np.array_equal(t["depths"], orc.rp_leaf_depths(o["lefts"], o["rights"], o["root"], 256))


def test_tree_build_refuses_wrapping_weights(hf):
    f = np.full(256, 1 << 25, np.uint64)  # sum = 2^33 wraps a 32-bit weight
    with pytest.raises(hf.HuffhipError) as ei:
        hf.HuffmanTreeBuilder(256, 32)(f)
    assert ei.value.code == -3


# ---- tree_adjust ----------------------------------------------------------------------------
def fib_counts(k=40, N=256):
    fib = [1, 1]
    while len(fib) < k:
        fib.append(fib[-1] + fib[-2])
    f = np.zeros(N, np.uint64)
    f[:k] = fib[::-1]
    return f


def kraft(lens):
    return sum(2.0 ** -int(l) for l in lens)


def test_tree_adjust_vs_oracle(hf, orc):
    rng = np.random.default_rng(2)
    cases = []
    for f in (fib_counts(), np.array([1 << min(30, 255 - i) if i < 31 else 0 for i in range(256)], np.uint64),
              rng.integers(0, 1000, 256).astype(np.uint64), np.zeros(256, np.uint64)):
        o = orc.aw_tree_build(f, 64)
        cases.append(orc.rp_leaf_depths(o["lefts"], o["rights"], o["root"], 256))
    cases.append(np.arange(1, 257).clip(max=255).astype(np.uint16))  # a 255-deep chain
    cases.append(np.zeros(256, np.uint16))                            # zeros (the dead depths port)
    for d in cases:
        for lmax in (16, 12, 8):
            got = hf.tree_adjust(256, lmax)(d)
            exp, _ = orc.rp_tree_adjust(d, lmax)
            assert np.array_equal(got, exp)
            assert got.max() <= lmax and kraft(got) <= 1.0 + 1e-12
            assert np.all(np.diff(got.astype(int)) >= 0)
    # small N: depths of the 4-leaf tree [1,3,2,3] are re-dealt in index order
    assert list(hf.tree_adjust(4, 16)([1, 3, 2, 3])) == list(orc.rp_tree_adjust([1, 3, 2, 3], 16)[0])


# ---- EntropyAwareTreeBuilder ----------------------------------------------------------------
@pytest.mark.parametrize("N", [4, 8, 16, 256])
def test_codebook_module_vs_oracle(hf, orc, N):
    rng = np.random.default_rng(N)
    for trial in range(4):
        f = rng.integers(0, [3, 100, 1 << 20, 1 << 40][trial], N).astype(np.uint64)
        thr = [0, 1, 50, 0][trial]
        code, lens = hf.EntropyAwareTreeBuilder(N, 64)(f, min_freq_threshold=thr)
        ec, el = orc.rp_codebook(f, 64, thr, 16)
        assert np.array_equal(code, ec) and np.array_equal(lens, el)


def test_codebook_length_limit(hf, orc):
    f = np.sort(fib_counts())[::-1].copy()
    code, lens = hf.EntropyAwareTreeBuilder(256, 64)(f)
    ec, el = orc.rp_codebook(f, 64, 0, 16)
    assert np.array_equal(code, ec) and np.array_equal(lens, el)
    assert lens.max() == 16 and abs(kraft(lens) - 1.0) < 1e-12


# ---- system codebook + encode + decode -------------------------------------------------------
def system_roundtrip(hf, orc, data_t, sym, mode=1, side=None):
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode(data_t, mode=mode)
    code, lens, m, ent = s.codebook.download()
    freq = orc.aw_symbol_stat(sym, 256, 64)
    assert ent == orc.rp_entropy_q16_u64(freq)
    if mode == hf.HF_MODE_AUTO:
        assert m == orc.aw_adaptive_mode(ent)
    if m == 1:
        ec, el = orc.rp_system_codebook(freq, 64, 0, 16)
    else:
        ec, el = np.arange(256, dtype=np.uint16), np.full(256, 8, np.uint8)
    assert np.array_equal(code, ec) and np.array_equal(lens, el)
    payload, off = orc.rp_encode(sym, ec, el)
    nch = orc.num_chunks(sym.size)
    assert s.total_words == payload.size
    assert np.array_equal(s.chunk_off[: nch + 1].cpu().numpy().view(np.uint32), off)
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    if side is not None:
        assert np.array_equal(s.side[: sym.size].cpu().numpy(), side)
    out = hf.ParrallelHuffmanDecoder().decode(s)
    assert torch.equal(out.view(torch.int16) if out.dtype == torch.bfloat16 else out,
                       data_t.view(torch.int16) if data_t.dtype == torch.bfloat16 else data_t)
    return s


@pytest.mark.parametrize("n", [0, 1, 3, 5, 4096, 16385, 65535, 65536])
@pytest.mark.parametrize("bf16", [False, True])
def test_block_codebook_in_one_launch(hf, orc, n, bf16):
    # SymbolStat + the system codebook of one block (hf_system_codebook_block_dev) against the
    # oracle and against the two separate calls
    import ctypes as C
    if bf16:
        w = gaussian_bf16(max(n, 1), seed=n + 3)[:n]
        sym, _ = orc.symbolize(w, orc.LAYOUT_BF16_EXP) if n else (np.zeros(0, np.uint8), None)
        t = dev_bf16(w) if n else torch.empty(0, dtype=torch.bfloat16, device=DEV)
    else:
        sym = skewed_symbols(max(n, 1), n + 1, "geometric")[:n]
        t = dev_u8(sym) if n else torch.empty(0, dtype=torch.uint8, device=DEV)
    sys_ = hf.EntropyAwareHuffmanSystem()
    stale = torch.full((256,), 12345, dtype=torch.int64, device=DEV)  # overwritten, not added to
    for mode in (hf.HF_MODE_HUFFMAN, hf.HF_MODE_AUTO):
        cb, freq = sys_.codebook_of_block(t, mode, freq256=stale)
        want = orc.aw_symbol_stat(sym, 256, 64)
        assert np.array_equal(freq.cpu().numpy().astype(np.uint64), want)
        code, lens, m, ent = cb.download()
        two = sys_.codebook(sys_.histogram(t), mode)
        c2, l2, m2, e2 = two.download()
        assert (m, ent) == (m2, e2) and np.array_equal(code, c2) and np.array_equal(lens, l2)
        assert ent == orc.rp_entropy_q16_u64(want)
        if m == 1:
            ec, el = orc.rp_system_codebook(want, 64, 0, 16)
            assert np.array_equal(code, ec) and np.array_equal(lens, el)
    if n == 65536:
        big = torch.zeros(65552, dtype=torch.uint8, device=DEV)
        rc = hf.L.lib().hf_system_codebook_block_dev(
            sys_.ctx.handle, C.c_void_p(big.data_ptr()), 65537, 0, 1, 0,
            C.c_void_p(stale.data_ptr()), C.c_void_p(cb.ptr))
        assert rc == hf.L.HF_ERR_RANGE


def test_config1_4k_uniform(hf, orc):
    # 4096 symbols, uniform over 0..255, seed 1, Huffman forced
    sym = np.random.default_rng(1).integers(0, 256, 4096, dtype=np.uint8)
    s = system_roundtrip(hf, orc, dev_u8(sym), sym, mode=hf.HF_MODE_HUFFMAN)
    assert s.codebook.download()[2] == 1


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 511, 512, 513, 16383, 16384, 16385, 40000, 3 * 16384 + 77])
def test_encode_decode_ragged_u8(hf, orc, n):
    sym = skewed_symbols(n, n, "geometric")
    system_roundtrip(hf, orc, dev_u8(sym), sym, mode=hf.HF_MODE_HUFFMAN)


@pytest.mark.parametrize("kind", ["uniform", "geometric", "one", "two", "fib"])
@pytest.mark.parametrize("n", [4096, 200000])
def test_encode_decode_skewed(hf, orc, kind, n):
    sym = skewed_symbols(n, 17, kind)
    system_roundtrip(hf, orc, dev_u8(sym), sym, mode=hf.HF_MODE_HUFFMAN)


@pytest.mark.parametrize("kind", ["geometric", "one", "two", "fib", "equal"])
def test_encode_decode_64mib_block(hf, orc, kind):
    # the degenerate count shapes at the large block size: 64 MiB of byte symbols, whole payload
    # compared with the oracle's
    n = 64 << 20
    if kind == "equal":
        sym = np.tile(np.arange(256, dtype=np.uint8), n // 256)
    else:
        sym = skewed_symbols(n, 23, kind)
    s = system_roundtrip(hf, orc, dev_u8(sym), sym, mode=hf.HF_MODE_HUFFMAN)
    lens = s.codebook.download()[1]
    assert lens.max() <= 16 and kraft(lens) <= 1.0
    if kind == "fib":
        assert lens.max() == 16  # the unconstrained tree is deeper; the limiter was at work


@pytest.mark.parametrize("n", [1, 8, 1000, 16384, 16384 * 5 + 13, 1 << 20])
def test_encode_decode_bf16(hf, orc, n):
    w = gaussian_bf16(n, seed=n + 1)
    sym, side = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
    system_roundtrip(hf, orc, dev_bf16(w), sym, mode=hf.HF_MODE_HUFFMAN, side=side)


def test_fp16_and_int16_words_use_the_16_bit_split(hf, orc):
    # fp16: symbol = 5 exponent bits + 3 mantissa bits, raw byte = sign + 7 mantissa bits
    rng = np.random.default_rng(5)
    w = (rng.standard_normal(300000) * 0.02).astype(np.float16).view(np.uint16)
    sym, side = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
    t = torch.from_numpy(w.view(np.int16)).to(DEV)
    for data in (t.view(torch.float16), t):
        s = system_roundtrip(hf, orc, data, sym, mode=hf.HF_MODE_HUFFMAN, side=side)
        assert s.dtype == data.dtype
        assert s.compressed_bytes() < 2 * w.size * 0.9
    # two fp4 values per byte are byte symbols
    p4 = np.clip(np.round(rng.standard_normal(200000) * 2), -7, 7).astype(np.int8) & 0xF
    packed = (p4[0::2] | (p4[1::2] << 4)).astype(np.uint8)
    s = system_roundtrip(hf, orc, dev_u8(packed), packed, mode=hf.HF_MODE_HUFFMAN)
    assert s.compressed_bytes() < packed.size * 0.8


def test_empty_stream(hf):
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode(torch.empty(0, dtype=torch.uint8, device=DEV), mode=hf.HF_MODE_HUFFMAN)
    assert s.total_words == 0 and int(s.chunk_off[0]) == 0
    assert hf.ParrallelHuffmanDecoder().decode(s).numel() == 0


def test_bypass_and_auto_modes(hf, orc):
    sym = skewed_symbols(50000, 4, "uniform")
    s = system_roundtrip(hf, orc, dev_u8(sym), sym, mode=hf.HF_MODE_BYPASS)
    assert s.codebook.download()[2] == 0
    for kind in ("uniform", "one", "geometric"):
        sym = skewed_symbols(30000, 6, kind)
        system_roundtrip(hf, orc, dev_u8(sym), sym, mode=hf.HF_MODE_AUTO)


@pytest.mark.parametrize("n", [1, 7, 8, 9, 10000, 10007, (1 << 20) + 5])
def test_encode_pairs_vs_oracle(hf, orc, n):
    sym = skewed_symbols(n, 8, "geometric")
    d = dev_u8(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    for mode in (hf.HF_MODE_HUFFMAN, hf.HF_MODE_BYPASS):
        cb = sys_.codebook(sys_.histogram(d), mode)
        code, lens, m, _ = cb.download()
        beats = sys_.encode_pairs(d, cb)
        exp = orc.aw_encode_pairs(sym, code, lens, m, 8)
        assert np.array_equal(beats[:, 0], exp[:, 0].astype(np.uint32) | (exp[:, 1].astype(np.uint32) << 8))
        assert np.array_equal(beats[:, 1], exp[:, 2]) and np.array_equal(beats[:, 2], exp[:, 3])


def test_encode_pairs_bf16_vs_oracle(hf, orc):
    w = gaussian_bf16(100003, seed=12)
    sym, _ = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
    d = dev_bf16(w)
    sys_ = hf.EntropyAwareHuffmanSystem()
    cb = sys_.codebook(sys_.histogram(d), hf.HF_MODE_HUFFMAN)
    code, lens, m, _ = cb.download()
    beats = sys_.encode_pairs(d, cb)
    exp = orc.aw_encode_pairs(sym, code, lens, m, 8)
    assert np.array_equal(beats[:, 0], exp[:, 0].astype(np.uint32) | (exp[:, 1].astype(np.uint32) << 8))
    assert np.array_equal(beats[:, 1], exp[:, 2]) and np.array_equal(beats[:, 2], exp[:, 3])


def test_memwrapper_uploaded_codebook(hf, orc):
    # a valid prefix code that is NOT canonical in symbol order, with codes longer than 12 bits
    f = np.sort(fib_counts())[::-1].copy()
    rng = np.random.default_rng(3)
    perm = rng.permutation(256)
    code_r, len_r = orc.rp_codebook(f, 64, 0, 16)
    mw = hf.MemWrapper(256, 16, 8)
    assert not mw.bulk_read()[2]
    for r in range(256):
        mw.write(int(perm[r]), int(code_r[r]), int(len_r[r]))
    code, lens, ready = mw.bulk_read()
    assert ready and mw.read(int(perm[0])) == (int(code_r[0]), int(len_r[0]), True)
    cb = mw.to_device()
    p = f[np.argsort(perm)].astype(np.float64)  # symbol s has the count of its rank
    sym = rng.choice(256, 70000, p=p / p.sum()).astype(np.uint8)
    sym[:256] = np.arange(256)  # every code, including the 16-bit ones, occurs
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode_with(dev_u8(sym), cb)
    payload, off = orc.rp_encode(sym, code, lens)
    assert s.total_words == payload.size
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    out = hf.ParrallelHuffmanDecoder().decode(s)
    assert np.array_equal(out.cpu().numpy(), sym)


def test_payload_overflow_is_reported(hf):
    sym = skewed_symbols(100000, 1, "uniform")
    d = dev_u8(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    cb = sys_.codebook(sys_.histogram(d), hf.HF_MODE_HUFFMAN)
    small = torch.empty(1024, dtype=torch.int32, device=DEV)
    with pytest.raises(hf.HuffhipError) as ei:
        sys_.encode_with(d, cb, payload=small)
    assert ei.value.code == -5


def test_corrupt_offsets_are_reported(hf):
    sym = skewed_symbols(50000, 2, "geometric")
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode(dev_u8(sym), mode=hf.HF_MODE_HUFFMAN)
    s.chunk_off[1] = 0x7FFFFFF0  # points far past the payload buffer
    with pytest.raises(hf.HuffhipError) as ei:
        hf.ParrallelHuffmanDecoder().decode(s)
    assert ei.value.code == -7


def test_high_entropy_chunks_spill(hf, orc):
    # a codebook built for one distribution applied to the opposite one: > 8 bits/symbol,
    # the chunk outgrows the LDS staging area
    f = np.sort(fib_counts())[::-1].copy()
    code_r, len_r = orc.rp_codebook(f, 64, 0, 16)
    mw = hf.MemWrapper()
    for s_ in range(256):
        mw.write(s_, int(code_r[s_]), int(len_r[s_]))
    cb = mw.to_device()
    sym = np.random.default_rng(9).integers(40, 256, 3 * 16384 + 100, dtype=np.uint8)  # all 16-bit codes
    sys_ = hf.EntropyAwareHuffmanSystem()
    bound = int(hf.L.lib().hf_payload_bound_words(sym.size))
    buf = torch.empty(bound, dtype=torch.int32, device=DEV)
    s = sys_.encode_with(dev_u8(sym), cb, payload=buf)
    code, lens, _ = mw.bulk_read()
    payload, off = orc.rp_encode(sym, code, lens)
    assert s.total_words == payload.size
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    assert np.array_equal(hf.ParrallelHuffmanDecoder().decode(s).cpu().numpy(), sym)


def test_quality_vs_oracle(hf, orc):
    rng = np.random.default_rng(12)
    f = rng.integers(0, 5000, 256).astype(np.uint64)
    _, lens = orc.rp_system_codebook(f, 64, 0, 16)
    assert hf.TreeQualityAnalyzer(256, 32)(f, lens) == orc.aw_quality(f, lens, 32)


def test_many_tiles_two_chunk_staging_with_overflow(hf, orc):
    # 4096 chunks (several tiles per workgroup), a skewed stream whose codebook promises small
    # chunks (two chunks staged per wavefront) with uniform-random chunks sprinkled in that
    # outgrow their half of the staging area and spill: bit-exact against the oracle
    n = 4096 * 16384 + 1234
    rng = np.random.default_rng(33)
    sym = np.minimum(rng.geometric(0.45, n) - 1, 255).astype(np.uint8)
    for c in range(5, 4096, 97):
        sym[c * 16384:(c + 1) * 16384] = rng.integers(0, 256, 16384, dtype=np.uint8)
    s = system_roundtrip(hf, orc, dev_u8(sym), sym, mode=hf.HF_MODE_HUFFMAN)
    lens = s.codebook.download()[1]
    assert float((np.bincount(sym, minlength=256) * lens).sum()) / n < 4.0  # two-chunk staging


def test_large_bf16_stream_properties(hf):
    # 192 Mi bf16 elements (12288 chunks): size-independent properties at a size the oracle
    # does not reach in seconds
    n = 192 * 1024 * 1024 + 4321
    g = torch.Generator(device=DEV).manual_seed(5)
    w = torch.randn(n, generator=g, device=DEV).mul_(0.02).to(torch.bfloat16)
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode(w, mode=hf.HF_MODE_HUFFMAN)
    nch = (n + 16383) // 16384
    off = s.chunk_off[: nch + 1].cpu().numpy().view(np.uint32).astype(np.int64)
    assert off[0] == 0 and off[-1] == s.total_words and np.all(np.diff(off) > 0)
    # payload bits = sum(freq * len) + look-ahead and padding words (at most 2 words per lane)
    code, lens, mode, _ = s.codebook.download()
    freq = sys_.histogram(w).cpu().numpy().astype(np.float64)
    ideal_bits = float((freq * lens).sum())
    assert ideal_bits <= 32.0 * s.total_words <= ideal_bits + nch * 64 * 96
    # deterministic: a second encode gives the same stream
    s2 = sys_.encode(w, mode=hf.HF_MODE_HUFFMAN)
    assert s2.total_words == s.total_words
    assert torch.equal(s2.payload[: s.total_words], s.payload[: s.total_words])
    assert torch.equal(s2.chunk_off[: nch + 1], s.chunk_off[: nch + 1])
    out = hf.ParrallelHuffmanDecoder().decode(s)
    assert torch.equal(out.view(torch.int16), w.view(torch.int16))
    assert 1.4 < s.ratio() < 1.6


def test_config2_full_size_properties(hf):
    # the headline workload itself: 8 030 261 248 bf16 elements (16.06 GB), one codebook, forced
    # Huffman.  The oracle cannot follow at this size; what must hold at any size is checked:
    # exact round trip, chunk offsets strictly increasing and ending at the total, payload bits =
    # sum(count x length) + bounded look-ahead, the histogram summing to n, a prefix code.
    import bench
    n = bench.LLAMA3_8B_PARAMS
    if torch.cuda.get_device_properties(0).total_memory < 80e9:
        pytest.skip("needs ~50 GB of device memory")
    w = bench.make_weights(torch, n, DEV)
    sys_ = hf.EntropyAwareHuffmanSystem()
    freq = sys_.histogram(w).cpu().numpy()
    assert int(freq.sum()) == n
    s = sys_.encode(w, mode=hf.HF_MODE_HUFFMAN)
    nch = (n + 16383) // 16384
    off = s.chunk_off[: nch + 1].cpu().numpy().view(np.uint32).astype(np.int64)
    assert off[0] == 0 and off[-1] == s.total_words and np.all(np.diff(off) > 0)
    code, lens, mode, _ = s.codebook.download()
    assert mode == 1 and lens.max() <= 16 and abs(kraft(lens) - 1.0) < 1e-12
    ideal_bits = float((freq.astype(np.float64) * lens).sum())
    assert ideal_bits <= 32.0 * s.total_words <= ideal_bits + nch * 64 * 96
    # the side stream is the sign and mantissa of every element
    probe = slice(n - 100000, n)
    wv = w[probe].view(torch.int16).to(torch.int32) & 0xFFFF
    assert torch.equal(s.side[probe].to(torch.int32), ((wv >> 8) & 0x80) | (wv & 0x7F))
    out = hf.ParrallelHuffmanDecoder().decode(s)
    assert torch.equal(out.view(torch.int16), w.view(torch.int16))
    assert 1.45 < s.ratio() < 1.52


def test_seeded_random_streams_against_the_oracle(hf, orc):
    # 60 streams of random length, symbol distribution (Dirichlet with random concentration, so
    # from near-uniform to a few hot symbols), layout and mode: histogram, codebook, payload,
    # offsets, side bytes and round trip against the oracle, bit for bit
    # (HF_SOAK_SEED / HF_SOAK_SCALE: another seed, and that many times the cases, for a soak run)
    rng = np.random.default_rng(int(os.environ.get("HF_SOAK_SEED", "20240611")))
    for case in range(60 * int(os.environ.get("HF_SOAK_SCALE", "1"))):
        n = int(rng.choice([rng.integers(1, 600), rng.integers(600, 70000), rng.integers(70000, 400000)]))
        alpha = float(10.0 ** rng.uniform(-2.5, 1.0))
        k = int(rng.integers(1, 257))
        p = rng.dirichlet(np.full(k, alpha))
        syms = rng.permutation(256)[:k]
        sym = syms[rng.choice(k, size=n, p=p)].astype(np.uint8)
        mode = int(rng.choice([hf.HF_MODE_HUFFMAN, hf.HF_MODE_HUFFMAN, hf.HF_MODE_AUTO, hf.HF_MODE_BYPASS]))
        if rng.random() < 0.5:
            system_roundtrip(hf, orc, dev_u8(sym), sym, mode=mode)
        else:
            low = rng.integers(0, 128, n).astype(np.uint16)
            sign = rng.integers(0, 2, n).astype(np.uint16)
            w = (sign << 15) | (sym.astype(np.uint16) << 7) | low
            s2, side = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
            assert np.array_equal(s2, sym)
            system_roundtrip(hf, orc, dev_bf16(w), sym, mode=mode, side=side)


def test_two_contexts_on_two_streams_overlap(hf, orc):
    # one context per stream, both streams busy at once with no host synchronisation in between:
    # every context has its own descriptors, ticket, spill area and error word, so the streams
    # must come out as if encoded alone
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    a = dev_bf16(gaussian_bf16(40 * 16384 + 77, seed=41))
    b = dev_u8(skewed_symbols(25 * 16384 + 5, 42, "geometric"))
    torch.cuda.synchronize()
    out = {}
    with torch.cuda.stream(s1):
        c1 = hf.Context(stream=s1)
        sys1 = hf.EntropyAwareHuffmanSystem(ctx=c1)
    with torch.cuda.stream(s2):
        c2 = hf.Context(stream=s2)
        sys2 = hf.EntropyAwareHuffmanSystem(ctx=c2)
    for rep in range(3):  # interleaved launches, nothing waits for anything
        with torch.cuda.stream(s1):
            f1 = sys1.histogram(a)
            cb1 = sys1.codebook(f1, hf.HF_MODE_HUFFMAN)
            out["a"] = sys1.encode_with(a, cb1, sync=False)
        with torch.cuda.stream(s2):
            f2 = sys2.histogram(b)
            cb2 = sys2.codebook(f2, hf.HF_MODE_HUFFMAN)
            out["b"] = sys2.encode_with(b, cb2, sync=False)
    c1.sync()
    c2.sync()
    torch.cuda.synchronize()
    for key, data, ctx in (("a", a, c1), ("b", b, c2)):
        st = out[key]
        nch = (data.numel() + 16383) // 16384
        st.total_words = int(st.chunk_off[nch].item()) & 0xFFFFFFFF
        alone = hf.EntropyAwareHuffmanSystem().encode(data, mode=hf.HF_MODE_HUFFMAN)
        assert st.total_words == alone.total_words
        assert torch.equal(st.payload[: st.total_words], alone.payload[: alone.total_words])
        assert torch.equal(st.chunk_off[: nch + 1], alone.chunk_off[: nch + 1])
        back = hf.ParrallelHuffmanDecoder(ctx=ctx).decode(st)
        ctx.sync()
        assert torch.equal(back.view(torch.uint8), data.view(torch.uint8))


def test_seeded_random_module_calls_against_the_oracle(hf, orc):
    # every module of the table stage at a random alphabet size 2..256, count magnitude, tie
    # density and depth limit: SymbolSort, ShannonEntropy, HuffmanTreeBuilder (nodes, children,
    # root, leaf depths), tree_adjust, EntropyAwareTreeBuilder -- all equal to the oracle's
    # (HF_SOAK_SEED / HF_SOAK_SCALE: another seed, and that many times the cases, for a soak run)
    rng = np.random.default_rng(int(os.environ.get("HF_SOAK_SEED", "1357")))
    for case in range(80 * int(os.environ.get("HF_SOAK_SCALE", "1"))):
        N = int(rng.choice([2, 3, 4, 5, 8, 16, rng.integers(2, 257), rng.integers(2, 257), 256]))
        bits = int(rng.choice([1, 2, 4, 8, 16, 30, 44]))
        f = rng.integers(0, 1 << bits, N).astype(np.uint64)
        if rng.random() < 0.3:
            f[rng.random(N) < rng.random()] = 0           # many zeros
        if rng.random() < 0.2:
            k = min(N, 60)
            f[:] = 0
            f[rng.permutation(N)[:k]] = (1 << np.arange(k).clip(max=40)).astype(np.uint64)  # deep
        W = 64 if int(f.sum()) >= (1 << 32) - 1 else int(rng.choice([32, 64]))
        srt = hf.SymbolSort(N, 64)
        out = srt(f)
        exp, ids = orc.aw_symbol_sort(f)
        assert np.array_equal(out, exp) and np.array_equal(srt.rank2sym, ids), case
        if N == 256:
            assert hf.ShannonEntropy(256, 32)(f)[0] == orc.rp_entropy_q16_u64(f), case
        t = hf.HuffmanTreeBuilder(N, W)(f)
        o = orc.aw_tree_build(f, W)
        for k in ("weights", "lefts", "rights"):
            assert np.array_equal(t[k], o[k]), (case, k)
        assert t["root"] == o["root"], case
        depths = orc.rp_leaf_depths(o["lefts"], o["rights"], o["root"], N)
        assert np.array_equal(t["depths"], depths), case
        lo = max(1, int(np.ceil(np.log2(N))))
        lmax = int(rng.integers(lo, 17))
        assert np.array_equal(hf.tree_adjust(N, lmax)(depths), orc.rp_tree_adjust(depths, lmax)[0]), case
        thr = int(rng.choice([0, 0, 1, int(rng.integers(1, 1 << bits))]))
        code, lens = hf.EntropyAwareTreeBuilder(N, 64)(f, tree_depth_limit=lmax,
                                                       min_freq_threshold=thr)
        ec, el = orc.rp_codebook(f, 64, thr, lmax)
        assert np.array_equal(code, ec) and np.array_equal(lens, el), (case, N, lmax, thr)


@pytest.mark.parametrize("blocks", [False, True])
def test_flipped_payload_bits_never_break_the_decoder(hf, blocks):
    # random bit flips in the payload: the decoder's loops are counted, its reads clamped to the
    # payload and its writes fixed by position, so the call ends with either a corrupt-stream
    # report or wrong symbols in the right places -- and the context decodes the intact stream
    # afterwards as if nothing had happened
    rng = np.random.default_rng(8)
    sym = np.minimum(rng.geometric(0.05, 20 * 16384 + 777) - 1, 255).astype(np.uint8)
    t = dev_u8(sym)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN) if blocks else sys_.encode(t, mode=hf.HF_MODE_HUFFMAN)
    run = dec.decode_blocks if blocks else dec.decode
    good = s.payload.clone()
    for trial in range(6):
        s.payload.copy_(good)
        words = torch.from_numpy(rng.integers(0, s.total_words, 64)).to(s.payload.device)
        masks = torch.from_numpy((1 << rng.integers(0, 31, 64)).astype(np.int32)).to(s.payload.device)
        s.payload[words] ^= masks
        try:
            out = run(s)
            assert out.numel() == sym.size
        except hf.HuffhipError as e:
            assert e.code == -7
    s.payload.copy_(good)
    assert torch.equal(run(s), t)


def test_errors_of_enqueue_only_calls_are_kept_until_reported(hf):
    # a decode that only enqueues work (check = 0) on a damaged stream leaves its error in the
    # context; later enqueue-only calls do not wipe it, hf_sync reports it once and clears it
    sym = skewed_symbols(50000, 2, "geometric")
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    s = sys_.encode(dev_u8(sym), mode=hf.HF_MODE_HUFFMAN)
    good = int(s.chunk_off[1].item())
    s.chunk_off[1] = 0x7FFFFFF0
    dec.decode(s, check_stream=False)      # damaged, not reported yet
    s.chunk_off[1] = good
    out = dec.decode(s, check_stream=False)  # intact, enqueue only: must not clear the error
    with pytest.raises(hf.HuffhipError) as ei:
        sys_.ctx.sync()
    assert ei.value.code == -7
    sys_.ctx.sync()                          # reported once
    assert np.array_equal(out.cpu().numpy(), sym)
    assert np.array_equal(dec.decode(s).cpu().numpy(), sym)


@pytest.mark.parametrize("blocks", [False, True])
@pytest.mark.parametrize("bf16", [False, True])
def test_decode_from_a_payload_buffer_of_exactly_the_stream_size(hf, orc, blocks, bf16):
    # the decoder fetches the payload 16 bytes at a time; a buffer of exactly total_words words
    # with total_words % 4 in {1, 2, 3} ends in a partial group, which must be read word by word
    # (and nothing behind it: the words after the buffer's end are poisoned here)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    seen = set()
    for k in range(60):
        n = 2 * 16384 + 900 + 37 * k
        if bf16:
            w = gaussian_bf16(n, seed=100 + k)
            t = dev_bf16(w)
        else:
            t = dev_u8(skewed_symbols(n, 100 + k, "geometric"))
        s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN) if blocks \
            else sys_.encode(t, mode=hf.HF_MODE_HUFFMAN)
        rem = s.total_words % 4
        if rem == 0 or rem in seen:
            continue
        seen.add(rem)
        buf = torch.full((s.total_words + 64,), -1, dtype=torch.int32, device=DEV)
        buf[: s.total_words] = s.payload[: s.total_words]
        s.payload = buf[: s.total_words]
        assert s.payload.numel() % 4 == rem and s.payload.data_ptr() % 16 == 0
        out = dec.decode(s)
        assert torch.equal(out.view(torch.uint8), t.view(torch.uint8)), (rem, n)
        if blocks:  # decode-on-read of the last full chunk and the first, from the same buffer
            ids = torch.tensor([1, 0], dtype=torch.int32, device=DEV)
            part = dec.decode_blocks(s, ids)
            tv = t.view(torch.uint8).view(-1)
            eb = 2 if bf16 else 1
            exp = torch.cat([tv[16384 * eb:2 * 16384 * eb], tv[:16384 * eb]])
            assert torch.equal(part.view(torch.uint8).view(-1), exp)
        if seen == {1, 2, 3}:
            break
    assert seen == {1, 2, 3}


def test_strided_views_are_made_contiguous_or_refused(hf, orc):
    # the library walks memory from data_ptr(): a strided input is copied first, a strided output
    # is refused (it could not be written through)
    base = dev_u8(skewed_symbols(2 * 40000, 77, "geometric"))
    view = base[::2]
    assert not view.is_contiguous()
    sym = view.cpu().numpy()
    sys_ = hf.EntropyAwareHuffmanSystem()
    freq = hf.SymbolStat(ctx=sys_.ctx)(view)
    assert np.array_equal(freq, orc.aw_symbol_stat(sym, 256, 32))
    f2 = sys_.histogram(view).cpu().numpy().astype(np.uint64)
    assert np.array_equal(f2, orc.aw_symbol_stat(sym, 256, 64))
    cb = sys_.codebook(sys_.histogram(view), hf.HF_MODE_HUFFMAN)
    s = sys_.encode_with(view, cb)
    code, lens, _, _ = cb.download()
    payload, _ = orc.rp_encode(sym, code, lens)
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    assert np.array_equal(dec.decode(s).cpu().numpy(), sym)
    out = torch.empty(2 * sym.size, dtype=torch.uint8, device=DEV)[::2]
    with pytest.raises(ValueError):
        dec.decode(s, out=out)


def test_default_payload_buffer_grows_for_a_foreign_codebook(hf, orc):
    # a table built for one distribution used on the opposite one: more than 8 bits per symbol,
    # which the default buffer does not hold; the host API answers the overflow report with the
    # worst-case size instead of failing
    f = np.sort(fib_counts())[::-1].copy()
    code_r, len_r = orc.rp_codebook(f, 64, 0, 16)
    mw = hf.MemWrapper()
    for s_ in range(256):
        mw.write(s_, int(code_r[s_]), int(len_r[s_]))
    cb = mw.to_device()
    sym = np.random.default_rng(10).integers(40, 256, 5 * 16384 + 9, dtype=np.uint8)
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode_with(dev_u8(sym), cb)
    code, lens, _ = mw.bulk_read()
    payload, off = orc.rp_encode(sym, code, lens)
    assert s.total_words == payload.size > sym.size // 4 + 6 * 192 + 64
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    assert np.array_equal(hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s).cpu().numpy(), sym)


def test_a_call_leaves_the_callers_current_device_alone(hf):
    # with one GPU there is nothing to move; the call must still leave device 0 current
    sys_ = hf.EntropyAwareHuffmanSystem()
    sys_.encode(dev_u8(skewed_symbols(5000, 3, "geometric")), mode=hf.HF_MODE_HUFFMAN)
    assert torch.cuda.current_device() == 0
    if torch.cuda.device_count() > 1:
        ctx1 = hf.Context(1)
        assert torch.cuda.current_device() == 0
        t1 = torch.randint(0, 256, (70000,), dtype=torch.uint8, device="cuda:1")
        s1 = hf.EntropyAwareHuffmanSystem(ctx=ctx1).encode(t1, mode=hf.HF_MODE_HUFFMAN)
        assert torch.cuda.current_device() == 0
        assert torch.equal(hf.ParrallelHuffmanDecoder(ctx=ctx1).decode(s1), t1)
        assert torch.cuda.current_device() == 0


def test_codebook_with_200_second_level_tables(hf, orc):
    # an uploaded prefix code in which 200 symbols have 16-bit codes with 200 different 12-bit
    # prefixes: 200 second-level tables, of which the decoder keeps the first 64 in LDS and reads
    # the others from the device codebook
    code = np.zeros(256, np.uint16)
    lens = np.zeros(256, np.uint8)
    code[0], lens[0] = 0b0, 1
    code[1], lens[1] = 0b10, 2
    for i in range(200):
        code[2 + i], lens[2 + i] = ((0xC00 + i) << 4) | (i & 0xF), 16
    for j in range(54):
        code[202 + j], lens[202 + j] = 0xC00 + 200 + j, 12
    mw = hf.MemWrapper()
    for s_ in range(256):
        mw.write(s_, int(code[s_]), int(lens[s_]))
    cb = mw.to_device()
    rng = np.random.default_rng(77)
    sym = rng.choice(256, 5 * 16384 + 1234, p=np.r_[0.3, 0.2, np.full(254, 0.5 / 254)]).astype(np.uint8)
    sym[:256] = np.arange(256)
    sys_ = hf.EntropyAwareHuffmanSystem()
    bound = int(hf.L.lib().hf_payload_bound_words(sym.size))
    s = sys_.encode_with(dev_u8(sym), cb, payload=torch.empty(bound, dtype=torch.int32, device=DEV))
    payload, off = orc.rp_encode(sym, code, lens)
    assert s.total_words == payload.size
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    assert np.array_equal(hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s).cpu().numpy(), sym)


@pytest.mark.parametrize("bf16", [False, True])
def test_streams_that_mix_pair_hits_misses_and_long_codes(hf, orc, bf16):
    # a short average code (decoded two symbols per lookup) with rare symbols sprinkled in, so that
    # within one wavefront some lanes resolve a pair in one lookup, some need the second lookup and
    # some meet a code longer than the first-level table; whole payload against the oracle
    rng = np.random.default_rng(31)
    n = 9 * 16384 + 777
    sym = np.minimum(rng.geometric(0.55, n) - 1, 255).astype(np.uint8)
    rare = rng.random(n) < 0.01
    sym[rare] = rng.integers(20, 256, int(rare.sum()), dtype=np.uint8)
    if bf16:
        low = rng.integers(0, 128, n).astype(np.uint16)
        w = (rng.integers(0, 2, n).astype(np.uint16) << 15) | (sym.astype(np.uint16) << 7) | low
        s2, side = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
        s = system_roundtrip(hf, orc, dev_bf16(w), s2, mode=hf.HF_MODE_HUFFMAN, side=side)
    else:
        s = system_roundtrip(hf, orc, dev_u8(sym), sym, mode=hf.HF_MODE_HUFFMAN)
    lens = s.codebook.download()[1]
    assert lens.max() > 12 and float((np.bincount(sym, minlength=256) * lens).sum()) / n < 3.0
