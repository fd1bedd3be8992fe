"""CPU suite: the oracle against the reference's own known-answer vectors, against the committed
golden vectors, and against the invariants of the repaired layers."""
import json
import os

import numpy as np
import pytest

from conftest import gaussian_bf16, skewed_symbols

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kats():
    return json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(HERE, "golden", "oracle_vectors.json")))


def kraft(lens):
    return sum(2.0 ** -int(l) for l in lens)


# ---- the reference's own vectors -----------------------------------------------------------
def test_reference_symbol_stat_kats(orc, kats):
    for k in kats["symbol_stat"]:
        p = k["params"]
        got = orc.aw_symbol_stat(np.array(k["inputs"], np.uint8), p["depth"], p["wtWidth"])
        assert list(got) == k["freq_out"], k["source"]


def test_reference_symbol_sort_kats(orc, kats):
    for k in kats["symbol_sort"]:
        got, _ = orc.aw_symbol_sort(np.array(k["inputs"], np.uint64))
        assert list(got) == k["sorted_out"], k["source"]


def test_reference_tree_builder_expectation_is_the_dead_depths_port(orc, kats):
    # the all-zero expectation equals what the never-written leaf entries of depthArr give
    k = kats["non_normative"][0]
    t = orc.aw_tree_build(k["inputs"], 8)
    assert not t["heights"][:4].any()           # the RTL's `depths` output: all zero
    d = orc.rp_leaf_depths(t["lefts"], t["rights"], t["root"], 4)
    assert list(d) == [2, 2, 2, 2]              # the real tree


# ---- committed golden vectors ----------------------------------------------------------------
def test_golden_trees(orc, gold):
    for t in gold["trees"]:
        W = t.get("W", 32)
        N = len(t["freqs"])
        o = orc.aw_tree_build(t["freqs"], W)
        assert list(o["lefts"]) == t["lefts"] and list(o["rights"]) == t["rights"]
        assert list(o["weights"]) == t["weights"] and o["root"] == t["root"]
        assert list(o["heights"]) == t["heights"]
        if "leaf_depths" in t:
            assert list(orc.rp_leaf_depths(o["lefts"], o["rights"], o["root"], N)) == t["leaf_depths"]


def test_golden_entropy(orc, gold):
    for e in gold["entropy"]:
        W = e.get("W", 32)
        assert orc.aw_entropy_q16(e["freq"], W) == e["entropy_q16"], e["name"]
        if "mode" in e:
            assert orc.aw_adaptive_mode(e["entropy_q16"]) == e["mode"]


def test_entropy_decision_thresholds(orc):
    # adaptive_threshold.scala:26-27,65-79: below 0.5 -> Huffman, above 6.0 -> bypass, else Huffman
    assert orc.aw_adaptive_mode(0x7FFF) == 1
    assert orc.aw_adaptive_mode(0x8000) == 1
    assert orc.aw_adaptive_mode(0x60000) == 1
    assert orc.aw_adaptive_mode(0x60001) == 0


def test_golden_tree_adjust(orc, gold):
    for a in gold["tree_adjust"]:
        out, steps = orc.rp_tree_adjust(a["depths_in"], a["lmax"])
        assert list(out) == a["depths_out"] and steps == a["migrate_steps"], a["name"]


def test_golden_codebooks(orc, gold):
    for c in gold["codebooks"]:
        f = np.array(c["freq"], np.uint64)
        if c.get("system"):
            code, lens = orc.rp_system_codebook(f, 64, 0, 16)
        else:
            code, lens = orc.rp_codebook(f, 32, 0, 16)
            assert list(orc.aw_codes_as_written(lens.astype(np.uint16))) == c["code_as_written"]
        assert list(code) == c["code"] and list(lens) == c["len"], c["name"]


def test_golden_streams(orc, gold):
    for s in gold["streams"]:
        if "sym" not in s:
            continue
        sym = np.array(s["sym"], np.uint8)
        payload, off = orc.rp_encode(sym, s["code"], s["len"])
        assert list(payload) == s["payload"] and list(off) == s["chunk_off"], s["name"]
        assert np.array_equal(orc.rp_decode(payload, off, sym.size, s["code"], s["len"]), sym)


def _page_inputs():
    """The inputs of the "pages" vectors that are too long to be stored (generated as in
    tests/golden/make_golden.py)."""
    prng = np.random.default_rng(77)
    g200 = np.minimum(prng.geometric(0.3, 200) - 1, 255).astype(np.uint8)
    mixed = np.concatenate([np.minimum(prng.geometric(0.05, 16384) - 1, 255),
                            prng.integers(0, 256, 16384),
                            np.minimum(prng.geometric(0.6, 9000) - 1, 255)]).astype(np.uint8)
    return {"200 geometric": g200, "16385 ramp": (np.arange(16385) % 7).astype(np.uint8),
            "3 pages mixed": mixed}


def test_golden_pages(orc, gold):
    # paged streams: pages with their codebook records, and pages without under one codebook
    crc = lambda w: int(np.bitwise_xor.reduce(w * np.arange(1, w.size + 1, dtype=np.uint32))) \
        if w.size else 0
    made = _page_inputs()
    for p in gold["pages"]:
        sym = np.array(p["sym"], np.uint8) if "sym" in p else made[p["name"]]
        assert sym.size == p["n"]
        rp, roff, _, _ = orc.rp_encode_pages(sym, 0, 1)
        assert list(roff) == p["record_page_off"] and rp.size == p["record_words"], p["name"]
        assert crc(rp) == p["record_crc"], p["name"]
        sp, soff = orc.rp_encode_pages_shared(sym, p["shared_code"], p["shared_len"])
        assert list(soff) == p["shared_page_off"] and sp.size == p["shared_words"], p["name"]
        assert crc(sp) == p["shared_crc"], p["name"]
        if "sym" in p:
            assert list(rp) == p["record_payload"] and list(sp) == p["shared_payload"], p["name"]
        assert np.array_equal(orc.rp_decode_pages(rp, roff, sym.size), sym)
        assert np.array_equal(orc.rp_decode_pages_shared(sp, soff, sym.size, p["shared_code"],
                                                         p["shared_len"]), sym)


def test_golden_quality(orc, gold):
    q = gold["quality"]
    assert list(orc.aw_quality(q["freq"], q["len"], 32)) == q["result"]


# ---- as-written properties ---------------------------------------------------------------------
def test_symbol_stat_wraps_and_masks(orc):
    sym = np.full(300, 200, np.uint8)
    assert orc.aw_symbol_stat(sym, 256, 8)[200] == 300 % 256       # W-bit counter wraps
    assert orc.aw_symbol_stat(sym, 16, 32)[200 & 15] == 300        # index keeps log2(N) low bits


@pytest.mark.parametrize("N", [2, 4, 16, 64, 256])
def test_symbol_sort_is_stable_descending(orc, N):
    rng = np.random.default_rng(N)
    for hi in (2, 7, 1 << 30):
        f = rng.integers(0, hi, N).astype(np.uint64)
        out, ids = orc.aw_symbol_sort(f)
        order = sorted(range(N), key=lambda i: (-int(f[i]), i))   # (count desc, index asc)
        assert list(ids) == order and list(out) == [int(f[i]) for i in order]


def two_queue_tree(freqs):
    """Leaf queue ordered by (weight, index), internal queue in creation order, leaf preferred
    on ties: the formulation the GPU kernel uses."""
    N = len(freqs)
    leaves = sorted(range(N), key=lambda i: (int(freqs[i]), i))
    w = [int(x) for x in freqs] + [0] * (N - 1)
    lefts, rights = [0] * (2 * N - 1), [0] * (2 * N - 1)
    li, ii = 0, N
    for k in range(N, 2 * N - 1):
        pick = []
        for _ in range(2):
            if li < N and (ii >= k or w[leaves[li]] <= w[ii]):
                pick.append(leaves[li]); li += 1
            else:
                pick.append(ii); ii += 1
        w[k] = w[pick[0]] + w[pick[1]]
        lefts[k], rights[k] = pick
    return w, lefts, rights


@pytest.mark.parametrize("seed", range(8))
def test_tree_build_equals_two_queue_merge(orc, seed):
    rng = np.random.default_rng(seed)
    N = [4, 8, 16, 33, 64, 128, 256, 256][seed]
    hi = [2, 3, 10, 100, 4, 1 << 20, 6, 1 << 40][seed]
    f = rng.integers(0, hi, N).astype(np.uint64)
    o = orc.aw_tree_build(f, 64)
    w, l, r = two_queue_tree(f)
    assert list(o["weights"]) == w and list(o["lefts"]) == l and list(o["rights"]) == r


def test_tree_is_complete_and_depths_match_weights(orc):
    rng = np.random.default_rng(3)
    f = rng.integers(0, 1000, 256).astype(np.uint64)
    o = orc.aw_tree_build(f, 64)
    d = orc.rp_leaf_depths(o["lefts"], o["rights"], o["root"], 256)
    assert abs(kraft(d) - 1.0) < 1e-12
    assert int(o["weights"][o["root"]]) == int(f.sum())
    assert int((f * d).sum()) == int(o["weights"][256:].sum())   # cost = sum of internal weights


def test_code_start_recurrence(orc):
    lens = np.array([1, 2, 3, 3], np.uint16)
    count, cs = orc.aw_code_start(lens)
    assert list(count[:4]) == [0, 1, 1, 2] and list(cs[:4]) == [0, 0, 2, 6]
    # as written every symbol of one length gets the same code (not prefix free)
    aw = orc.aw_codes_as_written(lens)
    assert aw[2] == aw[3]


# ---- repaired-layer invariants -------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["uniform", "geometric", "one", "two", "fib"])
@pytest.mark.parametrize("lmax", [16, 12, 9])
def test_limited_codebooks_are_prefix_codes(orc, kind, lmax):
    sym = skewed_symbols(100000, 5, kind)
    f = orc.aw_symbol_stat(sym, 256, 64)
    code, lens = orc.rp_system_codebook(f, 64, 0, lmax)
    assert lens.min() >= 1 and lens.max() <= lmax
    assert kraft(lens) <= 1.0 + 1e-12
    # prefix freedom: the left-aligned code intervals do not overlap
    iv = sorted((int(c) << (16 - int(l)), (int(c) + 1) << (16 - int(l))) for c, l in zip(code, lens))
    assert all(a[1] <= b[0] for a, b in zip(iv, iv[1:]))
    # more frequent symbols never get longer codes
    order = np.argsort(-f.astype(np.int64), kind="stable")
    assert np.all(np.diff(lens[order].astype(int)) >= 0)


def test_tree_adjust_identity_and_limits(orc):
    d = np.array([1, 2, 3, 4, 5, 6, 7, 7], np.uint16)          # already monotone and <= lmax
    out, steps = orc.rp_tree_adjust(d, 16)
    assert list(out) == list(d) and steps == 0
    out, _ = orc.rp_tree_adjust(np.array([1, 3, 2, 3], np.uint16), 16)
    assert list(out) == [1, 2, 3, 3]                            # re-dealt in index order
    with pytest.raises(ValueError):
        orc.rp_tree_adjust(np.ones(256, np.uint16), 7)          # 2^7 < 256


@pytest.mark.parametrize("n", [0, 1, 2, 9, 511, 512, 513, 16383, 16384, 16385, 70001])
def test_stream_round_trip_ragged(orc, n):
    sym = skewed_symbols(n, n + 1, "geometric") if n else np.zeros(0, np.uint8)
    f = orc.aw_symbol_stat(sym, 256, 64)
    code, lens = orc.rp_system_codebook(f, 64, 0, 16)
    payload, off = orc.rp_encode(sym, code, lens)
    assert off.size == orc.num_chunks(n) + 1 and off[-1] == payload.size
    assert payload.size <= orc.payload_bound_words(n)
    assert np.array_equal(orc.rp_decode(payload, off, n, code, lens), sym)
    # the payload holds at least the code bits and at most 64 bits more per lane and chunk
    bits = int((f * lens).sum())
    assert 32 * payload.size >= bits
    assert 32 * payload.size <= bits + 64 * 64 * max(1, orc.num_chunks(n))


def test_decode_rejects_truncated_offsets(orc):
    sym = skewed_symbols(40000, 3, "geometric")
    f = orc.aw_symbol_stat(sym, 256, 64)
    code, lens = orc.rp_system_codebook(f, 64, 0, 16)
    payload, off = orc.rp_encode(sym, code, lens)
    bad = off.copy()
    bad[1] -= 1
    with pytest.raises(ValueError):
        orc.rp_decode(payload, bad, sym.size, code, lens)


def test_bf16_front_end_round_trip(orc):
    w = gaussian_bf16(50000, 9)
    sym, side = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
    assert np.array_equal(sym, ((w >> 7) & 0xFF).astype(np.uint8))
    assert np.array_equal(side, (((w >> 8) & 0x80) | (w & 0x7F)).astype(np.uint8))
    assert np.array_equal(orc.desymbolize(sym, side, orc.LAYOUT_BF16_EXP), w)


def test_encode_pairs_bypass_and_lookup(orc):
    sym = np.array([0, 5, 255], np.uint8)
    code = np.arange(256, dtype=np.uint16) * 3
    lens = np.full(256, 11, np.uint8)
    by = orc.aw_encode_pairs(sym, code, lens, 0, 8)             # top_encoder.scala:150-153
    assert [list(r) for r in by] == [[0, 0, 8, 0], [5, 0, 8, 0], [255, 0, 8, 0]]
    hu = orc.aw_encode_pairs(sym, code, lens, 1, 8)             # top_encoder.scala:154-158
    assert [int(r[0]) | int(r[1]) << 8 for r in hu] == [0, 15, 765] and all(r[2] == 11 for r in hu)


def test_entropy_for_wide_counts_scales_down(orc):
    rng = np.random.default_rng(8)
    f = rng.integers(0, 1 << 38, 256).astype(np.uint64)
    tot = int(f.sum())
    sh = max(0, tot.bit_length() - 32)
    assert orc.rp_entropy_q16_u64(f) == orc.aw_entropy_q16(f >> np.uint64(sh), 32)


def test_sub_field_layouts_round_trip_and_pack_densely(orc):
    # fp16: symbol = 5 exponent bits, side field = sign | mantissa (11 bits); fp8 E4M3: symbol =
    # sign | exponent, side field = 3 mantissa bits; side stream = dense little-endian packing
    rng = np.random.default_rng(1)
    for n in (1, 7, 8, 9, 1000, 16385):
        w = rng.integers(0, 65536, n).astype(np.uint16)
        s, sd = orc.symbolize(w, orc.LAYOUT_FP16_EXP)
        assert np.array_equal(s, (w >> 10) & 0x1F) and sd.size == ((n + 7) // 8) * 11
        assert np.array_equal(orc.desymbolize(s, sd, orc.LAYOUT_FP16_EXP), w)
        b = rng.integers(0, 256, n).astype(np.uint8)
        s, sd = orc.symbolize(b, orc.LAYOUT_FP8_E4M3_EXP)
        assert np.array_equal(s, b >> 3) and sd.size == ((n + 7) // 8) * 3
        assert np.array_equal(orc.desymbolize(s, sd, orc.LAYOUT_FP8_E4M3_EXP), b)
    # known answers, written out by hand: 0xBC00 = -1.0 (sign 1, exponent 15, mantissa 0)
    s, sd = orc.symbolize(np.array([0xBC00, 0x3555], np.uint16), orc.LAYOUT_FP16_EXP)
    assert s.tolist() == [15, 13]
    assert int.from_bytes(sd.tobytes(), "little") == (0x400) | (0x155 << 11)
    b = np.array([1, 2, 3, 4, 5, 6, 7, 0xF8], np.uint8)
    s, sd = orc.symbolize(b, orc.LAYOUT_FP8_E4M3_EXP)
    assert s.tolist() == [0] * 7 + [31]
    assert int.from_bytes(sd.tobytes(), "little") == sum(int(x & 7) << (3 * i) for i, x in enumerate(b))
    assert orc.container_bytes(1000, orc.LAYOUT_FP8_E4M3_EXP, 0, 100) == \
        orc.container_bytes(1000, orc.LAYOUT_U8, 0, 100) + 384  # 375 side bytes, padded to 16


def test_fp4_pairs_and_bit_planes_round_trip_with_known_answers(orc):
    rng = np.random.default_rng(2)
    for n in (1, 7, 8, 9, 1000, 16385):
        b = rng.integers(0, 256, n).astype(np.uint8)
        s, sd = orc.symbolize(b, orc.LAYOUT_FP4_E2M1_EXP)
        assert np.array_equal(s, ((b >> 1) & 7) | (((b >> 5) & 7) << 3)) and sd.size == ((n + 7) // 8) * 2
        assert np.array_equal(orc.desymbolize(s, sd, orc.LAYOUT_FP4_E2M1_EXP), b)
    # by hand: 0x3B = high nibble 0011 (s 0, ee 01, m 1), low nibble 1011 (s 1, ee 01, m 1)
    s, sd = orc.symbolize(np.array([0x3B, 0xE0], np.uint8), orc.LAYOUT_FP4_E2M1_EXP)
    assert s.tolist() == [0b001_101, 0b111_000]
    assert int.from_bytes(sd.tobytes(), "little") == 0b00_11
    # bit planes: sizes around the block of 131072 bytes
    for n in (8, 40, 131072, 131072 + 24, 2 * 131072 + 16384):
        b = rng.integers(0, 256, n).astype(np.uint8)
        s, sd = orc.symbolize(b, orc.LAYOUT_BITPLANE8)
        assert sd is None and s.size == n and orc.side_bytes(n, orc.LAYOUT_BITPLANE8) == 0
        assert np.array_equal(orc.desymbolize(s, None, orc.LAYOUT_BITPLANE8), b)
        # the definition, spelled out with numpy for the first block
        m = min(n, orc.BITPLANE_BLOCK)
        bits = np.unpackbits(b[:m].reshape(-1, 8)[:, :, None], axis=2, bitorder="little")  # [g, i, p]
        planes = np.packbits(bits.transpose(2, 0, 1), axis=2, bitorder="little")[:, :, 0]   # [p, g]
        assert np.array_equal(s[:m], planes.reshape(-1))
    # by hand: the bytes 0..15 are two groups; plane 0 alternates, plane 3 is the second group only
    s, _ = orc.symbolize(np.arange(16, dtype=np.uint8), orc.LAYOUT_BITPLANE8)
    assert s.tolist() == [0xAA, 0xAA, 0xCC, 0xCC, 0xF0, 0xF0, 0x00, 0xFF] + [0] * 8
    assert orc.container_bytes(1024, orc.LAYOUT_BITPLANE8, 0, 100) == \
        orc.container_bytes(1024, orc.LAYOUT_U8, 0, 100)


def test_slotted_placement_keeps_every_chunk_inside_its_slot(orc):
    # slots are sized from the bits of a chunk's codes alone; the look-ahead words of 64 lanes must
    # fit the 126 extra words whatever the data
    rng = np.random.default_rng(77)
    ch = 16384
    cases = [rng.integers(0, 256, 3 * ch + 17).astype(np.uint8),
             np.minimum(rng.geometric(0.5, 4 * ch) - 1, 255).astype(np.uint8),
             np.full(2 * ch + 1, 9, np.uint8)]
    # lanes that differ: groups of 8 symbols alternate between a one-bit and a long code
    lanes = np.where((np.arange(5 * ch) // 8) % 2 == 0, 0, rng.integers(1, 256, 5 * ch)).astype(np.uint8)
    cases.append(lanes)
    # one lane (group index = 7 mod 64) all rare symbols, the others one symbol
    one = np.zeros(3 * ch, np.uint8)
    g = np.arange(3 * ch) // 8
    one[g % 64 == 7] = rng.integers(1, 256, int((g % 64 == 7).sum()))
    cases.append(one)
    for sym in cases:
        payload, slot_off, words, code, lens = orc.rp_encode_blocks_slotted(sym)
        dense, off, code2, lens2 = orc.rp_encode_blocks(sym)
        nch = orc.num_chunks(sym.size)
        assert np.array_equal(code, code2) and np.array_equal(lens, lens2)
        assert np.array_equal(np.diff(off.astype(np.int64)), words.astype(np.int64))
        for c in range(nch):
            part = sym[c * ch:(c + 1) * ch]
            bits = int(lens[c][part].astype(np.int64).sum())
            slot = int(slot_off[c + 1]) - int(slot_off[c])
            assert slot == orc.slot_words(bits)
            assert int(words[c]) <= bits // 32 + 124 <= slot
            a = int(slot_off[c])
            assert np.array_equal(payload[a:a + int(words[c])], dense[off[c]:off[c + 1]])
            assert not payload[a + int(words[c]):a + slot].any()
        assert payload.size == int(slot_off[nch])


def test_slotted_placement_with_one_codebook(orc):
    rng = np.random.default_rng(78)
    ch = 16384
    sym = np.concatenate([np.minimum(rng.geometric(p, ch) - 1, 255) for p in (0.6, 0.05, 0.2)]
                         + [rng.integers(0, 256, 777)]).astype(np.uint8)
    code, lens = orc.rp_system_codebook(orc.aw_symbol_stat(sym, 256, 64), 64, 0, 16)
    payload, slot_off, words = orc.rp_encode_slotted(sym, code, lens)
    dense, off = orc.rp_encode(sym, code, lens)
    nch = orc.num_chunks(sym.size)
    assert np.array_equal(np.diff(off.astype(np.int64)), words.astype(np.int64))
    for c in range(nch):
        part = sym[c * ch:(c + 1) * ch]
        bits = int(lens[part].astype(np.int64).sum())
        slot = int(slot_off[c + 1]) - int(slot_off[c])
        assert slot == orc.slot_words(bits) and int(words[c]) <= bits // 32 + 124
        a = int(slot_off[c])
        assert np.array_equal(payload[a:a + int(words[c])], dense[off[c]:off[c + 1]])
        assert not payload[a + int(words[c]):a + slot].any()
    # the chunks decode from their slots as from the dense stream
    back = np.concatenate([
        orc.rp_decode(payload[int(slot_off[c]):int(slot_off[c]) + int(words[c])],
                      np.array([0, words[c]], np.uint32), min(ch, sym.size - c * ch), code, lens)
        for c in range(nch)])
    assert np.array_equal(back, sym)
