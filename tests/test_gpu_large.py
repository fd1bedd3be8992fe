"""Streams whose payload passes 2^31 and 2^32 words: word offsets are 32-bit in the stream
(chunk_off), 64-bit inside the kernels; the first must round-trip, the second must be refused."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def random_bytes(n):
    t = torch.empty(n, dtype=torch.uint8, device=DEV)
    step = 1 << 30
    for a in range(0, n, step):
        t[a:a + step].random_(0, 256)
    return t


def test_payload_of_more_than_2_31_words_round_trips(hf):
    # 8.6e9 byte symbols sent through the bypass table (8 bits each, top_encoder.scala:131-137):
    # about 2.2e9 payload words, so tile bases and chunk offsets pass 2^31
    n = 8_600_000_000 + 12345
    t = random_bytes(n)
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode(t, mode=hf.HF_MODE_BYPASS)
    assert (1 << 31) < s.total_words < (1 << 32)
    nch = (n + 16383) // 16384
    off = s.chunk_off[: nch + 1].cpu().numpy().view(np.uint32).astype(np.int64)
    assert off[0] == 0 and off[-1] == s.total_words
    d = np.diff(off)
    # a full chunk of 8-bit codes: 64 code words and one look-ahead word in each of the 64 lanes
    assert np.all(d[:-1] == 65 * 64) and 0 < d[-1] <= 65 * 64
    out = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s)
    assert torch.equal(out, t)


def test_payload_of_more_than_2_32_words_is_refused(hf):
    # twice as many symbols: the stream would need word offsets above 32 bits; the encoder reports
    # an overflow instead of wrapping them (the buffer itself is large enough)
    n = 17_300_000_000
    t = random_bytes(n)
    sys_ = hf.EntropyAwareHuffmanSystem()
    with pytest.raises(hf.HuffhipError) as ei:
        sys_.encode(t, mode=hf.HF_MODE_BYPASS)
    assert ei.value.code == -5


def test_counts_above_32_bits(hf, orc):
    # 8.6e9 symbols of which all but a sprinkle are one value: a 64-bit count above 2^32 goes
    # through the histogram, the entropy stage (counts shifted into 32 bits first), the tree and
    # the encoder; the table must be the oracle's for the same counts and the stream round-trips
    n = 8_600_000_000
    t = torch.full((n,), 7, dtype=torch.uint8, device=DEV)
    idx = torch.arange(0, n, 1_000_003, device=DEV)
    t[idx] = (idx % 251).to(torch.uint8)
    sys_ = hf.EntropyAwareHuffmanSystem()
    freq = sys_.histogram(t).cpu().numpy().astype(np.uint64)
    exp = np.bincount((np.arange(0, n, 1_000_003) % 251).astype(np.int64), minlength=256).astype(np.uint64)
    exp[7] += n - idx.numel()
    assert np.array_equal(freq, exp) and int(freq[7]) > (1 << 32)
    s = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN)
    code, lens, m, ent = s.codebook.download()
    ec, el = orc.rp_system_codebook(freq, 64, 0, 16)
    assert ent == orc.rp_entropy_q16_u64(freq)
    assert np.array_equal(code, ec) and np.array_equal(lens, el)
    assert torch.equal(hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s), t)
