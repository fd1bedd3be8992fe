"""GPU parity of the container (wire format v1): the blob written by the library equals the
oracle's byte for byte, and a stream opened from a blob decodes in place."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import gaussian_bf16, skewed_symbols

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CH = 16384


def kv_pages_fp8(n_pages, seed=7):
    g = torch.Generator().manual_seed(seed)
    scale = torch.exp(torch.randn(1024, generator=g) * 0.7)
    x = torch.randn(n_pages * 16, 1024, generator=g) * scale
    return x.to(torch.float8_e4m3fn).view(torch.uint8).reshape(-1).numpy()


@pytest.mark.parametrize("n", [0, 5, 16384, 5 * 16384 + 13, 300001])
def test_stream_codebook_container_bf16(hf, orc, n):
    w = gaussian_bf16(n, seed=n + 3)
    t = torch.from_numpy(w.view(np.int16)).to(DEV).view(torch.bfloat16)
    s = hf.EntropyAwareHuffmanSystem().encode(t, mode=hf.HF_MODE_HUFFMAN)
    blob = s.to_container()
    # the oracle's container from the oracle's own stream
    sym, side = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
    freq = orc.aw_symbol_stat(sym, 256, 64)
    code, lens = orc.rp_system_codebook(freq, 64, 0, 16)
    payload, off = orc.rp_encode(sym, code, lens)
    exp = orc.container_pack(n, orc.LAYOUT_BF16_EXP, payload, off, side=side, code=code, lens=lens,
                             mode=1, entropy_q16=orc.rp_entropy_q16_u64(freq))
    got = blob.cpu().numpy()
    assert got.size == exp.size and np.array_equal(got, exp)
    # open it: sections are used in place, the codebook is rebuilt
    s2 = hf.EncodedStream.from_container(blob)
    assert (s2.n, s2.layout, s2.total_words) == (n, hf.HF_LAYOUT_BF16_EXP, payload.size)
    c2, l2, m2, e2 = s2.codebook.download()
    assert np.array_equal(c2, code) and np.array_equal(l2, lens) and m2 == 1
    assert e2 == orc.rp_entropy_q16_u64(freq)
    out = hf.ParrallelHuffmanDecoder().decode(s2)
    assert torch.equal(out.view(torch.int16), t.view(torch.int16))


@pytest.mark.parametrize("kind,n", [("geometric", 70000), ("uniform", 16385), ("one", 40000)])
def test_stream_codebook_container_u8(hf, orc, kind, n):
    sym = skewed_symbols(n, 5, kind)
    t = torch.from_numpy(sym).to(DEV)
    s = hf.EntropyAwareHuffmanSystem().encode(t, mode=hf.HF_MODE_AUTO)
    code, lens, mode, ent = s.codebook.download()
    payload, off = orc.rp_encode(sym, code, lens)
    exp = orc.container_pack(n, orc.LAYOUT_U8, payload, off, code=code, lens=lens, mode=mode,
                             entropy_q16=ent)
    blob = s.to_container()
    assert np.array_equal(blob.cpu().numpy(), exp)
    s2 = hf.EncodedStream.from_container(blob)
    assert torch.equal(hf.ParrallelHuffmanDecoder().decode(s2), t)


@pytest.mark.parametrize("mode", [1, 2])
def test_per_chunk_container_fp8_pages(hf, orc, mode):
    n_pages = 37
    sym = kv_pages_fp8(n_pages, seed=11)
    sym = np.concatenate([sym, sym[:777]])  # ragged last block
    n = sym.size
    nch = n_pages + 1
    t = torch.from_numpy(sym).to(DEV)
    sys_ = hf.EntropyAwareHuffmanSystem()
    s = sys_.encode_blocks(t, mode=mode, layout=hf.HF_LAYOUT_FP8)
    blob = s.to_container()
    # oracle: per-block canonical orders, payload block by block
    r2s = np.zeros((nch, 256), np.uint8)
    lbr = np.zeros((nch, 256), np.uint8)
    modes = np.zeros(nch, np.uint32)
    ents = np.zeros(nch, np.uint32)
    payloads, offs, total = [], [0], 0
    for c in range(nch):
        blk = sym[c * CH:(c + 1) * CH]
        f = orc.aw_symbol_stat(blk, 256, 32)
        ents[c] = orc.rp_entropy_q16_u64(f)
        modes[c] = orc.aw_adaptive_mode(int(ents[c])) if mode == 2 else mode
        if modes[c] == 1:
            code, lens = orc.rp_system_codebook(f, 64, 0, 16)
            rank2sym = orc.aw_symbol_sort(f)[1]
        else:
            code, lens = np.arange(256, dtype=np.uint16), np.full(256, 8, np.uint8)
            rank2sym = np.arange(256)
        r2s[c] = rank2sym
        lbr[c] = lens[rank2sym]
        p, _ = orc.rp_encode(blk, code, lens)
        payloads.append(p)
        total += p.size
        offs.append(total)
    exp = orc.container_pack(n, orc.LAYOUT_FP8, np.concatenate(payloads),
                             np.array(offs, np.uint32), rank2sym=r2s, len_by_rank=lbr,
                             modes=modes, entropies=ents)
    got = blob.cpu().numpy()
    assert got.size == exp.size and np.array_equal(got, exp)
    # reopened: rebuilt records equal the encoder's, decode whole and on read
    s2 = hf.EncodedStream.from_container(blob)
    a = s.chunk_codebooks.view(-1, hf.L.HF_CHUNK_CODEBOOK_BYTES)[:nch].cpu().numpy()
    b = s2.chunk_codebooks.view(-1, hf.L.HF_CHUNK_CODEBOOK_BYTES)[:nch].cpu().numpy()
    assert np.array_equal(a[:, :1544], b[:, :1544])
    dec = hf.ParrallelHuffmanDecoder()
    assert torch.equal(dec.decode(s2), t)
    ids = torch.tensor([30, 2, 17, 2, 0], dtype=torch.int32, device=DEV)
    part = dec.decode_blocks(s2, chunk_ids=ids)
    assert torch.equal(part, t[: n_pages * CH].view(n_pages, CH)[ids.long()].reshape(-1))


def test_bad_containers_are_rejected(hf, orc):
    from huffman_amd import _lib as L
    sym = skewed_symbols(50000, 2, "geometric")
    t = torch.from_numpy(sym).to(DEV)
    sys_ = hf.EntropyAwareHuffmanSystem()
    blob = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN).to_container()
    ctx = hf.default_context()
    info = L.hf_container_info()

    def open_rc(b, nbytes=None):
        return L.lib().hf_container_open(ctx.handle, C.c_void_p(b.data_ptr()),
                                         b.numel() if nbytes is None else nbytes, C.byref(info))

    assert open_rc(blob) == 0
    assert open_rc(blob, blob.numel() - 16) == L.HF_ERR_CORRUPT     # truncated
    assert open_rc(blob, 64) == L.HF_ERR_CORRUPT
    for pos in (0, 8, 16, 24, 48, 72, 96, 120):                     # magic, version, ... reserved
        bad = blob.clone()
        bad[pos] = bad[pos] + 1
        assert open_rc(bad) == L.HF_ERR_CORRUPT, pos
    # a per-chunk record whose length counts do not describe 256 ranks
    blk = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN).to_container()
    assert open_rc(blk) == 0
    bad = blk.clone()
    bad[int(info.off_codebook) + 256 + 2 * 7] += 1
    assert open_rc(bad) == 0
    cbs = torch.empty(int(info.nchunks) * L.HF_CHUNK_CODEBOOK_BYTES, dtype=torch.uint8, device=DEV)
    rc = L.lib().hf_container_codebook(ctx.handle, C.c_void_p(bad.data_ptr()), C.byref(info),
                                       C.c_void_p(cbs.data_ptr()))
    assert rc == L.HF_ERR_CORRUPT


def test_sharded_codec_single_rank_and_its_directory(hf):
    # the multi-rank helper on one rank (no process group): same stream, a one-row directory whose
    # stored size is the size of the container actually packed
    from huffman_amd.dist import ShardedCodec
    w = torch.from_numpy(gaussian_bf16(3 * 16384 + 999, seed=31).view(np.int16)).to(DEV).view(torch.bfloat16)
    codec = ShardedCodec(global_codebook=True)
    s = codec.encode(w, mode=hf.HF_MODE_HUFFMAN)
    d = codec.directory(s, lo=0)
    assert tuple(d.shape) == (1, 4)
    assert d[0].tolist() == [0, w.numel(), s.total_words, s.to_container().numel()]
    assert torch.equal(codec.decode(s).view(torch.int16), w.view(torch.int16))
