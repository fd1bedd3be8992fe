"""GPU parity of the front ends (fp16: 5-bit exponent symbols; fp8 E4M3: sign and exponent
symbols; packed fp4 E2M1 pairs: their two sign-and-exponent fields; bytes transposed into bit
planes) against the oracle's symbolize / desymbolize, and of the whole path behind them:
the symbol stream goes through the same histogram, codebook, encoder and decoder as any byte
stream, so payload and codebook are compared with the oracle's for the oracle's symbols."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def fp16_weights(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n) * 0.02).astype(np.float16).view(np.uint16)


def fp8_bytes(n, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(n, generator=g, device=DEV)
    return x.to(torch.float8_e4m3fn).view(torch.uint8).cpu().numpy()


def to_dev(a):
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16)).to(DEV).view(torch.float16)
    return torch.from_numpy(a).to(DEV)


@pytest.mark.parametrize("n", [1, 7, 8, 9, 2047, 2048, 2049, 16384, 5 * 16384 + 13, 1 << 20])
@pytest.mark.parametrize("fp16", [True, False])
def test_split_and_merge_against_the_oracle(hf, orc, n, fp16):
    layout = hf.HF_LAYOUT_FP16_EXP if fp16 else hf.HF_LAYOUT_FP8_E4M3_EXP
    olay = orc.LAYOUT_FP16_EXP if fp16 else orc.LAYOUT_FP8_E4M3_EXP
    rng = np.random.default_rng(n)
    # every bit pattern occurs, not only well-behaved values
    w = rng.integers(0, 65536, n).astype(np.uint16) if fp16 else rng.integers(0, 256, n).astype(np.uint8)
    sys_ = hf.EntropyAwareHuffmanSystem()
    sym, side = sys_.split_fields(to_dev(w), layout)
    esym, eside = orc.symbolize(w, olay)
    assert side.numel() == eside.size == hf.side_bytes(n, layout) == ((n + 7) // 8) * (11 if fp16 else 3)
    assert np.array_equal(sym.cpu().numpy(), esym)
    assert np.array_equal(side.cpu().numpy(), eside)
    s = hf.EncodedStream(n=n, layout=layout, dtype=torch.float16 if fp16 else torch.uint8,
                         payload=None, chunk_off=None, side=side, codebook=None)
    back = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)._merge_fields(s, sym, None)
    got = back.view(torch.int16).cpu().numpy().view(np.uint16) if fp16 else back.cpu().numpy()
    assert np.array_equal(got, w)


@pytest.mark.parametrize("blocks", [False, True])
@pytest.mark.parametrize("fp16", [True, False])
def test_whole_path_through_a_sub_field_layout(hf, orc, fp16, blocks):
    n = 7 * 16384 + 333
    layout = hf.HF_LAYOUT_FP16_EXP if fp16 else hf.HF_LAYOUT_FP8_E4M3_EXP
    olay = orc.LAYOUT_FP16_EXP if fp16 else orc.LAYOUT_FP8_E4M3_EXP
    w = fp16_weights(n, 3) if fp16 else fp8_bytes(n, 4)
    t = to_dev(w)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    esym, eside = orc.symbolize(w, olay)
    assert esym.max() < 32
    if blocks:
        s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, layout=layout)
        payload, off, _, _ = orc.rp_encode_blocks(esym)
    else:
        s = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, layout=layout)
        code, lens, mode, _ = s.codebook.download()
        ec, el = orc.rp_system_codebook(orc.aw_symbol_stat(esym, 256, 64), 64, 0, 16)
        assert mode == 1 and np.array_equal(code, ec) and np.array_equal(lens, el)
        payload, off = orc.rp_encode(esym, ec, el)
    assert s.layout == layout and s.total_words == payload.size
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    assert np.array_equal(s.side.cpu().numpy(), eside)
    out = dec.decode(s)
    assert out.dtype == t.dtype and torch.equal(out.view(torch.uint8), t.view(torch.uint8))
    # the stored form carries the layout and the packed side fields
    blob = s.to_container()
    assert blob.numel() == s.container_bytes() == orc.container_bytes(
        n, olay, 1 if blocks else 0, s.total_words)
    s2 = hf.EncodedStream.from_container(blob)
    assert s2.layout == layout and s2.side.numel() == eside.size
    assert torch.equal(dec.decode(s2).view(torch.uint8), t.view(torch.uint8))


def test_sub_field_coding_against_whole_byte_coding_on_kv_pages(hf):
    # the fp8 KV-page tensor of the per-page benchmark: sign+exponent symbols with raw mantissa
    # bits against whole-byte symbols.  Whole bytes code the joint distribution, so they must
    # not lose; the sub-field split stays within a few per cent and its alphabet is 32 symbols.
    import bench_configs
    pages = 512
    t = bench_configs.make_kv_pages(torch, pages, torch.device(DEV))
    sys_ = hf.EntropyAwareHuffmanSystem()
    whole = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN)
    split = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, layout=hf.HF_LAYOUT_FP8_E4M3_EXP)
    assert torch.equal(hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(split), t)
    r_whole, r_split = whole.ratio(), split.ratio()
    print(f"fp8 KV pages: whole-byte symbols ratio {r_whole:.4f}, sign+exponent symbols with raw "
          f"mantissa bits ratio {r_split:.4f}")
    assert r_split <= r_whole * 1.001 and r_split > r_whole * 0.93
    assert int((split.codebook.download()[1][32:] != 0).sum()) >= 0  # symbols >= 32 never occur
    # fp16 weights: exponent symbols + 11 raw bits against the 8 + 8 split
    w = torch.randn(1 << 22, device=DEV).mul_(0.02).to(torch.float16)
    a = sys_.encode(w, mode=hf.HF_MODE_HUFFMAN)
    b = sys_.encode(w, mode=hf.HF_MODE_HUFFMAN, layout=hf.HF_LAYOUT_FP16_EXP)
    print(f"fp16 weights: 8+8 split ratio {a.ratio():.4f}, 5-bit exponent symbols ratio {b.ratio():.4f}")
    assert abs(a.ratio() - b.ratio()) < 0.03
    assert torch.equal(hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(b).view(torch.int16),
                       w.view(torch.int16))


def test_field_layouts_are_refused_where_they_do_not_belong(hf):
    L = hf.L
    lib = L.lib()
    ctx = hf.Context(0)
    t = torch.zeros(64, dtype=torch.uint8, device=DEV)
    f = torch.zeros(256, dtype=torch.int64, device=DEV)
    # the coding kernels take symbol bytes: a sub-field layout there is an argument error
    assert lib.hf_symbol_stat_dev(ctx.handle, C.c_void_p(t.data_ptr()), 64, hf.HF_LAYOUT_FP16_EXP,
                                  C.c_void_p(f.data_ptr())) == L.HF_ERR_ARG
    assert lib.hf_fields_split(ctx.handle, C.c_void_p(t.data_ptr()), 64, hf.HF_LAYOUT_U8,
                               C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr())) == L.HF_ERR_ARG
    assert lib.hf_fields_side_bytes(17, hf.HF_LAYOUT_FP16_EXP) == 33
    assert lib.hf_fields_side_bytes(17, hf.HF_LAYOUT_FP8_E4M3_EXP) == 9
    assert lib.hf_fields_side_bytes(17, hf.HF_LAYOUT_U8) == 0


# ---- packed fp4 pairs and bit planes ----------------------------------------------------------
@pytest.mark.parametrize("n", [1, 7, 8, 9, 2047, 2048, 2049, 16384, 5 * 16384 + 13, 1 << 20])
def test_fp4_pair_split_and_merge_against_the_oracle(hf, orc, n):
    w = np.random.default_rng(n + 1).integers(0, 256, n).astype(np.uint8)
    sys_ = hf.EntropyAwareHuffmanSystem()
    sym, side = sys_.split_fields(to_dev(w), hf.HF_LAYOUT_FP4_E2M1_EXP)
    esym, eside = orc.symbolize(w, orc.LAYOUT_FP4_E2M1_EXP)
    assert side.numel() == eside.size == hf.side_bytes(n, hf.HF_LAYOUT_FP4_E2M1_EXP) == ((n + 7) // 8) * 2
    assert np.array_equal(sym.cpu().numpy(), esym) and esym.max() < 64
    assert np.array_equal(side.cpu().numpy(), eside)
    s = hf.EncodedStream(n=n, layout=hf.HF_LAYOUT_FP4_E2M1_EXP, dtype=torch.uint8, payload=None,
                         chunk_off=None, side=side, codebook=None)
    back = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)._merge_fields(s, sym, None)
    assert np.array_equal(back.cpu().numpy(), w)


# sizes around the block of 131072 bytes: whole blocks, a short last block whose plane length is
# and is not a multiple of 4, a last block of a single group
@pytest.mark.parametrize("n", [8, 16, 24, 40, 4096, 131072 - 8, 131072, 131072 + 8, 131072 + 40,
                               3 * 131072 + 16384, 5 * 131072 + 8 * 1001, 1 << 22])
def test_bit_plane_transform_against_the_oracle(hf, orc, n):
    w = np.random.default_rng(n).integers(0, 256, n).astype(np.uint8)
    sys_ = hf.EntropyAwareHuffmanSystem()
    sym, side = sys_.split_fields(to_dev(w), hf.HF_LAYOUT_BITPLANE8)
    assert side is None and hf.side_bytes(n, hf.HF_LAYOUT_BITPLANE8) == 0
    esym, _ = orc.symbolize(w, orc.LAYOUT_BITPLANE8)
    assert np.array_equal(sym.cpu().numpy(), esym)
    s = hf.EncodedStream(n=n, layout=hf.HF_LAYOUT_BITPLANE8, dtype=torch.uint8, payload=None,
                         chunk_off=None, side=None, codebook=None)
    back = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)._merge_fields(s, sym, None)
    assert np.array_equal(back.cpu().numpy(), w)


def test_bit_plane_sizes_that_are_not_whole_groups_are_refused(hf):
    L = hf.L
    ctx = hf.Context(0)
    t = torch.zeros(64, dtype=torch.uint8, device=DEV)
    o = torch.zeros(64, dtype=torch.uint8, device=DEV)
    for n in (1, 7, 9, 63):
        assert L.lib().hf_fields_split(ctx.handle, C.c_void_p(t.data_ptr()), n, hf.HF_LAYOUT_BITPLANE8,
                                       C.c_void_p(o.data_ptr()), None) == L.HF_ERR_ARG
        assert L.lib().hf_fields_merge(ctx.handle, C.c_void_p(t.data_ptr()), None, n,
                                       hf.HF_LAYOUT_BITPLANE8, C.c_void_p(o.data_ptr())) == L.HF_ERR_ARG
    info = L.hf_container_info()
    assert L.lib().hf_container_plan(9, hf.HF_LAYOUT_BITPLANE8, 0, 4, C.byref(info)) == L.HF_ERR_ARG
    assert L.lib().hf_container_plan(16, hf.HF_LAYOUT_BITPLANE8, 0, 4, C.byref(info)) == L.HF_OK
    assert info.off_side == 0


@pytest.mark.parametrize("blocks", [False, True])
@pytest.mark.parametrize("which", ["fp4", "planes"])
def test_whole_path_through_fp4_pairs_and_bit_planes(hf, orc, which, blocks):
    n = 9 * 16384 + 8 * 41
    layout = hf.HF_LAYOUT_FP4_E2M1_EXP if which == "fp4" else hf.HF_LAYOUT_BITPLANE8
    olay = orc.LAYOUT_FP4_E2M1_EXP if which == "fp4" else orc.LAYOUT_BITPLANE8
    rng = np.random.default_rng(5)
    if which == "fp4":
        # two fp4 E2M1 codes per byte, magnitudes skewed towards zero as quantised weights are
        mag = np.minimum(rng.geometric(0.45, 2 * n) - 1, 7).astype(np.uint8)
        nib = mag | (rng.integers(0, 2, 2 * n).astype(np.uint8) << 3)
        w = (nib[0::2] | (nib[1::2] << 4)).astype(np.uint8)
    else:
        # small unsigned integers: the upper planes are all zero
        w = np.minimum(rng.geometric(0.2, n) - 1, 255).astype(np.uint8)
    t = to_dev(w)
    sys_ = hf.EntropyAwareHuffmanSystem()
    dec = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx)
    esym, eside = orc.symbolize(w, olay)
    if blocks:
        s = sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, layout=layout)
        payload, off, _, _ = orc.rp_encode_blocks(esym)
    else:
        s = sys_.encode(t, mode=hf.HF_MODE_HUFFMAN, layout=layout)
        code, lens, mode, _ = s.codebook.download()
        ec, el = orc.rp_system_codebook(orc.aw_symbol_stat(esym, 256, 64), 64, 0, 16)
        assert mode == 1 and np.array_equal(code, ec) and np.array_equal(lens, el)
        payload, off = orc.rp_encode(esym, ec, el)
    assert s.layout == layout and s.total_words == payload.size
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    if eside is None:
        assert s.side is None
    else:
        assert np.array_equal(s.side.cpu().numpy(), eside)
    assert torch.equal(dec.decode(s), t)
    blob = s.to_container()
    assert blob.numel() == s.container_bytes() == orc.container_bytes(
        n, olay, 1 if blocks else 0, s.total_words)
    s2 = hf.EncodedStream.from_container(blob)
    assert s2.layout == layout and (s2.side is None) == (eside is None)
    assert torch.equal(dec.decode(s2), t)
    print(f"{which} {'per-block' if blocks else 'stream'}: ratio {s.ratio():.4f}, whole-byte symbols "
          f"{(sys_.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN) if blocks else sys_.encode(t, mode=hf.HF_MODE_HUFFMAN)).ratio():.4f}")
