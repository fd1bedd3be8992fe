"""The entry point that allocates nothing, hf_system_codebook_chunks_static, through the C ABI:
its records must equal hf_system_codebook_chunks_dev's bit for bit, call after call on the same
static scratch block, the second call smaller than the first (stale scratch behind it)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _counts(rng, pages, kind):
    """int32[pages][256] block histograms summing to 16384, from skewed to flat."""
    out = np.zeros((pages, 256), np.int32)
    for p in range(pages):
        if kind == "mixed":
            k = int(rng.integers(1, 257))
            alpha = float(10.0 ** rng.uniform(-2.0, 1.0))
            pr = np.zeros(256)
            pr[rng.permutation(256)[:k]] = rng.dirichlet(np.full(k, alpha))
        else:
            pr = np.full(256, 1.0 / 256)
        out[p] = rng.multinomial(16384, pr)
    return out


def _records(hf, fn, ctx, freq_t, nchunks, mode, thr):
    L = hf.L
    cbs = torch.full((max(1, nchunks) * L.HF_CHUNK_CODEBOOK_BYTES,), 0xA5, dtype=torch.uint8,
                     device=DEV)
    rc = fn(ctx.handle, C.c_void_p(freq_t.data_ptr()), nchunks, mode, thr,
            C.c_void_p(cbs.data_ptr()))
    assert rc == 0, rc
    ctx.sync()
    return cbs


def test_static_scratch_entry_point_matches_the_allocating_one(hf):
    L = hf.L
    lib = L.lib()
    ctx = hf.Context(0)
    rng = np.random.default_rng(2024)
    # two calls in a row on the shared block: 1500 pages, then 300 other pages
    for pages, kind, mode, thr in ((1500, "mixed", hf.HF_MODE_HUFFMAN, 0),
                                   (300, "mixed", hf.HF_MODE_BEST, 0),
                                   (64, "flat", hf.HF_MODE_AUTO, 3),
                                   (1, "mixed", hf.HF_MODE_HUFFMAN, 0)):
        freq = torch.from_numpy(_counts(rng, pages, kind)).to(DEV).contiguous()
        got = _records(hf, lib.hf_system_codebook_chunks_static, ctx, freq, pages, mode, thr)
        exp = _records(hf, lib.hf_system_codebook_chunks_dev, ctx, freq, pages, mode, thr)
        assert torch.equal(got, exp), (pages, kind, mode)
        n = pages * L.HF_CHUNK_CODEBOOK_BYTES
        assert not torch.all(got[:n] == 0xA5)
    # counts no real block has (sum >= 2^16) take the general kernel on both paths
    big = torch.from_numpy(rng.integers(0, 1 << 20, (40, 256)).astype(np.int32)).to(DEV)
    got = _records(hf, lib.hf_system_codebook_chunks_static, ctx, big, 40, hf.HF_MODE_HUFFMAN, 0)
    exp = _records(hf, lib.hf_system_codebook_chunks_dev, ctx, big, 40, hf.HF_MODE_HUFFMAN, 0)
    assert torch.equal(got, exp)
    # the limit: 4096 pages go through, one more is refused and nothing is launched
    full = torch.from_numpy(_counts(rng, 8, "mixed")).to(DEV).repeat(512, 1).contiguous()
    assert full.shape[0] == L.HF_STATIC_CB_MAX_CHUNKS
    got = _records(hf, lib.hf_system_codebook_chunks_static, ctx, full, 4096, hf.HF_MODE_HUFFMAN, 0)
    exp = _records(hf, lib.hf_system_codebook_chunks_dev, ctx, full, 4096, hf.HF_MODE_HUFFMAN, 0)
    assert torch.equal(got, exp)
    cbs = torch.empty(16, dtype=torch.uint8, device=DEV)
    rc = lib.hf_system_codebook_chunks_static(ctx.handle, C.c_void_p(full.data_ptr()), 4097,
                                              hf.HF_MODE_HUFFMAN, 0, C.c_void_p(cbs.data_ptr()))
    assert rc == L.HF_ERR_RANGE
    assert lib.hf_system_codebook_chunks_static(ctx.handle, None, 0, hf.HF_MODE_HUFFMAN, 0,
                                                None) == 0   # an empty job is fine


def test_records_from_the_static_path_encode_and_decode(hf, orc):
    # the records are what the per-block encoder and decoder consume: a page batch coded with them
    # round-trips, and one page's table equals the oracle's
    L = hf.L
    lib = L.lib()
    sys_ = hf.EntropyAwareHuffmanSystem()
    h = sys_.ctx.handle
    rng = np.random.default_rng(5)
    pages = 48
    sym = np.minimum(rng.geometric(0.07, pages * 16384) - 1, 255).astype(np.uint8)
    t = torch.from_numpy(sym).to(DEV)
    n = t.numel()
    freq = torch.empty(pages * 256, dtype=torch.int32, device=DEV)
    cbs = torch.empty(pages * L.HF_CHUNK_CODEBOOK_BYTES, dtype=torch.uint8, device=DEV)
    assert lib.hf_symbol_stat_chunks_dev(h, C.c_void_p(t.data_ptr()), n, hf.HF_LAYOUT_U8,
                                         C.c_void_p(freq.data_ptr())) == 0
    assert lib.hf_system_codebook_chunks_static(h, C.c_void_p(freq.data_ptr()), pages,
                                                hf.HF_MODE_HUFFMAN, 0,
                                                C.c_void_p(cbs.data_ptr())) == 0
    payload = torch.empty(n // 4 + pages * 192 + 64, dtype=torch.int32, device=DEV)
    off = torch.empty(pages + 4, dtype=torch.int32, device=DEV)
    total = C.c_uint64(0)
    assert lib.hf_encode_chunks(h, C.c_void_p(t.data_ptr()), n, hf.HF_LAYOUT_U8,
                                C.c_void_p(cbs.data_ptr()), None, C.c_void_p(payload.data_ptr()),
                                payload.numel(), C.c_void_p(off.data_ptr()), C.byref(total)) == 0
    out = torch.empty(n, dtype=torch.uint8, device=DEV)
    assert lib.hf_decode_chunks(h, C.c_void_p(payload.data_ptr()), payload.numel(),
                                C.c_void_p(off.data_ptr()), None, n, hf.HF_LAYOUT_U8,
                                C.c_void_p(cbs.data_ptr()), None, 0, C.c_void_p(out.data_ptr()),
                                1) == 0
    assert torch.equal(out, t)
    s = hf.EncodedStream(n=n, layout=hf.HF_LAYOUT_U8, dtype=torch.uint8, payload=payload,
                         chunk_off=off, side=None, codebook=None, total_words=int(total.value),
                         chunk_codebooks=cbs)
    code, lens, mode, _ = sys_.block_codebook(s, 7)
    f7 = orc.aw_symbol_stat(sym[7 * 16384:8 * 16384], 256, 64)
    ec, el = orc.rp_system_codebook(f7, 64, 0, 16)
    assert mode == 1 and np.array_equal(code, ec) and np.array_equal(lens, el)


def _static_call(hf, ctx, freq_t, nchunks, cbs, mode=None):
    return hf.L.lib().hf_system_codebook_chunks_static(
        ctx.handle, C.c_void_p(freq_t.data_ptr()), nchunks,
        hf.HF_MODE_HUFFMAN if mode is None else mode, 0, C.c_void_p(cbs.data_ptr()))


def test_static_entry_point_allocates_nothing(hf):
    # the property the entry point exists for: (a) free device memory does not move across the
    # first call of a fresh context, (b) that first call goes through inside a stream capture,
    # where any hipMalloc, hipFree or synchronisation fails the call and the capture
    L = hf.L
    rng = np.random.default_rng(11)
    pages = 700
    freq = torch.from_numpy(_counts(rng, pages, "mixed")).to(DEV).contiguous()
    exp = _records(hf, L.lib().hf_system_codebook_chunks_dev, hf.Context(0), freq, pages,
                   hf.HF_MODE_HUFFMAN, 0)
    cbs = torch.full_like(exp, 0xA5)       # (record bytes no kernel writes keep the fill)
    ctx = hf.Context(0)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    assert _static_call(hf, ctx, freq, pages, cbs) == 0
    ctx.sync()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0
    assert torch.equal(cbs, exp)
    ctx.close()

    side = torch.cuda.Stream()
    fresh = hf.Context(0, stream=side)   # no call of any kind before the capture
    cbs2 = torch.full_like(exp, 0xA5)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        rc = _static_call(hf, fresh, freq, pages, cbs2)
    assert rc == 0
    assert not torch.equal(cbs2, exp)    # captured, not run
    for seed in (1, 2):
        nxt = torch.from_numpy(_counts(np.random.default_rng(seed), pages, "mixed")).to(DEV)
        freq.copy_(nxt)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        want = _records(hf, L.lib().hf_system_codebook_chunks_dev, hf.Context(0), freq, pages,
                        hf.HF_MODE_HUFFMAN, 0)
        assert torch.equal(cbs2, want)
    fresh.close()


def test_static_scratch_is_private_to_a_context(hf):
    # two contexts on two streams of one GPU, full-size calls enqueued back to back without any
    # synchronisation between them: each works in its own slice of the static array, so both get
    # the records of the allocating entry point
    L = hf.L
    lib = L.lib()
    rng = np.random.default_rng(99)
    pages = L.HF_STATIC_CB_MAX_CHUNKS
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ctxs = [hf.Context(0, stream=s) for s in streams]
    freqs = [torch.from_numpy(_counts(rng, 64, "mixed")).to(DEV).repeat(pages // 64, 1).contiguous()
             for _ in range(2)]
    exps = [_records(hf, lib.hf_system_codebook_chunks_dev, hf.Context(0), f, pages,
                     hf.HF_MODE_BEST, 0) for f in freqs]
    outs = [[torch.full_like(exps[0], 0xA5) for _ in range(6)] for _ in range(2)]
    torch.cuda.synchronize()
    for it in range(6):
        for k in (0, 1):
            assert _static_call(hf, ctxs[k], freqs[k], pages, outs[k][it], hf.HF_MODE_BEST) == 0
    torch.cuda.synchronize()
    for k in (0, 1):
        ctxs[k].sync()
        for it in range(6):
            assert torch.equal(outs[k][it], exps[k]), (k, it)
    # the slices are a fixed pool: contexts past HF_STATIC_CB_SLOTS on one GPU are refused, with
    # nothing launched, until a holder is destroyed
    one = freqs[0][:256].contiguous()
    cbs = torch.full((L.HF_CHUNK_CODEBOOK_BYTES,), 0xA5, dtype=torch.uint8, device=DEV)
    extra, refused = [], None
    for _ in range(L.HF_STATIC_CB_SLOTS + 1):
        c = hf.Context(0)
        rc = _static_call(hf, c, one, 1, cbs)
        if rc == L.HF_ERR_NOMEM:
            refused = c
            break
        assert rc == 0
        extra.append(c)
    assert refused is not None and len(extra) <= L.HF_STATIC_CB_SLOTS - 2
    torch.cuda.synchronize()
    ctxs[0].close()
    assert _static_call(hf, refused, one, 1, cbs) == 0
    refused.sync()
    assert torch.equal(cbs, exps[0][: L.HF_CHUNK_CODEBOOK_BYTES])
    for c in extra + [refused, ctxs[1]]:
        c.close()
