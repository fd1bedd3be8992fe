"""The enqueue-only calls under HIP graph capture: histogram -> codebook -> encode -> decode of a
block captured once on a side stream and replayed on new data.  (After a context's first use none
of these calls allocates or synchronises; hf_encode with h_total_words = NULL and hf_decode with
check = 0 only enqueue.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n", [4096, 5 * 16384 + 100])
def test_block_pipeline_replays_from_a_captured_graph(hf, orc, n):
    rng = np.random.default_rng(n)
    first = np.minimum(rng.geometric(0.08, n) - 1, 255).astype(np.uint8)
    t = torch.from_numpy(first).to(DEV)
    side = torch.cuda.Stream()
    ctx = hf.Context(0, stream=side)
    system, decoder, stat = (hf.EntropyAwareHuffmanSystem(ctx=ctx),
                             hf.ParrallelHuffmanDecoder(ctx=ctx), hf.SymbolStat(ctx=ctx))
    freq = torch.zeros(256, dtype=torch.int64, device=DEV)
    cb = hf.DeviceCodebook(ctx)
    nch = (n + 16383) // 16384
    payload = torch.empty(n // 4 + nch * 192 + 64, dtype=torch.int32, device=DEV)
    off = torch.empty(((nch + 1 + 3) // 4) * 4, dtype=torch.int32, device=DEV)
    out = torch.empty(n, dtype=torch.uint8, device=DEV)

    def block():
        freq.zero_()
        stat.accumulate(t, freq)
        system.codebook(freq, hf.HF_MODE_HUFFMAN, out=cb)
        s = system.encode_with(t, cb, payload=payload, chunk_off=off, sync=False)
        decoder.decode(s, out=out, check_stream=False)

    with torch.cuda.stream(side):
        block()  # first use: the context's lazy allocations happen here, outside the capture
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        block()
    for seed in (1, 2):
        # new contents in the captured input buffer, another distribution each time
        nxt = rng.integers(0, 256 >> seed, n).astype(np.uint8)
        t.copy_(torch.from_numpy(nxt).to(DEV))
        out.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        ctx.sync()  # (reports kernel errors kept by the enqueue-only calls, if any)
        assert np.array_equal(out.cpu().numpy(), nxt)
        code, lens = orc.rp_system_codebook(orc.aw_symbol_stat(nxt, 256, 64), 64, 0, 16)
        ref_payload, ref_off = orc.rp_encode(nxt, code, lens)
        total = int(off[nch].item())
        assert total == ref_payload.size
        assert np.array_equal(payload[:total].cpu().numpy().view(np.uint32), ref_payload)
        assert np.array_equal(off[: nch + 1].cpu().numpy().view(np.uint32), ref_off)


@pytest.mark.parametrize("slotted", [False, True])
def test_per_block_pipeline_replays_from_a_captured_graph(hf, orc, slotted):
    # per-page histograms -> per-page codebooks -> encode -> decode, captured once
    pages = 24
    n = pages * 16384
    rng = np.random.default_rng(3)
    t = torch.from_numpy(np.minimum(rng.geometric(0.1, n) - 1, 255).astype(np.uint8)).to(DEV)
    side = torch.cuda.Stream()
    ctx = hf.Context(0, stream=side)
    system, decoder = hf.EntropyAwareHuffmanSystem(ctx=ctx), hf.ParrallelHuffmanDecoder(ctx=ctx)
    out = torch.empty(n, dtype=torch.uint8, device=DEV)
    keep = {}

    def block():
        s = system.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, sync=False, slotted=slotted)
        decoder.decode_blocks(s, out=out, check_stream=False)
        keep["s"] = s

    with torch.cuda.stream(side):
        block()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        block()
    s = keep["s"]  # the buffers the captured calls write to
    for seed in (1, 2, 3):
        nxt = rng.integers(0, 256 >> seed, n).astype(np.uint8)
        t.copy_(torch.from_numpy(nxt).to(DEV))
        out.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        ctx.sync()
        assert np.array_equal(out.cpu().numpy(), nxt)
        if slotted:
            payload, off, words, _, _ = orc.rp_encode_blocks_slotted(nxt)
            assert np.array_equal(s.chunk_words[:pages].cpu().numpy().view(np.uint32), words)
        else:
            payload, off, _, _ = orc.rp_encode_blocks(nxt)
        assert np.array_equal(s.chunk_off[: pages + 1].cpu().numpy().view(np.uint32), off)
        total = int(s.chunk_off[pages].item())
        assert total == payload.size
        assert np.array_equal(s.payload[:total].cpu().numpy().view(np.uint32), payload)


def test_reserved_context_captures_calls_it_never_ran(hf):
    # a context whose only earlier call was a histogram of a tiny tensor: hf_ctx_reserve sizes every
    # workspace for 70 000 blocks (more than the descriptor array's first size), after which the
    # packed, the per-block (dense and slotted) and both paged pipelines are captured cold
    pages = 70_000
    n = pages * 16384
    side = torch.cuda.Stream()
    ctx = hf.Context(0, stream=side)
    system, decoder, stat = (hf.EntropyAwareHuffmanSystem(ctx=ctx),
                             hf.ParrallelHuffmanDecoder(ctx=ctx), hf.SymbolStat(ctx=ctx))
    gen = torch.Generator(device=DEV).manual_seed(1)
    t = (torch.randn(n, device=DEV, generator=gen) * 9).abs().to(torch.uint8)
    freq = torch.zeros(256, dtype=torch.int64, device=DEV)
    with torch.cuda.stream(side):
        stat.accumulate(t[:1000].contiguous(), freq)      # the warm-up: another call, a smaller n
    ctx.reserve(n)
    side.synchronize()
    outs = [torch.empty(n, dtype=torch.uint8, device=DEV) for _ in range(5)]
    cb = hf.DeviceCodebook(ctx)
    keep = []

    def block():
        freq.zero_()
        stat.accumulate(t, freq)
        system.codebook(freq, hf.HF_MODE_HUFFMAN, out=cb)
        s = system.encode_with(t, cb, sync=False)
        decoder.decode(s, out=outs[0], check_stream=False)
        keep.append(s)
        s = system.encode_with(t, cb, sync=False, paged=True)   # pages placed while encoded
        decoder.decode(s, out=outs[4], check_stream=False)
        keep.append(s)
        for k, kw in enumerate(({}, {"slotted": True}, {"paged": True})):
            s = system.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN, sync=False, **kw)
            decoder.decode_blocks(s, out=outs[1 + k], check_stream=False)
            keep.append(s)

    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        block()
    for seed in (2, 3):
        gen.manual_seed(seed)
        t.copy_((torch.randn(n, device=DEV, generator=gen) * (3 * seed)).abs().to(torch.uint8))
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        ctx.sync()
        for o in outs:
            assert torch.equal(o, t)
