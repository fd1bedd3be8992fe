#!/usr/bin/env python3
"""The page encoder on payload buffers allocated afresh each round (developer probe, meant to run
under rocprofv3 --pmc: three encode calls per round, the per-dispatch counters tell the rounds apart)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from huffman_amd import api as hf

dev = torch.device("cuda", 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8030261248
g = torch.Generator(device=dev).manual_seed(1234)
w = torch.empty(n, dtype=torch.bfloat16, device=dev)
slab = 1 << 28
for lo in range(0, n, slab):
    hi = min(n, lo + slab)
    w[lo:hi] = (torch.randn(hi - lo, generator=g, device=dev) * 0.02).to(torch.bfloat16)
ctx = hf.Context(0)
system = hf.EntropyAwareHuffmanSystem(ctx=ctx)
nch = (n + hf.HF_CHUNK_SYMS - 1) // hf.HF_CHUNK_SYMS
ctx.reserve(n)
freq = torch.zeros(256, dtype=torch.int64, device=dev)
freq_pages = torch.empty(nch * 256, dtype=torch.int32, device=dev)
cb = hf.DeviceCodebook(ctx)
chunk_off = torch.empty(((nch + 4) // 4) * 4, dtype=torch.int32, device=dev)
pay_words = n // 8 + nch * 192 + 64
system.histogram_pages(w, freq_chunks=freq_pages, freq256=freq)
system.codebook(freq, hf.HF_MODE_HUFFMAN, out=cb)
side = torch.empty(n, dtype=torch.uint8, device=dev)
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
payload = None
for trial in range(12):
    del payload
    torch.cuda.empty_cache()
    pad = torch.empty(random.randrange(1, 3000) << 20, dtype=torch.uint8, device=dev)
    payload = torch.empty(pay_words, dtype=torch.int32, device=dev)
    del pad
    torch.cuda.empty_cache()
    ts = []
    for k in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        system.encode_with(w, cb, payload=payload, side=side, chunk_off=chunk_off, sync=False,
                           paged=True, chunk_freq=freq_pages)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"round {trial:2d} payload {payload.data_ptr():#x} encode {min(ts):.3f} ms", flush=True)
ctx.sync()
