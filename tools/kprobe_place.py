#!/usr/bin/env python3
"""Encoder time against where the output buffers land in physical memory (developer probe): one of
the side / payload buffers at a time is freed and allocated again behind a pad of another size
(same or another virtual address, other physical pages), everything else kept."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from huffman_amd import api as hf

dev = torch.device("cuda", 0)
n = 8030261248
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
bufs = {"side": torch.empty(n, dtype=torch.uint8, device=dev),
        "payload": torch.empty(pay_words, dtype=torch.int32, device=dev)}

def enc_ms():
    ts = []
    for k in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        system.encode_with(w, cb, payload=bufs["payload"], side=bufs["side"], chunk_off=chunk_off,
                           sync=False, paged=True, chunk_freq=freq_pages)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts[1:])[2]

decoder = hf.ParrallelHuffmanDecoder(ctx=ctx)
out = torch.empty(n, dtype=torch.bfloat16, device=dev)

def both_ms():
    te, td = [], []
    for k in range(5):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        s_ = system.encode_with(w, cb, payload=bufs["payload"], side=bufs["side"], chunk_off=chunk_off,
                                sync=False, paged=True, chunk_freq=freq_pages)
        e[1].record()
        decoder.decode(s_, out=out, check_stream=False)
        e[2].record()
        torch.cuda.synchronize()
        te.append(e[0].elapsed_time(e[1])); td.append(e[1].elapsed_time(e[2]))
    return sorted(te[1:])[2], sorted(td[1:])[2]

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
slack = 96 << 20
for trial in range(8):
    del bufs["payload"]
    torch.cuda.empty_cache()
    pad = torch.empty(random.randrange(1, 3000) << 20, dtype=torch.uint8, device=dev)
    big = torch.empty(pay_words * 4 + slack, dtype=torch.uint8, device=dev)
    del pad
    torch.cuda.empty_cache()
    line = f"payload alloc {big.data_ptr():#x}:"
    # (is the range itself slow to write, or to read?  a fill of it and a sum over it)
    tf = []
    for k in range(4):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); big.zero_(); e1.record(); big.view(torch.int32).sum(); e2.record()
        torch.cuda.synchronize()
        tf.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
    line += f"  fill {min(t[0] for t in tf[1:]):.3f} sum {min(t[1] for t in tf[1:]):.3f}"
    for off in (0, 4096, 65536, 1 << 20, (2 << 20) + 8192, 32 << 20, 80 << 20):
        bufs["payload"] = big[off:off + 4 * pay_words].view(torch.int32)
        e_, d_ = both_ms()
        line += f"  +{off >> 10}K {e_:.2f}/{d_:.2f}"
    print(line, flush=True)
    bufs["payload"] = big[:4 * pay_words].view(torch.int32)
    del big
# the side buffer allocated afresh, the payload buffer kept
for trial in range(4):
    del bufs["side"]
    torch.cuda.empty_cache()
    pad = torch.empty(random.randrange(1, 3000) << 20, dtype=torch.uint8, device=dev)
    bufs["side"] = torch.empty(n, dtype=torch.uint8, device=dev)
    del pad
    torch.cuda.empty_cache()
    e_, d_ = both_ms()
    print(f"side alloc {bufs['side'].data_ptr():#x}: {e_:.2f}/{d_:.2f}", flush=True)
ctx.sync()
