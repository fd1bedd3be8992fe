#!/usr/bin/env python3
"""Does the place of the buffers in device memory change the page kernels' times?  (developer probe)

Config 2's step (per-page counts, codebook, pages without records, decode) on one tensor, the
side, payload and output buffers taken at a given byte offset inside larger allocations; prints
the per-step times of every phase.
usage: python tools/kprobe_place.py [--elems N] [--steps K] [--offsets a,b,c ...]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from huffman_amd import api as hf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--elems", type=int, default=8030261248)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--offsets", default="0", help="comma list of byte offsets (multiples of 16) tried in turn")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = a.elems
    g = torch.Generator(device=dev).manual_seed(1234)
    w = torch.empty(n, dtype=torch.bfloat16, device=dev)
    slab = 1 << 28
    for lo in range(0, n, slab):
        hi = min(n, lo + slab)
        w[lo:hi] = (torch.randn(hi - lo, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    ctx = hf.Context(0)
    system = hf.EntropyAwareHuffmanSystem(ctx=ctx)
    decoder = hf.ParrallelHuffmanDecoder(ctx=ctx)
    nch = (n + hf.HF_CHUNK_SYMS - 1) // hf.HF_CHUNK_SYMS
    ctx.reserve(n)
    slack = 1 << 22
    freq = torch.zeros(256, dtype=torch.int64, device=dev)
    freq_pages = torch.empty(nch * 256, dtype=torch.int32, device=dev)
    cb = hf.DeviceCodebook(ctx)
    side_all = torch.empty(n + slack, dtype=torch.uint8, device=dev)
    pay_words = n // 8 + nch * 192 + 64
    pay_all = torch.empty(pay_words * 4 + slack, dtype=torch.uint8, device=dev)
    out_all = torch.empty(2 * n + slack, dtype=torch.uint8, device=dev)
    chunk_off = torch.empty(((nch + 4) // 4) * 4, dtype=torch.int32, device=dev)
    print(f"w {w.data_ptr():#x} side {side_all.data_ptr():#x} payload {pay_all.data_ptr():#x} "
          f"out {out_all.data_ptr():#x} counts {freq_pages.data_ptr():#x}")
    for off in [int(x) for x in a.offsets.split(",")]:
        side = side_all[off:off + n]
        payload = pay_all[off:off + 4 * pay_words].view(torch.int32)
        out = out_all[off:off + 2 * n].view(torch.bfloat16)
        rows = []
        for k in range(a.steps + 2):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            ev[0].record()
            freq.zero_()
            system.histogram_pages(w, freq_chunks=freq_pages, freq256=freq)
            ev[1].record()
            system.codebook(freq, hf.HF_MODE_HUFFMAN, out=cb)
            ev[2].record()
            s = system.encode_with(w, cb, payload=payload, side=side, chunk_off=chunk_off, sync=False,
                                   paged=True, chunk_freq=freq_pages)
            ev[3].record()
            decoder.decode(s, out=out, check_stream=False)
            ev[4].record()
            torch.cuda.synchronize()
            if k >= 2:
                rows.append([ev[i].elapsed_time(ev[i + 1]) for i in range(4)])
        ctx.sync()
        ok = torch.equal(out.view(torch.int16), w.view(torch.int16))
        for i, name in enumerate(("hist", "codebook", "encode", "decode")):
            if name == "codebook":
                continue
            print(f"offset {off:8d} {name:7s} " + " ".join(f"{r[i]:6.3f}" for r in rows))
        print(f"offset {off:8d} round trip {'ok' if ok else 'MISMATCH'}")


if __name__ == "__main__":
    main()
