#!/usr/bin/env python3
"""Decode-on-read latency against batch size (developer probe).

fp8 KV-cache pages with one codebook each are encoded once; then batches of randomly chosen pages
are decoded into caller-named slots of a page pool (hf_decode_chunks_to), as a paged-attention
runtime would fetch them.  Prints, per batch size, the kernel time (HIP events around the launch,
no host synchronisation inside) and what that is per page and in output GB/s.
usage: python tools/kprobe_serve.py [--pages N]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from huffman_amd import _lib as L
from huffman_amd import api as hf

CH = 16384


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=1 << 16)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(7)
    scale = torch.exp(torch.randn(1024, generator=g, device=dev) * 0.7)
    t = torch.empty(a.pages * CH, dtype=torch.uint8, device=dev)
    for lo in range(0, a.pages, 4096):
        hi = min(a.pages, lo + 4096)
        x = torch.randn((hi - lo) * 16, 1024, generator=g, device=dev) * scale
        t[lo * CH:hi * CH] = x.to(torch.float8_e4m3fn).view(torch.uint8).reshape(-1)
    system = hf.EntropyAwareHuffmanSystem()
    s = system.encode_blocks(t, mode=hf.HF_MODE_HUFFMAN)
    ctx = system.ctx
    ctx.set_timing(True)
    lib, h = L.lib(), ctx.handle
    pool_pages = 1 << 15
    pool = torch.empty(pool_pages * CH, dtype=torch.uint8, device=dev)
    P = lambda x: C.c_void_p(x.data_ptr())
    print(f"{a.pages} pages stored, ratio {s.ratio():.3f}; pool of {pool_pages} pages")
    print(f"{'batch':>7s} {'kernel us':>10s} {'ns/page':>9s} {'out GB/s':>9s}")
    for batch in (1, 16, 64, 256, 1024, 4096, 16384):
        ids = torch.randperm(a.pages, device=dev)[:batch].to(torch.int32)
        slots = torch.randperm(pool_pages, device=dev)[:batch].to(torch.int32)
        best = 1e9
        for _ in range(5):
            L.check(lib.hf_decode_chunks_to(h, P(s.payload), s.payload.numel(), P(s.chunk_off), None,
                                            s.n, s.layout, P(s.chunk_codebooks), P(ids), P(slots),
                                            batch, P(pool), pool_pages, 0), "read")
            best = min(best, ctx.last_kernel_ms())
        ok = torch.equal(pool.view(pool_pages, CH)[slots.long()], t.view(a.pages, CH)[ids.long()])
        print(f"{batch:7d} {best * 1e3:10.1f} {best * 1e6 / batch:9.0f} {batch * CH / best / 1e6:9.0f}"
              + ("" if ok else "  MISMATCH"))


if __name__ == "__main__":
    main()
