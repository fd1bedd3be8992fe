#!/usr/bin/env python3
"""Time the container path (developer probe): pack an encoded stream into one blob, open it,
decode it in place.  usage: python tools/kprobe_container.py [--elems N]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from huffman_amd import api as hf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--elems", type=int, default=1 << 30)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    w = torch.empty(a.elems, dtype=torch.bfloat16, device=dev)
    for lo in range(0, a.elems, 1 << 28):
        hi = min(a.elems, lo + (1 << 28))
        w[lo:hi] = torch.randn(hi - lo, generator=g, device=dev).mul_(0.02).to(torch.bfloat16)
    ctx = hf.default_context()
    ctx.set_timing(True)
    system, dec = hf.EntropyAwareHuffmanSystem(), hf.ParrallelHuffmanDecoder()
    s = system.encode(w, mode=hf.HF_MODE_HUFFMAN)
    best = {"pack": 1e9, "open": 1e9, "decode": 1e9}
    for _ in range(3):
        blob = s.to_container()
        best["pack"] = min(best["pack"], ctx.last_kernel_ms())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s2 = hf.EncodedStream.from_container(blob)
        torch.cuda.synchronize()
        best["open"] = min(best["open"], (time.perf_counter() - t0) * 1e3)
        out = dec.decode(s2)
        best["decode"] = min(best["decode"], ctx.last_kernel_ms())
    ok = torch.equal(out.view(torch.int16), w.view(torch.int16))
    gb = blob.numel() / 1e9
    print(f"elems={a.elems} blob={gb:.3f} GB ratio={2 * a.elems / blob.numel():.4f} ok={ok}")
    print(f"  pack   {best['pack']:8.3f} ms  {2 * gb / (best['pack'] * 1e-3):8.0f} GB/s read+write")
    print(f"  open   {best['open']:8.3f} ms  (header check + table rebuild, host-synchronous)")
    print(f"  decode {best['decode']:8.3f} ms  {2 * a.elems / 1e9 / (best['decode'] * 1e-3):8.0f} GB/s of output, in place")


if __name__ == "__main__":
    main()
