#!/usr/bin/env python3
"""Time the three streaming kernels on a few synthetic streams (developer probe, not a test).

usage: python tools/kprobe.py [--elems N] [--reps R] [--kinds gauss,uniform,geom]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from huffman_amd import api as hf


def make(kind, n, dev):
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    if kind == "gauss":
        return torch.randn(n, generator=g, device=dev).mul_(0.02).to(torch.bfloat16)
    if kind == "uniform":
        return torch.randint(0, 256, (n,), generator=g, device=dev, dtype=torch.uint8)
    if kind == "geom":
        u = torch.rand(n, generator=g, device=dev)
        return torch.clamp(torch.floor(-torch.log2(u)), 0, 255).to(torch.uint8)
    if kind == "fp8":
        x = torch.randn(n, generator=g, device=dev)
        return x.to(torch.float8_e4m3fn).view(torch.uint8)
    raise ValueError(kind)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--elems", type=int, default=1 << 30)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--kinds", default="gauss,uniform,geom,fp8")
    ap.add_argument("--no-decode", action="store_true", help="skip decode (experimental encoder builds)")
    ap.add_argument("--lmax", type=int, default=0,
                    help="code length limit (tree_depth_limit); the codebook then comes from the "
                         "module-level builder and is uploaded through MemWrapper")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = hf.Context(0)
    ctx.set_timing(True)
    sysm = hf.EntropyAwareHuffmanSystem(ctx=ctx)
    dec = hf.ParrallelHuffmanDecoder(ctx=ctx)
    for kind in a.kinds.split(","):
        w = make(kind, a.elems, dev)
        eb = w.element_size()
        freq = torch.zeros(256, dtype=torch.int64, device=dev)
        t = {"hist": [], "enc": [], "dec": []}
        for _ in range(a.reps):
            freq.zero_()
            hf.SymbolStat(ctx=ctx).accumulate(w, freq)
            t["hist"].append(ctx.last_kernel_ms())
            if a.lmax:
                srt = hf.SymbolSort(ctx=ctx)
                by_rank = srt(freq.cpu().numpy())
                code, lens = hf.EntropyAwareTreeBuilder(ctx=ctx)(by_rank, 1, a.lmax)
                mw = hf.MemWrapper(ctx=ctx)
                for r in range(256):
                    mw.write(int(srt.rank2sym[r]), int(code[r]), int(lens[r]))
                cb = mw.to_device()
            else:
                cb = sysm.codebook(freq, hf.HF_MODE_HUFFMAN)
            s = sysm.encode_with(w, cb)
            t["enc"].append(ctx.last_kernel_ms())
            if a.no_decode:
                t["dec"].append(float("nan"))
                out = w
                continue
            out = dec.decode(s)
            t["dec"].append(ctx.last_kernel_ms())
        ok = torch.equal(out.view(torch.int16) if eb == 2 else out, w.view(torch.int16) if eb == 2 else w)
        gb = a.elems * eb / 1e9
        bits = 32.0 * s.total_words / a.elems
        print(f"{kind:8s} n={a.elems} bits/sym={bits:.3f} ratio={s.ratio():.3f} ok={ok} "
              + " ".join(f"{k}={min(v):.3f}ms({gb / (min(v) * 1e-3):.0f}GB/s)" for k, v in t.items()),
              flush=True)
        del w, s, out


if __name__ == "__main__":
    main()
