#!/usr/bin/env python3
"""Paged stream against the dense and slotted per-block streams (developer probe).

fp8 KV-cache pages with one codebook each: per-call device time (HIP events inside the library
around the call's launches) of the three encoders and decoders over all pages, and decode-on-read
latency against batch size for the dense and the paged stream.
usage: python tools/kprobe_pages.py [--pages N] [--kind fp8|geom|uniform|bf16]
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


def make(kind, pages, dev):
    g = torch.Generator(device=dev).manual_seed(7)
    n = pages * CH
    if kind == "fp8":
        scale = torch.exp(torch.randn(1024, generator=g, device=dev) * 0.7)
        t = torch.empty(n, dtype=torch.uint8, device=dev)
        for lo in range(0, pages, 4096):
            hi = min(pages, lo + 4096)
            x = torch.randn((hi - lo) * 16, 1024, generator=g, device=dev) * scale
            t[lo * CH:hi * CH] = x.to(torch.float8_e4m3fn).view(torch.uint8).reshape(-1)
        return t
    if kind == "uniform":
        return torch.randint(0, 256, (n,), generator=g, device=dev, dtype=torch.uint8)
    if kind == "geom":
        u = torch.rand(n, generator=g, device=dev)
        return torch.clamp(torch.floor(torch.log(u) / torch.log(torch.tensor(0.87, device=dev))), 0, 255).to(torch.uint8)
    if kind == "bf16":
        return (torch.randn(n, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    raise ValueError(kind)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=1 << 16)
    ap.add_argument("--kind", default="fp8")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-read", action="store_true", help="skip the decode-on-read batches")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    t = make(a.kind, a.pages, dev)
    n, pages = t.numel(), a.pages
    lay = hf.layout_of(t)
    eb = t.element_size()
    ctx = hf.Context(0)
    ctx.set_timing(True)
    lib, h = L.lib(), ctx.handle
    P = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
    freq = torch.empty(pages * 256, dtype=torch.int32, device=dev)
    cbs = torch.empty(pages * L.HF_CHUNK_CODEBOOK_BYTES, dtype=torch.uint8, device=dev)
    side = torch.empty(n, dtype=torch.uint8, device=dev) if lay == 1 else None
    cap = pages * (CH // 4 + 256) + 64
    pay = {k: torch.empty(cap, dtype=torch.int32, device=dev)
           for k in ("dense", "slotted", "paged", "shared", "dense1", "once")}
    off = {k: torch.empty(pages + 4, dtype=torch.int32, device=dev) for k in pay}
    words_s = torch.empty(pages + 4, dtype=torch.int32, device=dev)
    out = torch.empty(n, dtype=t.dtype, device=dev)
    total = C.c_uint64(0)

    def timed(fn, reps=a.reps):
        best = 1e9
        for _ in range(reps):
            L.check(fn(), "call")
            best = min(best, ctx.last_kernel_ms())
        return best

    ms = {}
    ms["hist"] = timed(lambda: lib.hf_symbol_stat_chunks_dev(h, P(t), n, lay, P(freq)))
    ms["codebooks"] = timed(lambda: lib.hf_system_codebook_chunks_dev(h, P(freq), pages, 1, 0, P(cbs)))
    ms["enc dense"] = timed(lambda: lib.hf_encode_chunks(h, P(t), n, lay, P(cbs), P(side), P(pay["dense"]), cap, P(off["dense"]), None))
    ms["enc slotted"] = timed(lambda: lib.hf_encode_chunks_slotted(h, P(t), n, lay, P(cbs), P(freq), P(side), P(pay["slotted"]), cap, P(off["slotted"]), P(words_s), None))
    ms["enc paged"] = timed(lambda: lib.hf_encode_pages(h, P(t), n, lay, P(cbs), P(freq), P(side), P(pay["paged"]), cap, P(off["paged"]), None))
    # one codebook for the stream: the packed stream and the pages without records
    freq1 = torch.zeros(256, dtype=torch.int64, device=dev)
    cb1 = torch.zeros(L.HF_CODEBOOK_DEV_BYTES, dtype=torch.uint8, device=dev)
    ms["hist (one)"] = timed(lambda: (freq1.zero_(), lib.hf_symbol_stat_dev(h, P(t), n, lay, P(freq1)))[1])
    fq2 = torch.empty_like(freq)
    ms["hist (pages+one)"] = timed(lambda: (freq1.zero_(), lib.hf_symbol_stat_pages_dev(h, P(t), n, lay, P(fq2), P(freq1)))[1])
    ok_h = torch.equal(fq2, freq)
    ms["fold"] = timed(lambda: (freq1.zero_(), lib.hf_symbol_stat_fold_dev(h, P(freq), pages, P(freq1)))[1])
    ok_h = ok_h and int((freq1 - freq.view(pages, 256).sum(0)).abs().sum()) == 0
    print("per-page counts and stream totals of the one-pass kernel:", "ok" if ok_h else "MISMATCH")
    ms["codebook (one)"] = timed(lambda: lib.hf_system_codebook_dev(h, P(freq1), 1, 0, P(cb1)))
    ms["enc dense1"] = timed(lambda: lib.hf_encode(h, P(t), n, lay, P(cb1), P(side), P(pay["dense1"]), cap, P(off["dense1"]), None))
    ms["enc shared"] = timed(lambda: lib.hf_encode_pages_shared(h, P(t), n, lay, P(cb1), P(freq), P(side), P(pay["shared"]), cap, P(off["shared"]), None))
    # the same pages placed while they are encoded (no per-page counts)
    ms["enc once"] = timed(lambda: lib.hf_encode_pages_shared(h, P(t), n, lay, P(cb1), None, P(side), P(pay["once"]), cap, P(off["once"]), None))
    ctx.sync()
    sb = n if lay == 1 else 0
    words = {k: int(off[k][pages].item()) & 0xFFFFFFFF for k in pay}
    same = words["once"] == words["shared"] and torch.equal(off["once"][:pages + 1], off["shared"][:pages + 1]) \
        and torch.equal(pay["once"][:words["once"]], pay["shared"][:words["shared"]])
    print("pages placed while encoded equal pages placed from counts:", "ok" if same else "MISMATCH")
    stored = {"dense1": 4 * words["dense1"] + 4 * (pages + 1) + 768,
              "shared": 4 * words["shared"] + 4 * (pages + 1) + 768,
              "once": 4 * words["once"] + 4 * (pages + 1) + 768,
              "dense": 4 * words["dense"] + 4 * (pages + 1) + 296 * pages,
              "slotted": 4 * words["slotted"] + 8 * pages + 4 + 296 * pages,
              "paged": 4 * words["paged"] + 4 * (pages + 1)}
    out.zero_()
    ms["dec dense"] = timed(lambda: lib.hf_decode_chunks(h, P(pay["dense"]), cap, P(off["dense"]), P(side), n, lay, P(cbs), None, 0, P(out), 0))
    ok = torch.equal(out.view(torch.uint8), t.view(torch.uint8))
    out.zero_()
    ms["dec slotted"] = timed(lambda: lib.hf_decode_chunks_slotted(h, P(pay["slotted"]), cap, P(off["slotted"]), P(words_s), P(side), n, lay, P(cbs), None, None, 0, P(out), 0, 0))
    ok = ok and torch.equal(out.view(torch.uint8), t.view(torch.uint8))
    out.zero_()
    ms["dec paged"] = timed(lambda: lib.hf_decode_pages(h, P(pay["paged"]), cap, P(off["paged"]), P(side), n, lay, None, None, 0, P(out), 0, 0))
    ok = ok and torch.equal(out.view(torch.uint8), t.view(torch.uint8))
    out.zero_()
    ms["dec dense1"] = timed(lambda: lib.hf_decode(h, P(pay["dense1"]), cap, P(off["dense1"]), P(side), n, lay, P(cb1), P(out), 0))
    ok = ok and torch.equal(out.view(torch.uint8), t.view(torch.uint8))
    out.zero_()
    ms["dec shared"] = timed(lambda: lib.hf_decode_pages_shared(h, P(pay["shared"]), cap, P(off["shared"]), P(side), n, lay, P(cb1), None, None, 0, P(out), 0, 0))
    ok = ok and torch.equal(out.view(torch.uint8), t.view(torch.uint8))
    ctx.sync()
    print(f"{a.kind}: {pages} pages, {n * eb / 1e9:.2f} GB; round trips {'ok' if ok else 'MISMATCH'}")
    for k in ("dense", "slotted", "paged", "dense1", "shared"):
        print(f"  {k:8s} payload {32.0 * words[k] / n:.4f} bits/symbol, stored ratio "
              f"{n * eb / (stored[k] + sb):.4f}")
    for k, v in ms.items():
        kind = k.split()[-1]
        alg = n * eb
        if k.startswith("enc") or k.startswith("dec"):
            alg = n * eb + sb + 4.0 * words[kind]   # elements, side bytes, payload words
        print(f"  {k:12s} {v:8.3f} ms  {n * eb / v / 1e6:8.0f} GB/s of input  "
              f"{alg / v / 1e6 / 8000:6.3f} of peak (algorithmic bytes)")
    if lay != 0 or a.no_read:
        return
    pool_pages = min(1 << 15, pages)
    pool = torch.empty(pool_pages * CH, dtype=torch.uint8, device=dev)
    print(f"  decode-on-read into a pool of {pool_pages} pages: kernel us (dense | paged)")
    for batch in (1, 16, 64, 256, 1024, 4096, 16384):
        if batch > pool_pages:
            break
        ids = torch.randperm(pages, device=dev)[:batch].to(torch.int32)
        slots = torch.randperm(pool_pages, device=dev)[:batch].to(torch.int32)
        d = timed(lambda: lib.hf_decode_chunks_to(h, P(pay["dense"]), cap, P(off["dense"]), None, n, lay, P(cbs), P(ids), P(slots), batch, P(pool), pool_pages, 0))
        okd = torch.equal(pool.view(pool_pages, CH)[slots.long()], t.view(pages, CH)[ids.long()])
        pool.zero_()
        p = timed(lambda: lib.hf_decode_pages(h, P(pay["paged"]), cap, P(off["paged"]), None, n, lay, P(ids), P(slots), batch, P(pool), pool_pages, 0))
        okp = torch.equal(pool.view(pool_pages, CH)[slots.long()], t.view(pages, CH)[ids.long()])
        bits = 32.0 * words["paged"] / n
        frac = batch * CH * (1 + bits / 8) / (p * 1e-3) / 8e12
        print(f"  {batch:7d} {d * 1e3:9.1f} | {p * 1e3:9.1f}   paged {batch * CH / p / 1e6:7.0f} GB/s out, "
              f"{frac:.3f} of peak" + ("" if okd and okp else "  MISMATCH"))
    ctx.sync()


if __name__ == "__main__":
    main()
