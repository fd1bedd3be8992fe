#!/usr/bin/env python3
"""Time the per-block path on synthetic fp8 KV-cache pages (developer probe).

16-token x 1024-byte pages, N(0,1) x per-channel scale -> fp8 E4M3; one codebook per page.
usage: python tools/kprobe_blocks.py [--pages N] [--reps R]
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
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--read-batch", type=int, default=4096)
    ap.add_argument("--no-check", action="store_true", help="ignore decode errors (experimental builds)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(7)
    scale = torch.exp(torch.randn(1024, generator=g, device=dev) * 0.7)
    t = torch.empty(a.pages * CH, dtype=torch.uint8, device=dev)
    step = 4096
    for lo in range(0, a.pages, step):
        hi = min(a.pages, lo + step)
        x = torch.randn((hi - lo) * 16, 1024, generator=g, device=dev) * scale
        t[lo * CH:hi * CH] = x.to(torch.float8_e4m3fn).view(torch.uint8).reshape(-1)
    ctx = hf.Context(0)
    ctx.set_timing(True)
    lib, h = L.lib(), ctx.handle
    n, nch = t.numel(), a.pages
    freq = torch.empty(nch * 256, dtype=torch.int32, device=dev)
    cbs = torch.empty(nch * L.HF_CHUNK_CODEBOOK_BYTES, dtype=torch.uint8, device=dev)
    payload = torch.empty(n // 4 + nch * 192 + 64, dtype=torch.int32, device=dev)
    off = torch.empty(nch + 4, dtype=torch.int32, device=dev)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    ids = torch.randperm(nch, device=dev)[:a.read_batch].to(torch.int32)
    out2 = torch.empty(ids.numel() * CH, dtype=torch.uint8, device=dev)
    P = lambda x: C.c_void_p(x.data_ptr())
    tm = {k: [] for k in ("hist", "codebook", "encode", "decode", "read", "encode_slot",
                          "decode_slot", "read_slot")}
    total, total_s = C.c_uint64(0), C.c_uint64(0)
    payload_s = torch.empty(nch * L.HF_SLOT_WORDS(CH * 8), dtype=torch.int32, device=dev)
    off_s = torch.empty(nch + 4, dtype=torch.int32, device=dev)
    words_s = torch.empty(nch + 4, dtype=torch.int32, device=dev)
    out_s = torch.empty(n, dtype=torch.uint8, device=dev)
    out2_s = torch.empty(ids.numel() * CH, dtype=torch.uint8, device=dev)
    for _ in range(a.reps):
        L.check(lib.hf_symbol_stat_chunks_dev(h, P(t), n, 0, P(freq)), "stat")
        tm["hist"].append(ctx.last_kernel_ms())
        L.check(lib.hf_system_codebook_chunks_dev(h, P(freq), nch, 1, 0, P(cbs)), "codebook")
        tm["codebook"].append(ctx.last_kernel_ms())
        L.check(lib.hf_encode_chunks(h, P(t), n, 0, P(cbs), None, P(payload), payload.numel(),
                                     P(off), C.byref(total)), "encode")
        tm["encode"].append(ctx.last_kernel_ms())
        rc = lib.hf_decode_chunks(h, P(payload), payload.numel(), P(off), None, n, 0, P(cbs),
                                  None, 0, P(out), 1)
        if not a.no_check:
            L.check(rc, "decode")
        tm["decode"].append(ctx.last_kernel_ms())
        rc = lib.hf_decode_chunks(h, P(payload), payload.numel(), P(off), None, n, 0, P(cbs),
                                  P(ids), ids.numel(), P(out2), 1)
        if not a.no_check:
            L.check(rc, "read")
        tm["read"].append(ctx.last_kernel_ms())
        # the same pages in slots sized from their histograms
        L.check(lib.hf_encode_chunks_slotted(h, P(t), n, 0, P(cbs), P(freq), None, P(payload_s),
                                             payload_s.numel(), P(off_s), P(words_s),
                                             C.byref(total_s)), "encode_slotted")
        tm["encode_slot"].append(ctx.last_kernel_ms())
        L.check(lib.hf_decode_chunks_slotted(h, P(payload_s), payload_s.numel(), P(off_s),
                                             P(words_s), None, n, 0, P(cbs), None, None, 0,
                                             P(out_s), 0, 1), "decode_slotted")
        tm["decode_slot"].append(ctx.last_kernel_ms())
        L.check(lib.hf_decode_chunks_slotted(h, P(payload_s), payload_s.numel(), P(off_s),
                                             P(words_s), None, n, 0, P(cbs), P(ids), None,
                                             ids.numel(), P(out2_s), 0, 1), "read_slotted")
        tm["read_slot"].append(ctx.last_kernel_ms())
    ok = torch.equal(out, t) and torch.equal(out2, t.view(nch, CH)[ids.long()].reshape(-1))
    ok_s = torch.equal(out_s, t) and torch.equal(out2_s, out2)
    gb = n / 1e9
    bits = 32.0 * total.value / n
    comp = 4 * total.value + 4 * (nch + 1) + 273 * nch
    print(f"pages={nch} ({gb:.2f} GB) bits/sym={bits:.3f} ratio={n / comp:.3f} ok={ok}")
    comp_s = 4 * total_s.value + 4 * (2 * nch + 1) + 273 * nch
    print(f"slotted: words {total_s.value} (+{100.0 * (total_s.value / total.value - 1):.2f} %) "
          f"ratio={n / comp_s:.3f} ok={ok_s}")
    for k, v in tm.items():
        rd = k.startswith("read")
        b = gb if not rd else ids.numel() * CH / 1e9
        per = min(v) * 1e3 / (nch if not rd else ids.numel())
        print(f"  {k:9s} {min(v):8.3f} ms  {b / (min(v) * 1e-3):8.0f} GB/s  {per:7.3f} us/page-slot")


if __name__ == "__main__":
    main()
