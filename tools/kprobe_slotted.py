#!/usr/bin/env python3
"""Dense against slotted placement of a per-block stream, per kernel call (developer probe).

usage: python tools/kprobe_slotted.py [--mib N] [--reps R] [--kinds gauss,fp8,uniform,geom]
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


def make(kind, n, dev):
    g = torch.Generator(device=dev).manual_seed(11)
    if kind == "gauss":  # bf16 N(0, 0.02) weights
        return (torch.randn(n, generator=g, device=dev) * 0.02).to(torch.bfloat16), 1
    if kind == "fp8":
        x = torch.randn(n, generator=g, device=dev)
        return x.to(torch.float8_e4m3fn).view(torch.uint8), 0
    if kind == "uniform":
        return torch.randint(0, 256, (n,), generator=g, device=dev, dtype=torch.uint8), 0
    if kind == "geom":
        u = torch.rand(n, generator=g, device=dev)
        return torch.clamp(torch.floor(torch.log(u) / -0.0834), max=255).to(torch.uint8), 0
    raise ValueError(kind)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024, help="symbols, in Mi")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--kinds", default="gauss,fp8,uniform,geom")
    ap.add_argument("--no-check", action="store_true", help="ignore decode errors (experimental builds)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = hf.Context(0)
    ctx.set_timing(True)
    lib, h = L.lib(), ctx.handle
    P = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
    n = a.mib << 20
    nch = (n + CH - 1) // CH
    for kind in a.kinds.split(","):
        t, lay = make(kind, n, dev)
        freq = torch.empty(nch * 256, dtype=torch.int32, device=dev)
        cbs = torch.empty(nch * L.HF_CHUNK_CODEBOOK_BYTES, dtype=torch.uint8, device=dev)
        payload = torch.empty(n // 4 + nch * 192 + 64, dtype=torch.int32, device=dev)
        payload_s = torch.empty(nch * L.HF_SLOT_WORDS(CH * 8), dtype=torch.int32, device=dev)
        off = torch.empty(nch + 4, dtype=torch.int32, device=dev)
        off_s = torch.empty(nch + 4, dtype=torch.int32, device=dev)
        words_s = torch.empty(nch + 4, dtype=torch.int32, device=dev)
        side = torch.empty(n, dtype=torch.uint8, device=dev) if lay == 1 else None
        out = torch.empty_like(t)
        out_s = torch.empty_like(t)
        tm = {k: [] for k in ("hist", "codebook", "encode", "decode", "encode_slot", "decode_slot",
                              "1cb_hist", "1cb_fold", "1cb_encode", "1cb_decode", "1cb_encode_slot",
                              "1cb_decode_slot")}
        total, total_s, total_1, total_1s = (C.c_uint64(0) for _ in range(4))
        freq1 = torch.zeros(256, dtype=torch.int64, device=dev)
        cb1 = hf.DeviceCodebook(ctx)
        out_1 = torch.empty_like(t)
        out_1s = torch.empty_like(t)
        for _ in range(a.reps):
            L.check(lib.hf_symbol_stat_chunks_dev(h, P(t), n, lay, P(freq)), "stat")
            tm["hist"].append(ctx.last_kernel_ms())
            L.check(lib.hf_system_codebook_chunks_dev(h, P(freq), nch, 1, 0, P(cbs)), "codebook")
            tm["codebook"].append(ctx.last_kernel_ms())
            L.check(lib.hf_encode_chunks(h, P(t), n, lay, P(cbs), P(side), P(payload),
                                         payload.numel(), P(off), C.byref(total)), "encode")
            tm["encode"].append(ctx.last_kernel_ms())
            L.check(lib.hf_decode_chunks(h, P(payload), payload.numel(), P(off), P(side), n, lay,
                                         P(cbs), None, 0, P(out), 1), "decode")
            tm["decode"].append(ctx.last_kernel_ms())
            L.check(lib.hf_encode_chunks_slotted(h, P(t), n, lay, P(cbs), P(freq), P(side),
                                                 P(payload_s), payload_s.numel(), P(off_s),
                                                 P(words_s), C.byref(total_s)), "encode_slotted")
            tm["encode_slot"].append(ctx.last_kernel_ms())
            rc = lib.hf_decode_chunks_slotted(h, P(payload_s), payload_s.numel(), P(off_s),
                                              P(words_s), P(side), n, lay, P(cbs), None, None,
                                              0, P(out_s), 0, 1)
            if not a.no_check:
                L.check(rc, "decode_slotted")
            tm["decode_slot"].append(ctx.last_kernel_ms())
            # one codebook for the stream: dense (counted in one piece) against slotted (counted
            # per chunk and folded)
            freq1.zero_()
            L.check(lib.hf_symbol_stat_dev(h, P(t), n, lay, P(freq1)), "stat1")
            tm["1cb_hist"].append(ctx.last_kernel_ms())
            freq1.zero_()
            L.check(lib.hf_symbol_stat_fold_dev(h, P(freq), nch, P(freq1)), "fold")
            tm["1cb_fold"].append(ctx.last_kernel_ms())
            L.check(lib.hf_system_codebook_dev(h, P(freq1), 1, 0, C.c_void_p(cb1.ptr)), "codebook1")
            L.check(lib.hf_encode(h, P(t), n, lay, C.c_void_p(cb1.ptr), P(side), P(payload),
                                  payload.numel(), P(off), C.byref(total_1)), "encode1")
            tm["1cb_encode"].append(ctx.last_kernel_ms())
            L.check(lib.hf_decode(h, P(payload), payload.numel(), P(off), P(side), n, lay,
                                  C.c_void_p(cb1.ptr), P(out_1), 1), "decode1")
            tm["1cb_decode"].append(ctx.last_kernel_ms())
            L.check(lib.hf_encode_slotted(h, P(t), n, lay, C.c_void_p(cb1.ptr), P(freq), P(side),
                                          P(payload_s), payload_s.numel(), P(off_s), P(words_s),
                                          C.byref(total_1s)), "encode1_slotted")
            tm["1cb_encode_slot"].append(ctx.last_kernel_ms())
            rc = lib.hf_decode_slotted(h, P(payload_s), payload_s.numel(), P(off_s), P(words_s),
                                       P(side), n, lay, C.c_void_p(cb1.ptr), P(out_1s), 1)
            if not a.no_check:
                L.check(rc, "decode1_slotted")
            tm["1cb_decode_slot"].append(ctx.last_kernel_ms())
        ok = torch.equal(out.view(torch.uint8), t.view(torch.uint8)) and \
            torch.equal(out_s.view(torch.uint8), t.view(torch.uint8)) and \
            torch.equal(out_1.view(torch.uint8), t.view(torch.uint8)) and \
            torch.equal(out_1s.view(torch.uint8), t.view(torch.uint8))
        eb = 2 if lay == 1 else 1
        print(f"{kind}: {n * eb / 1e9:.2f} GB  dense words {total.value}  slotted words "
              f"{total_s.value} (+{100.0 * (total_s.value / total.value - 1):.2f} %)  one codebook: "
              f"{total_1.value} / {total_1s.value} "
              f"(+{100.0 * (total_1s.value / total_1.value - 1):.2f} %)  ok={ok}")
        for k, v in tm.items():
            print(f"  {k:12s} {min(v):8.3f} ms  {n * eb / 1e9 / (min(v) * 1e-3):8.0f} GB/s in")
        del t, freq, cbs, payload, payload_s, out, out_s, side, out_1, out_1s
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
