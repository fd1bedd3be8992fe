#!/usr/bin/env python3
"""Time the sub-field split / merge kernels alone (developer probe, not a test).

usage: python tools/kprobe_fields.py [--elems N] [--reps R]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from huffman_amd import _lib as L
from huffman_amd import api as hf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--elems", type=int, default=1 << 30)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = hf.Context(0)
    ctx.set_timing(True)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    n = a.elems
    for name, layout, make in (
        ("fp16 5-bit exponent", hf.HF_LAYOUT_FP16_EXP,
         lambda: torch.randn(n, generator=g, device=dev).mul_(0.02).to(torch.float16)),
        ("fp8 sign+exponent", hf.HF_LAYOUT_FP8_E4M3_EXP,
         lambda: torch.randn(n, generator=g, device=dev).to(torch.float8_e4m3fn).view(torch.uint8)),
        ("fp4 pairs", hf.HF_LAYOUT_FP4_E2M1_EXP,
         lambda: torch.randint(0, 256, (n,), generator=g, device=dev, dtype=torch.uint8)),
        ("bit planes", hf.HF_LAYOUT_BITPLANE8,
         lambda: torch.randint(0, 256, (n,), generator=g, device=dev, dtype=torch.uint8)),
    ):
        w = make()
        eb = w.element_size()
        nside = hf.side_bytes(n, layout)
        sym = torch.empty(n, dtype=torch.uint8, device=dev)
        side = torch.empty(max(16, nside), dtype=torch.uint8, device=dev)
        side_ptr = side.data_ptr() if nside else None
        out = torch.empty_like(w)
        ts, tm = [], []
        for _ in range(a.reps):
            hf.check(L.lib().hf_fields_split(ctx.handle, C.c_void_p(w.data_ptr()), n, layout,
                                             C.c_void_p(sym.data_ptr()), C.c_void_p(side_ptr)),
                     "hf_fields_split")
            ctx.sync()
            ts.append(ctx.last_kernel_ms())
            hf.check(L.lib().hf_fields_merge(ctx.handle, C.c_void_p(sym.data_ptr()),
                                             C.c_void_p(side_ptr), n, layout,
                                             C.c_void_p(out.data_ptr())), "hf_fields_merge")
            ctx.sync()
            tm.append(ctx.last_kernel_ms())
        ok = torch.equal(out.view(torch.uint8), w.view(torch.uint8))
        moved = n * eb + n + nside  # bytes read + written by either kernel
        print(f"{name:22s} n={n} ok={ok} split={min(ts):.3f}ms({moved / min(ts) / 1e6:.0f}GB/s moved) "
              f"merge={min(tm):.3f}ms({moved / min(tm) / 1e6:.0f}GB/s moved)", flush=True)
        del w, sym, side, out


if __name__ == "__main__":
    main()
