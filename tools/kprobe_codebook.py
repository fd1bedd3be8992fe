#!/usr/bin/env python3
"""Time the codebook kernels alone on the stress distributions (developer probe).

Single codebook (stat counts -> sort -> entropy -> tree -> limiter -> canonical codes -> decode
tables, one wavefront) in microseconds per codebook, and the batched per-block kernel in
codebooks per second, for: Fibonacci weights (unconstrained depth far above 16), geometric
p = 0.5, one symbol, two symbols 99.9 % / 0.1 %, all equal, all zero.
usage: python tools/kprobe_codebook.py [--blocks N] [--reps R]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from huffman_amd import _lib as L
from huffman_amd import api as hf


def fib_counts():
    f = np.zeros(256, np.uint64)
    a, b = 1, 1
    for i in range(256):
        f[i] = min(a, (1 << 40))
        a, b = b, a + b
    return f


def block_counts(kind, rng, total=16384):
    """uint32[256] counts of one 16384-symbol block."""
    if kind == "fib":
        p = fib_counts().astype(np.float64)[:40]
        sym = rng.choice(40, total, p=p / p.sum())
    elif kind == "geometric":
        sym = np.minimum(rng.geometric(0.5, total) - 1, 255)
    elif kind == "one":
        sym = np.full(total, 7)
    elif kind == "two":
        sym = (rng.random(total) < 0.001).astype(np.int64)
    elif kind == "equal":
        sym = np.arange(total) % 256
    elif kind == "zero":
        return np.zeros(256, np.uint32)
    else:
        sym = rng.integers(0, 256, total)
    return np.bincount(sym, minlength=256).astype(np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=1 << 16)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = hf.Context(0)
    ctx.set_timing(True)
    lib, h = L.lib(), ctx.handle
    rng = np.random.default_rng(5)
    P = lambda x: C.c_void_p(x.data_ptr())
    print(f"{'kind':10s} {'single us':>10s} {'batched ms':>11s} {'Mcodebooks/s':>13s}  max_len")
    for kind in ("fib", "geometric", "one", "two", "equal", "zero", "uniform"):
        # single: the stream-wide codebook kernel with 64-bit counts
        f64 = fib_counts() if kind == "fib" else block_counts(kind, rng).astype(np.uint64) * 4096
        d_f = torch.from_numpy(f64.view(np.int64)).to(dev)
        cb = hf.DeviceCodebook(ctx)
        t_single = []
        for _ in range(a.reps):
            L.check(lib.hf_system_codebook_dev(h, P(d_f), 1, 0, C.c_void_p(cb.ptr)), "codebook")
            t_single.append(ctx.last_kernel_ms())
        lens = cb.download()[1]
        # batched: one codebook per block, blocks drawn independently
        few = np.stack([block_counts(kind, rng) for _ in range(64)])
        counts = np.tile(few, (a.blocks // 64 + 1, 1))[: a.blocks]
        d_c = torch.from_numpy(counts.view(np.int32).reshape(-1)).to(dev)
        cbs = torch.empty(a.blocks * L.HF_CHUNK_CODEBOOK_BYTES, dtype=torch.uint8, device=dev)
        t_batch = []
        for _ in range(a.reps):
            L.check(lib.hf_system_codebook_chunks_dev(h, P(d_c), a.blocks, 1, 0, P(cbs)), "batched")
            t_batch.append(ctx.last_kernel_ms())
        tb = min(t_batch)
        print(f"{kind:10s} {min(t_single) * 1e3:10.1f} {tb:11.3f} {a.blocks / tb / 1e3:13.2f}  "
              f"{int(lens.max())}")


if __name__ == "__main__":
    main()
