#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_vectors.json from the CPU oracle.

The reference (Scala/Chisel) cannot run in this image, so these vectors are produced by the
oracle's restatement of it; they pin the oracle (and, through the GPU parity tests, the HIP
kernels) against silent drift.  The four known-answer vectors the reference's own chiseltest
specs hold are kept separately in reference_kats.json, typed in from the spec sources.

usage: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as O  # noqa: E402


def fib_counts(k=40, N=256):
    fib = [1, 1]
    while len(fib) < k:
        fib.append(fib[-1] + fib[-2])
    f = np.zeros(N, np.uint64)
    f[:k] = fib[::-1]
    return f


def L(a):
    return [int(x) for x in a]


def main():
    out = {}

    # --- HuffmanTreeBuilder small cases (tree_build.scala:47-125) ---------------------------
    trees = []
    for fr in ([5, 1, 3, 2], [5, 3, 2, 1], [0, 0, 0, 0], [8, 4, 2, 1, 1, 0, 0, 0], [1] * 8,
               [13, 8, 5, 3, 2, 1, 1, 0], [255, 255, 255, 255], [3, 3, 3, 3, 3, 3, 3, 3, 1]):
        N = len(fr)
        t = O.aw_tree_build(fr, W=32)
        d = O.rp_leaf_depths(t["lefts"], t["rights"], t["root"], N)
        trees.append(dict(freqs=fr, lefts=L(t["lefts"]), rights=L(t["rights"]),
                          weights=L(t["weights"]), root=t["root"], heights=L(t["heights"]),
                          leaf_depths=L(d)))
    # weights that wrap at W = 8 (the RTL adds in W bits, tree_build.scala:115)
    t = O.aw_tree_build([200, 100, 50, 25], W=8)
    trees.append(dict(freqs=[200, 100, 50, 25], W=8, lefts=L(t["lefts"]), rights=L(t["rights"]),
                      weights=L(t["weights"]), root=t["root"], heights=L(t["heights"])))
    out["trees"] = trees

    # --- entropy (shannon_entropy.scala:58-109) ----------------------------------------------
    ent = []
    rng = np.random.default_rng(2024)
    cases = []
    f = np.zeros(256, np.uint64); f[7] = 4096; cases.append(("one symbol x4096", f))
    cases.append(("uniform 16 x 256", np.full(256, 16, np.uint64)))
    f = np.zeros(256, np.uint64); f[1] = f[9] = 2048; cases.append(("two symbols 2048/2048", f))
    cases.append(("all zero", np.zeros(256, np.uint64)))
    cases.append(("fibonacci", fib_counts()))
    for i in range(4):
        cases.append((f"random {i}", rng.integers(0, 1 << (6 + 5 * i), 256).astype(np.uint64)))
    for name, f in cases:
        e = O.aw_entropy_q16(f, 32)
        ent.append(dict(name=name, freq=L(f), entropy_q16=e, mode=O.aw_adaptive_mode(e)))
    f16 = np.array([3, 1, 7, 0, 5, 2, 6, 4, 9, 8, 15, 12, 10, 11, 14, 13], np.uint64)
    ent.append(dict(name="N=16 W=8", freq=L(f16), W=8, entropy_q16=O.aw_entropy_q16(f16, 8)))
    out["entropy"] = ent

    # --- tree_adjust (repaired) --------------------------------------------------------------
    adj = []
    for name, f in (("fibonacci", fib_counts()), ("geometric",
                    np.array([1 << max(0, 30 - i) if i < 31 else 0 for i in range(256)], np.uint64)),
                    ("single symbol", np.eye(1, 256, 5, dtype=np.uint64)[0] * 1000),
                    ("all equal", np.full(256, 9, np.uint64))):
        srt, _ = O.aw_symbol_sort(f)
        t = O.aw_tree_build(srt, 64)
        d = O.rp_leaf_depths(t["lefts"], t["rights"], t["root"], 256)
        for lmax in (16, 12):
            a, steps = O.rp_tree_adjust(d, lmax)
            adj.append(dict(name=name, lmax=lmax, depths_in=L(d), depths_out=L(a),
                            migrate_steps=steps))
    out["tree_adjust"] = adj

    # --- codebooks -----------------------------------------------------------------------------
    cbs = []
    for name, f in (("t_tree_builder 5,1,3,2", np.array([5, 1, 3, 2], np.uint64)),
                    ("zeros N=4", np.zeros(4, np.uint64)),
                    ("N=8", np.array([8, 4, 2, 1, 1, 0, 0, 0], np.uint64))):
        code, lens = O.rp_codebook(f, 32, 0, 16)
        cbs.append(dict(name=name, freq=L(f), code=L(code), len=L(lens),
                        code_as_written=L(O.aw_codes_as_written(lens.astype(np.uint16)))))
    for name, f in (("fibonacci by symbol", np.roll(fib_counts(), 100)),
                    ("random", rng.integers(0, 5000, 256).astype(np.uint64))):
        code, lens = O.rp_system_codebook(f, 64, 0, 16)
        cbs.append(dict(name=name, system=True, freq=L(f), code=L(code), len=L(lens)))
    out["codebooks"] = cbs

    # --- packed stream ---------------------------------------------------------------------------
    streams = []
    for name, sym in (("3 symbols", np.array([1, 2, 1], np.uint8)),
                      ("t_symbol_stat stream", np.array([3, 3, 5, 5, 5, 15], np.uint8)),
                      ("600 geometric", np.minimum(rng.geometric(0.4, 600) - 1, 255).astype(np.uint8)),
                      ("16385 ramp", (np.arange(16385) % 7).astype(np.uint8))):
        f = O.aw_symbol_stat(sym, 256, 64)
        code, lens = O.rp_system_codebook(f, 64, 0, 16)
        payload, off = O.rp_encode(sym, code, lens)
        rec = dict(name=name, code=L(code), len=L(lens), chunk_off=L(off),
                   payload_words=int(payload.size),
                   payload_crc=int(np.bitwise_xor.reduce(payload * np.arange(1, payload.size + 1,
                                                                             dtype=np.uint32)))
                   if payload.size else 0)
        if sym.size <= 600:
            rec["sym"] = L(sym)
            rec["payload"] = L(payload)
        streams.append(rec)
    out["streams"] = streams

    # --- quality ----------------------------------------------------------------------------------
    f = rng.integers(0, 5000, 256).astype(np.uint64)
    _, lens = O.rp_system_codebook(f, 64, 0, 16)
    out["quality"] = dict(freq=L(f), len=L(lens), result=list(O.aw_quality(f, lens, 32)))

    # --- paged streams (format version 2): pages with their records, pages without -----------
    prng = np.random.default_rng(77)
    pages = []

    def crc(w):
        return int(np.bitwise_xor.reduce(w * np.arange(1, w.size + 1, dtype=np.uint32))) if w.size else 0

    for name, sym in (("3 symbols", np.array([1, 2, 1], np.uint8)),
                      ("200 geometric", np.minimum(prng.geometric(0.3, 200) - 1, 255).astype(np.uint8)),
                      ("16385 ramp", (np.arange(16385) % 7).astype(np.uint8)),
                      ("3 pages mixed", np.concatenate([
                          np.minimum(prng.geometric(0.05, 16384) - 1, 255),
                          prng.integers(0, 256, 16384),
                          np.minimum(prng.geometric(0.6, 9000) - 1, 255)]).astype(np.uint8))):
        rp, roff, _, _ = O.rp_encode_pages(sym, 0, 1)
        f = O.aw_symbol_stat(sym, 256, 64)
        code, lens = O.rp_system_codebook(f, 64, 0, 16)
        sp, soff = O.rp_encode_pages_shared(sym, code, lens)
        rec = dict(name=name, n=int(sym.size), record_page_off=L(roff), record_words=int(rp.size),
                   record_crc=crc(rp), shared_code=L(code), shared_len=L(lens),
                   shared_page_off=L(soff), shared_words=int(sp.size), shared_crc=crc(sp))
        if sym.size <= 200:
            rec["sym"] = L(sym)
            rec["record_payload"] = L(rp)
            rec["shared_payload"] = L(sp)
        else:
            rec["sym_seed"] = name
        pages.append(rec)
    out["pages"] = pages

    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as fo:
        json.dump(out, fo, separators=(",", ":"))
    print("wrote", os.path.join(HERE, "oracle_vectors.json"))


if __name__ == "__main__":
    main()
