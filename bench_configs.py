"""The single-GPU workloads of bench.py besides the headline one (`bench.py --config 1|3|5`).

config 1  one 4 KB block of uniform random bytes (the only size the reference's RTL speaks about):
          histogram -> codebook -> encode -> decode, latency per block and blocks per second
config 3  fp8 E4M3 KV-cache pages (16 tokens x 1024 bytes), one codebook per page: per-page
          histogram -> batched codebooks -> encode -> decode, plus decode-on-read of 4096 random
          pages into a page pool
config 5  the codebook kernels alone on the degenerate count shapes: microseconds per codebook
          (one stream-wide codebook) and codebooks per second (batched per-block path)

Each prints one JSON line of the same shape as the headline run, with `roofline` (or
`"bound": "latency"` where a kernel's time is fixed by a serial chain, not by bytes moved) and
`cpu_baseline` (the CPU oracle, a scalar C restatement, on a bounded sample).
"""
import ctypes as C
import json
import sys
import time

import bench

CH = 16384
# Derived from the reference's state machines (not simulated, there is no simulator here):
# SymbolStat n + 2, SymbolSort 2N + 3, ShannonEntropy N + 2, HuffmanTreeBuilder N + 2, tree_adjust
# ~N, code generation 2, the lookup mux n cycles (symbol_stat.scala:34-38,
# symbol_sort.scala:138-141, shannon_entropy.scala:67-97, tree_build.scala:111-125,
# tree_adjust.scala:108-119, top_encoder.scala:148-162): about 2 n + 1300 cycles per block.
RTL_CYCLES_4K = 2 * 4096 + 1300


def _events(torch, n):
    return [torch.cuda.Event(enable_timing=True) for _ in range(n)]


def _phase_ms(evs, names):
    k = len(evs)
    return {p: sum(e[i].elapsed_time(e[i + 1]) for e in evs) / k for i, p in enumerate(names)}


def _phase_median(evs, names):
    return {p: sorted(e[i].elapsed_time(e[i + 1]) for e in evs)[len(evs) // 2]
            for i, p in enumerate(names)}


def _roof(alg_bytes, ms, bound="hbm", **extra):
    ach = alg_bytes / (ms * 1e-3) / 1e9
    d = {"bound": bound, "achieved": ach, "peak": bench.HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": ach / bench.HBM_PEAK_GBPS, "traffic": None, "ms": ms,
         "algorithmic_bytes": alg_bytes}
    d.update(extra)
    return d


def config1(args):
    """4096 symbols, uniform over 0..255, seed 1, Huffman forced."""
    import numpy as np
    import torch
    from huffman_amd import api as hf
    from oracle import oracle as orc
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n = 4096
    sym = np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8)
    t = torch.from_numpy(sym).to(dev)
    ctx = hf.Context(0)
    system = hf.EntropyAwareHuffmanSystem(ctx=ctx)
    decoder = hf.ParrallelHuffmanDecoder(ctx=ctx)
    freq = torch.zeros(256, dtype=torch.int64, device=dev)
    cb = hf.DeviceCodebook(ctx)
    payload = torch.empty(n // 4 + 192 + 64, dtype=torch.int32, device=dev)
    off = torch.empty(4, dtype=torch.int32, device=dev)
    out = torch.empty(n, dtype=torch.uint8, device=dev)

    def block(ev=None):
        # SymbolStat and the codebook in one launch (hf_system_codebook_block_dev), then the packer
        # and the decoder: three calls per block
        if ev: ev[0].record()
        system.codebook_of_block(t, hf.HF_MODE_HUFFMAN, freq256=freq, out=cb)
        if ev: ev[1].record()
        s = system.encode_with(t, cb, payload=payload, chunk_off=off, sync=False)
        if ev: ev[2].record()
        decoder.decode(s, out=out, check_stream=False)
        if ev: ev[3].record()
        return s

    for _ in range(max(args.warmup, 3)):
        s = block()
    ctx.sync()
    steps = max(args.steps, 20)
    # latency: one block at a time, the host waits for each
    lat = []
    for _ in range(steps):
        t0 = time.perf_counter()
        block()
        ctx.sync()
        lat.append(time.perf_counter() - t0)
    # device time per phase
    evs = [_events(torch, 4) for _ in range(steps)]
    for k in range(steps):
        block(evs[k])
    torch.cuda.synchronize()
    ms = _phase_ms(evs, ["codebook", "encode", "decode"])

    # throughput: blocks issued back to back, one wait at the end
    nb = 512
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nb):
        block()
    ctx.sync()
    batched = time.perf_counter() - t0

    # the same four calls captured once into a HIP graph and replayed (none of them allocates or
    # synchronises after the first use of a context, so the sequence is capturable as it stands)
    graph_us = graph_lat_us = None
    try:
        side = torch.cuda.Stream()
        ctx_g = hf.Context(0, stream=side)
        sys_g, dec_g = hf.EntropyAwareHuffmanSystem(ctx=ctx_g), hf.ParrallelHuffmanDecoder(ctx=ctx_g)
        cb_g = hf.DeviceCodebook(ctx_g)
        out_g = torch.empty(n, dtype=torch.uint8, device=dev)

        def block_g():
            sys_g.codebook_of_block(t, hf.HF_MODE_HUFFMAN, freq256=freq, out=cb_g)
            sg = sys_g.encode_with(t, cb_g, payload=payload, chunk_off=off, sync=False)
            dec_g.decode(sg, out=out_g, check_stream=False)

        with torch.cuda.stream(side):
            for _ in range(3):
                block_g()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            block_g()
        out_g.zero_()
        g.replay()
        torch.cuda.synchronize()
        if not torch.equal(out_g, t):
            raise RuntimeError("replayed graph decoded something else")
        glat = []
        for _ in range(steps):
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            glat.append(time.perf_counter() - t0)
        graph_lat_us = 1e6 * sorted(glat)[len(glat) // 2]
        t0 = time.perf_counter()
        for _ in range(nb):
            g.replay()
        torch.cuda.synchronize()
        graph_us = 1e6 * (time.perf_counter() - t0) / nb
    except Exception as e:  # reported, not fatal: the plain path above is the measurement
        print(f"config 1: graph capture not measured ({type(e).__name__}: {e})", file=sys.stderr)

    # outside the timed region: the stream against the oracle, byte for byte
    s = block()
    ctx.sync()
    f = orc.aw_symbol_stat(sym, 256, 64)
    code, lens = orc.rp_system_codebook(f, 64, 0, 16)
    ref_payload, ref_off = orc.rp_encode(sym, code, lens)
    total_words = int(off[1].item())
    gcode, glens, mode, ent = cb.download()
    ok = (np.array_equal(gcode, code) and np.array_equal(glens, lens)
          and total_words == ref_payload.size
          and np.array_equal(payload[:total_words].cpu().numpy().view(np.uint32), ref_payload)
          and np.array_equal(out.cpu().numpy(), sym))
    lbar = float((f * lens).sum()) / n

    # the oracle on the same block
    reps = 4000
    t0 = time.perf_counter()
    for _ in range(reps):
        f2 = orc.aw_symbol_stat(sym, 256, 64)
        c2, l2 = orc.rp_system_codebook(f2, 64, 0, 16)
        p2, o2 = orc.rp_encode(sym, c2, l2)
        orc.rp_decode(p2, o2, n, c2, l2)
    cpu_s = (time.perf_counter() - t0) / reps

    dev_us = sum(ms.values()) * 1e3
    dominant = max(ms, key=ms.get)
    alg = {"codebook": n + 2048 + 36864, "encode": n * (1 + lbar / 8) + 4,
           "decode": n * (lbar / 8 + 1) + 4}
    line = {
        "metric": bench.METRIC, "value": nb * n / batched / 1e9, "unit": "GB/s", "n_gpus": 1,
        "steps": steps, "warmup": max(args.warmup, 3), "ms_per_step": batched / nb * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "config 1: one 4 KB block, 4096 symbols uniform over 0..255 "
                               "(seed 1), 256-symbol alphabet, Huffman forced, Lmax=16; "
                               "histogram + codebook (one launch) -> encode -> decode per block",
                   "block_symbols": n},
        "us_per_block_latency_host": 1e6 * sorted(lat)[len(lat) // 2],
        "us_per_block_device": dev_us,
        "us_per_phase_device": {k: v * 1e3 for k, v in ms.items()},
        "blocks_per_s_back_to_back": nb / batched,
        "us_per_block_graph_replay_latency_host": graph_lat_us,
        "us_per_block_graph_replay_back_to_back": graph_us,
        "reference_rtl_cycles_per_block_derived_not_simulated": RTL_CYCLES_4K,
        "reference_rtl_us_per_block_at_1GHz_derived": RTL_CYCLES_4K / 1e3,
        "compression_mode": mode, "entropy_q16": ent, "bits_per_symbol_ideal": lbar,
        "bits_per_symbol_stream": 32.0 * total_words / n,
        "compression_ratio": n / (4 * total_words + 8 + 768),
        "bit_exact_vs_oracle": bool(ok), "library_sha256_16": bench.library_hash(),
        "roofline": _roof(alg[dominant], ms[dominant], bound="latency", kernel=dominant,
                          note="a 4 KB block occupies one wavefront per kernel: every phase is a "
                               "launch plus a serial chain, not a bandwidth-bound transfer"),
        "cpu_baseline": {"value": n / cpu_s / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                         "sample": f"the same 4 KB block, {reps} repetitions, "
                                   "histogram+codebook+encode+decode, single thread, gcc -O3",
                         "us_per_block": cpu_s * 1e6},
    }
    print(json.dumps(line), flush=True)
    if not ok:
        raise SystemExit("config 1: stream differs from the oracle's")


def make_kv_pages(torch, pages, dev, seed=7):
    """K/V activations N(0,1) x per-channel scale -> fp8 E4M3, 16 tokens x 1024 channels a page."""
    g = torch.Generator(device=dev).manual_seed(seed)
    scale = torch.exp(torch.randn(1024, generator=g, device=dev) * 0.7)
    t = torch.empty(pages * CH, dtype=torch.uint8, device=dev)
    step = 4096
    for lo in range(0, pages, step):
        hi = min(pages, lo + step)
        x = torch.randn((hi - lo) * 16, 1024, generator=g, device=dev) * scale
        t[lo * CH:hi * CH] = x.to(torch.float8_e4m3fn).view(torch.uint8).reshape(-1)
    return t


def config3(args):
    """fp8 KV pages with per-page codebooks through the paged stream (format version 2: pages with
    their own records, sizes from the histograms, four sub-decoders per page); the dense and the
    slotted per-block streams are timed beside it and never part of `value`."""
    import numpy as np
    import torch
    from huffman_amd import _lib as L
    from huffman_amd import api as hf
    from oracle import oracle as orc
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    pages = args.pages
    t = make_kv_pages(torch, pages, dev)
    n = t.numel()
    ctx = hf.Context(0)
    lib, h = L.lib(), ctx.handle
    P = lambda x: C.c_void_p(x.data_ptr())
    freq = torch.empty(pages * 256, dtype=torch.int32, device=dev)
    cbs = torch.empty(pages * L.HF_CHUNK_CODEBOOK_BYTES, dtype=torch.uint8, device=dev)
    payload = torch.empty(n // 4 + pages * 192 + 64, dtype=torch.int32, device=dev)
    off = torch.empty(pages + 4, dtype=torch.int32, device=dev)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    n_read = min(4096, pages)
    ids = torch.randperm(pages, device=dev, generator=torch.Generator(device=dev).manual_seed(3))[
        :n_read].to(torch.int32)
    slots = torch.randperm(n_read, device=dev,
                           generator=torch.Generator(device=dev).manual_seed(4)).to(torch.int32)
    pool = torch.empty(n_read * CH, dtype=torch.uint8, device=dev)
    L.check(lib.hf_ctx_reserve(h, n, L.HF_RESERVE_ALL), "reserve")

    def step(ev=None):
        if ev: ev[0].record()
        L.check(lib.hf_symbol_stat_chunks_dev(h, P(t), n, 0, P(freq)), "stat")
        if ev: ev[1].record()
        L.check(lib.hf_system_codebook_chunks_dev(h, P(freq), pages, 1, 0, P(cbs)), "codebooks")
        if ev: ev[2].record()
        L.check(lib.hf_encode_pages(h, P(t), n, 0, P(cbs), P(freq), None, P(payload),
                                    payload.numel(), P(off), None), "encode")
        if ev: ev[3].record()
        L.check(lib.hf_decode_pages(h, P(payload), payload.numel(), P(off), None, n, 0, None, None,
                                    0, P(out), 0, 0), "decode")
        if ev: ev[4].record()
        L.check(lib.hf_decode_pages(h, P(payload), payload.numel(), P(off), None, n, 0, P(ids),
                                    P(slots), n_read, P(pool), n_read, 0), "read")
        if ev: ev[5].record()

    names = ["hist", "codebook", "encode", "decode", "decode_on_read"]
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    evs = [_events(torch, 6) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = _phase_ms(evs, names)
    med = _phase_median(evs, names)

    ctx.sync()  # reports in-kernel errors of the enqueue-only calls
    total_words = int(off[pages].item()) & 0xFFFFFFFF
    ok = bool(torch.equal(out, t))
    ok = ok and bool(torch.equal(pool.view(n_read, CH)[slots.long()],
                                 t.view(pages, CH)[ids.long()]))
    # bits of the codes alone (header word 0 of every page holds them), the stored size in full
    o64 = off[:pages].long()
    bits = float((payload[o64].long() >> 10).sum().item()) if pages else 0.0
    lbar = bits / n
    lbar_stream = 32.0 * total_words / n
    comp = 4 * total_words + 4 * (pages + 1)

    alg = {"hist": float(n), "encode": n * (1 + lbar_stream / 8) + 4.0 * pages,
           "decode": n * (lbar_stream / 8 + 1) + 4.0 * pages,
           "decode_on_read": n_read * CH * (lbar_stream / 8 + 1)}
    roof = {p: _roof(alg[p], ms[p], ms_median=med[p]) for p in alg}
    dominant = "encode" if ms["encode"] >= ms["decode"] else "decode"
    line = {
        "metric": bench.METRIC, "value": n * args.steps / elapsed / 1e9, "unit": "GB/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"config 3: {pages} fp8 E4M3 KV-cache pages of 16 tokens x 1024 "
                               f"bytes ({n / 1e9:.2f} GB), N(0,1) x per-channel scale, one "
                               "EntropyAwareTreeBuilder codebook per page (Huffman forced), paged "
                               "stream (every page carries its record; the stored form), "
                               f"decode of all pages plus decode-on-read of {n_read} random pages "
                               "into a page pool", "pages": pages, "page_bytes": CH},
        "encode_GBps": n / ((ms["hist"] + ms["codebook"] + ms["encode"]) * 1e-3) / 1e9,
        "encode_kernel_GBps": n / (ms["encode"] * 1e-3) / 1e9,
        "decode_GBps": n / (ms["decode"] * 1e-3) / 1e9,
        "hist_GBps": n / (ms["hist"] * 1e-3) / 1e9,
        "codebooks_per_s": pages / (ms["codebook"] * 1e-3),
        "decode_on_read_us": ms["decode_on_read"] * 1e3,
        "decode_on_read_GBps": n_read * CH / (ms["decode_on_read"] * 1e-3) / 1e9,
        "compression_ratio": n / comp, "bits_per_symbol_stream": lbar_stream,
        "bits_per_symbol_ideal": lbar, "roundtrip_bit_exact": ok,
        "ms_per_phase_mean": ms, "ms_per_phase_median": med,
        "library_sha256_16": bench.library_hash(),
        "roofline": dict(roof[dominant], kernel=dominant), "roofline_all": roof,
    }
    # one page alone: the latency floor of decode-on-read
    one = _events(torch, 2)
    lat = []
    for _ in range(20):
        one[0].record()
        L.check(lib.hf_decode_pages(h, P(payload), payload.numel(), P(off), None, n, 0, P(ids),
                                    P(slots), 1, P(pool), n_read, 0), "read one")
        one[1].record()
        torch.cuda.synchronize()
        lat.append(one[0].elapsed_time(one[1]) * 1e3)
    line["decode_on_read_one_page_us"] = sorted(lat)[len(lat) // 2]

    # The same pages as a dense per-block stream (look-back placement, records beside the payload)
    # and in slots sized from their histograms: timed beside the step above.
    payload_s = torch.empty(pages * L.HF_SLOT_WORDS(CH * 8), dtype=torch.int32, device=dev)
    off_s = torch.empty(pages + 4, dtype=torch.int32, device=dev)
    words_s = torch.empty(pages + 4, dtype=torch.int32, device=dev)

    def other(kind):
        slotted = kind == "slotted"

        def step_other(ev=None):
            if ev: ev[0].record()
            if slotted:
                L.check(lib.hf_encode_chunks_slotted(h, P(t), n, 0, P(cbs), P(freq), None,
                                                     P(payload_s), payload_s.numel(), P(off_s),
                                                     P(words_s), None), "encode_slotted")
            else:
                L.check(lib.hf_encode_chunks(h, P(t), n, 0, P(cbs), None, P(payload_s),
                                             payload_s.numel(), P(off_s), None), "encode_dense")
            if ev: ev[1].record()
            L.check(lib.hf_decode_chunks_slotted(h, P(payload_s), payload_s.numel(), P(off_s),
                                                 P(words_s) if slotted else None, None, n, 0,
                                                 P(cbs), None, None, 0, P(out), 0, 0)
                    if slotted else
                    lib.hf_decode_chunks(h, P(payload_s), payload_s.numel(), P(off_s), None, n, 0,
                                         P(cbs), None, 0, P(out), 0), "decode_other")
            if ev: ev[2].record()
            L.check(lib.hf_decode_chunks_slotted(h, P(payload_s), payload_s.numel(), P(off_s),
                                                 P(words_s), None, n, 0, P(cbs), P(ids), P(slots),
                                                 n_read, P(pool), n_read, 0)
                    if slotted else
                    lib.hf_decode_chunks_to(h, P(payload_s), payload_s.numel(), P(off_s), None, n,
                                            0, P(cbs), P(ids), P(slots), n_read, P(pool), n_read,
                                            0), "read_other")
            if ev: ev[3].record()

        out.zero_()
        pool.zero_()
        for _ in range(args.warmup):
            step_other()
        torch.cuda.synchronize()
        evs2 = [_events(torch, 4) for _ in range(args.steps)]
        for k in range(args.steps):
            step_other(evs2[k])
        torch.cuda.synchronize()
        ctx.sync()
        m2 = _phase_ms(evs2, ["encode", "decode", "decode_on_read"])
        w2 = int(off_s[pages].item()) & 0xFFFFFFFF
        good = bool(torch.equal(out, t)) and bool(torch.equal(
            pool.view(n_read, CH)[slots.long()], t.view(pages, CH)[ids.long()]))
        stored = 4 * w2 + 4 * (pages + 1) + (4 * pages if slotted else 0) + \
            L.HF_CONTAINER_CHUNK_CODEBOOK_BYTES * pages
        return good, {
            "encode_kernel_GBps": n / (m2["encode"] * 1e-3) / 1e9,
            "decode_GBps": n / (m2["decode"] * 1e-3) / 1e9,
            "decode_on_read_us": m2["decode_on_read"] * 1e3,
            "payload_words_over_paged": w2 / max(1, total_words),
            "compression_ratio": n / stored, "roundtrip_bit_exact": good,
            "roofline_encode": _roof(n + 4.0 * w2, m2["encode"]),
            "roofline_decode": _roof(n + 4.0 * w2, m2["decode"]),
        }
    for kind in ("dense", "slotted"):
        good, line[kind] = other(kind)
        ok = ok and good
    if not args.no_cpu_baseline:
        m = min(pages, 12288) * CH  # 0.2 GB: about 20 s of one core
        sy = t[:m].cpu().numpy()
        t0 = time.perf_counter()
        p2, o2, c2, l2 = orc.rp_encode_pages(sy, 0, 1)
        t1 = time.perf_counter()
        back = orc.rp_decode_pages(p2, o2, m)
        t2 = time.perf_counter()
        assert np.array_equal(back, sy)
        same = bool(np.array_equal(payload[:p2.size].cpu().numpy().view(np.uint32), p2))
        same = same and bool(np.array_equal(off[:m // CH + 1].cpu().numpy().view(np.uint32), o2))
        line["cpu_baseline"] = {
            "value": m / (t2 - t0) / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"the first {m // CH} pages of the same tensor ({m / 1e9:.2f} GB), per-page "
                      "histogram+codebook+paged encode (sizing pass included)+decode, single "
                      "thread, gcc -O3",
            "encode_GBps": m / (t1 - t0) / 1e9, "decode_GBps": m / (t2 - t1) / 1e9,
            "seconds": t2 - t0, "gpu_payload_equals_oracle_on_sample": same}
        ok = ok and same
    print(json.dumps(line), flush=True)
    if not ok:
        raise SystemExit("config 3: round trip or oracle mismatch")


def _fib_counts(np):
    f = np.zeros(256, np.uint64)
    a, b = 1, 1
    for i in range(256):
        f[i] = min(a, 1 << 40)
        a, b = b, a + b
    return f


def _block_counts(np, kind, rng, total=CH):
    if kind == "fibonacci":
        p = _fib_counts(np).astype(np.float64)[:40]
        sym = rng.choice(40, total, p=p / p.sum())
    elif kind == "geometric":
        sym = np.minimum(rng.geometric(0.5, total) - 1, 255)
    elif kind == "one_symbol":
        sym = np.full(total, 7)
    elif kind == "two_symbols":
        sym = (rng.random(total) < 0.001).astype(np.int64)
    elif kind == "all_equal":
        sym = np.arange(total) % 256
    elif kind == "all_zero":
        return np.zeros(256, np.uint32)
    else:
        raise ValueError(kind)
    return np.bincount(sym, minlength=256).astype(np.uint32)


def config5(args):
    import numpy as np
    import torch
    from huffman_amd import _lib as L
    from huffman_amd import api as hf
    from oracle import oracle as orc
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = hf.Context(0)
    lib, h = L.lib(), ctx.handle
    P = lambda x: C.c_void_p(x.data_ptr())
    rng = np.random.default_rng(5)
    blocks = 65536
    reps = max(args.steps, 5)
    shapes = {}
    ok = True
    cpu_us = []
    for kind in ("fibonacci", "geometric", "one_symbol", "two_symbols", "all_equal", "all_zero"):
        f64 = _fib_counts(np) if kind == "fibonacci" \
            else _block_counts(np, kind, rng).astype(np.uint64) * 4096  # a 64 MiB block's counts
        d_f = torch.from_numpy(f64.view(np.int64)).to(dev)
        cb = hf.DeviceCodebook(ctx)
        evs = [_events(torch, 2) for _ in range(reps)]
        for k in range(args.warmup + reps):
            e = evs[k - args.warmup] if k >= args.warmup else None
            if e: e[0].record()
            L.check(lib.hf_system_codebook_dev(h, P(d_f), 1, 0, C.c_void_p(cb.ptr)), "codebook")
            if e: e[1].record()
        torch.cuda.synchronize()
        single_us = min(e[0].elapsed_time(e[1]) for e in evs) * 1e3
        code, lens, mode, _ = cb.download()
        ec, el = orc.rp_system_codebook(f64, 64, 0, 16)
        kraft = float(np.sum(2.0 ** -lens[lens > 0].astype(np.float64)))
        good = bool(np.array_equal(code, ec) and np.array_equal(lens, el) and lens.max() <= 16
                    and kraft <= 1.0 + 1e-12)
        # batched: one codebook per 16384-symbol block, 64 different blocks tiled
        few = np.stack([_block_counts(np, kind, rng) for _ in range(64)])
        counts = np.tile(few, (blocks // 64, 1))
        d_c = torch.from_numpy(counts.view(np.int32).reshape(-1)).to(dev)
        cbs = torch.empty(blocks * L.HF_CHUNK_CODEBOOK_BYTES, dtype=torch.uint8, device=dev)
        evs = [_events(torch, 2) for _ in range(reps)]
        for k in range(args.warmup + reps):
            e = evs[k - args.warmup] if k >= args.warmup else None
            if e: e[0].record()
            L.check(lib.hf_system_codebook_chunks_dev(h, P(d_c), blocks, 1, 0, P(cbs)), "batched")
            if e: e[1].record()
        torch.cuda.synchronize()
        batch_ms = min(e[0].elapsed_time(e[1]) for e in evs)
        st = hf.EncodedStream(n=blocks * CH, layout=0, dtype=torch.uint8, payload=None,
                              chunk_off=None, side=None, codebook=None, chunk_codebooks=cbs)
        for c in (0, 63):
            bc, bl, _, _ = hf.EntropyAwareHuffmanSystem(ctx=ctx).block_codebook(st, c)
            oc, ol = orc.rp_system_codebook(few[c].astype(np.uint64), 64, 0, 16)
            good = good and bool(np.array_equal(bc, oc) and np.array_equal(bl, ol))
        t0 = time.perf_counter()
        for _ in range(2000):
            orc.rp_system_codebook(f64, 64, 0, 16)
        cu = (time.perf_counter() - t0) / 2000 * 1e6
        cpu_us.append(cu)
        shapes[kind] = {"us_per_codebook_single": single_us, "ms_per_65536_batched": batch_ms,
                        "codebooks_per_s_batched": blocks / (batch_ms * 1e-3),
                        "max_code_length": int(lens.max()), "kraft_sum": kraft,
                        "equals_oracle": good, "cpu_oracle_us_per_codebook": cu}
        ok = ok and good
    worst = max(shapes.values(), key=lambda d: d["us_per_codebook_single"])
    slowest_batch = max(d["ms_per_65536_batched"] for d in shapes.values())
    line = {
        "metric": "codebook build on degenerate count shapes: codebooks per second (batched "
                  "per-block path, slowest shape) and microseconds per stream-wide codebook",
        "value": blocks / (slowest_batch * 1e-3), "unit": "codebooks/s", "n_gpus": 1,
        "steps": reps, "warmup": args.warmup, "ms_per_step": slowest_batch,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "config 5: 256 counts -- Fibonacci weights (unconstrained depth far "
                               "above 16), geometric p = 0.5, one symbol, two symbols 99.9/0.1 %, "
                               "all equal, all zero; Lmax = 16; one stream-wide codebook (counts "
                               "of a 64 MiB block) and 65536 per-block codebooks per launch"},
        "shapes": shapes, "all_equal_oracle": bool(ok),
        "library_sha256_16": bench.library_hash(),
        "roofline": _roof(2048 + 36864, worst["us_per_codebook_single"] * 1e-3, bound="latency",
                          kernel="hf_k_system",
                          note="one workgroup; the time is a chain of wave-wide steps (sort, "
                               "merge runs, limiter, tables), not bytes moved"),
        "cpu_baseline": {"value": 1e6 / max(cpu_us), "unit": "codebooks/s", "cores": 1,
                         "kind": "port",
                         "sample": "2000 stream-wide codebooks per shape (sort, tree, limiter, "
                                   "canonical codes), slowest shape, single thread, gcc -O3"},
    }
    print(json.dumps(line), flush=True)
    if not ok:
        raise SystemExit("config 5: a codebook differs from the oracle's")
