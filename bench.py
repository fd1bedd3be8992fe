#!/usr/bin/env python3
"""Headline benchmark: exponent-byte Huffman encode + decode of a Llama-3-8B-sized synthetic
Gaussian bf16 weight stream on MI355X.

One step = one pass of the hot path over the whole tensor, inputs resident in HBM:
    SymbolStat per 16384-element page, folded into the tensor's 256 counts
    -> [all-reduce of the 256 counts when N > 1: global codebook]
    -> sort/entropy/tree/adjust/canonical codes -> encode -> decode,
the stored form being the paged stream with one codebook (pages without records, sized exactly
from their counts; hf_encode_pages_shared / hf_decode_pages_shared).  The packed stream of format
version 1 (hf_encode / hf_decode) is timed beside it, outside the timed region, as `dense_v1`.
`value` = uncompressed bytes through that round trip per second, summed over all ranks.
Run as `python bench.py --gpus N`: for N > 1 the script starts one rank per GPU itself (through
torch.distributed.run, before anything touches a GPU) unless it already runs under such a
launcher (WORLD_SIZE set), and rank 0 prints the one JSON line.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LLAMA3_8B_PARAMS = 8_030_261_248
SLAB = 256 * 1024 * 1024
HBM_PEAK_GBPS = 8000.0
METRIC = ("encode+decode GB/s (uncompressed bytes through histogram, codebook, encode and decode "
          "per second), % HBM roofline, comp. ratio")


def library_hash():
    """First 16 hex digits of sha256(libhuffhip.so): ties a bench line and a counter summary to
    the build they were taken from."""
    import hashlib
    from huffman_amd import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()[:16]


def make_weights(torch, n, device):
    """fp32 N(0, 0.02^2) -> bf16 (round to nearest even), seed 1234, 256 Mi-element slabs."""
    g = torch.Generator(device=device)
    g.manual_seed(1234)
    out = torch.empty(n, dtype=torch.bfloat16, device=device)
    for lo in range(0, n, SLAB):
        hi = min(n, lo + SLAB)
        slab = torch.randn(hi - lo, generator=g, device=device, dtype=torch.float32)
        out[lo:hi] = slab.mul_(0.02).to(torch.bfloat16)
        del slab
    return out


def cpu_baseline(torch, w_gpu, sample_elems, gpu_code, gpu_lens, gpu_words, gpu_off):
    """The CPU oracle (a scalar C restatement of the same pipeline) on a bounded sample; on the way
    the first pages of the GPU's stream (gpu_words, gpu_off, coded with gpu_code / gpu_lens) are
    compared with the pages the oracle makes of the same elements and the same codebook."""
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    npages = gpu_off.numel() - 1
    mp = min(w_gpu.numel(), npages * 16384)
    sym, _ = orc.symbolize(w_gpu[:mp].view(torch.int16).cpu().numpy().view(np.uint16),
                           orc.LAYOUT_BF16_EXP)
    ref_words, ref_off = orc.rp_encode_pages_shared(sym, gpu_code, gpu_lens)
    pages_ok = bool(np.array_equal(gpu_off.numpy().view(np.uint32), ref_off)) and bool(
        np.array_equal(gpu_words.numpy().view(np.uint32), ref_words))
    m = min(sample_elems, w_gpu.numel())
    w = w_gpu[:m].view(torch.int16).cpu().numpy().view(np.uint16)
    t0 = time.perf_counter()
    sym, side = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
    freq = orc.aw_symbol_stat(sym, 256, 64)
    code, lens = orc.rp_system_codebook(freq, 64, 0, 16)
    t1 = time.perf_counter()
    payload, off = orc.rp_encode_pages_shared(sym, code, lens)  # sizes, then packs: two passes
    t2 = time.perf_counter()
    out = orc.rp_decode_pages_shared(payload, off, m, code, lens)
    back = orc.desymbolize(out, side, orc.LAYOUT_BF16_EXP)
    t3 = time.perf_counter()
    assert np.array_equal(back, w)
    del sym, side, payload, out, back
    gb = 2.0 * m / 1e9
    res = {
        "value": gb / (t3 - t0), "unit": "GB/s", "cores": 1, "kind": "port",
        "sample": f"first {m} bf16 elements of the same tensor ({gb:.2f} GB), "
                  "symbolize+histogram+codebook+encode+decode, single thread, gcc -O3",
        "encode_GBps": gb / (t2 - t0), "decode_GBps": gb / (t3 - t2),
        "seconds": t3 - t0,
        "gpu_pages_equal_port": pages_ok, "gpu_pages_compared": int(npages),
    }
    # the same sample on the host cores this process may use: chunk-aligned shards, one thread
    # each (the C calls release the interpreter lock), one global codebook
    from concurrent.futures import ThreadPoolExecutor
    try:
        ncore = len(os.sched_getaffinity(0))
    except AttributeError:
        ncore = os.cpu_count() or 1
    threads = max(1, min(16, ncore))
    per = ((m // threads) // 16384 + 1) * 16384
    bounds = [(lo, min(m, lo + per)) for lo in range(0, m, per)]

    def front(b):
        s, sd = orc.symbolize(w[b[0]:b[1]], orc.LAYOUT_BF16_EXP)
        return s, sd, orc.aw_symbol_stat(s, 256, 64)

    def back_half(args):
        (s, sd, _), b = args
        p, o = orc.rp_encode_pages_shared(s, code2, lens2)
        d = orc.rp_decode_pages_shared(p, o, s.size, code2, lens2)
        return bool(np.array_equal(orc.desymbolize(d, sd, orc.LAYOUT_BF16_EXP), w[b[0]:b[1]]))

    with ThreadPoolExecutor(threads) as ex:
        u0 = time.perf_counter()
        parts = list(ex.map(front, bounds))
        code2, lens2 = orc.rp_system_codebook(sum(p[2] for p in parts), 64, 0, 16)
        oks = list(ex.map(back_half, zip(parts, bounds)))
        u1 = time.perf_counter()
    assert all(oks) and np.array_equal(code2, code) and np.array_equal(lens2, lens)
    res["all_cores_value"] = gb / (u1 - u0)
    res["all_cores"] = threads
    return res


def pmc_traffic(n, lib_hash):
    """HBM bytes per launch from the committed rocprofv3 counter passes of this same workload,
    taken with this very build of the library (the summary records the library's hash; one from
    another build is not quoted, so that a kernel change cannot silently carry old traffic).

    The counters cannot be read from inside this process; tools/profile_bench.sh collects
    FETCH_SIZE and WRITE_SIZE in separate --pmc runs of this script and writes a summary under
    profiles/.  FETCH_SIZE is in KiB and counts gfx950's 128-byte requests as 64 bytes, so the
    read side is doubled; WRITE_SIZE (KiB) is exact.  Returns ({phase: bytes}, file name), or
    ({}, None) when no summary for `n` elements per GPU is present.
    """
    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, "profiles", "*_bench_summary.json")),
                       reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("elements_per_gpu") != n or d.get("library_sha256_16") != lib_hash:
            continue
        out = {}
        for phase, kern in (("hist", "hf_k_hist_pages"), ("encode", "hf_k_encode_pages"),
                            ("decode", "hf_k_decode_pages")):
            k = d.get(kern, {})
            if "FETCH_SIZE_KiB_per_launch" in k and "WRITE_SIZE_KiB_per_launch" in k:
                out[phase] = (2.0 * k["FETCH_SIZE_KiB_per_launch"]
                              + k["WRITE_SIZE_KiB_per_launch"]) * 1024.0
        return out, os.path.join("profiles", os.path.basename(path))
    return {}, None


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n_ranks):
    """`bench.py --gpus N` run directly: start N ranks of this script as child processes and pass
    their output through.  Nothing in this process has touched a GPU (torch is not even imported
    yet), the children are fresh interpreters, and this process only waits for them."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--elems", type=int, default=LLAMA3_8B_PARAMS,
                    help="bf16 elements per GPU (default: Llama-3-8B parameter count)")
    ap.add_argument("--cpu-sample", type=int, default=1536 * 1024 * 1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", type=int, default=2, choices=(1, 2, 3, 5),
                    help="workload: 2 = the headline bf16 weight stream (default); 1 = one 4 KB "
                         "block of uniform bytes; 3 = fp8 KV-cache pages with per-page codebooks; "
                         "5 = the codebook kernels on degenerate count shapes (1, 3, 5: one GPU)")
    ap.add_argument("--pages", type=int, default=1 << 18, help="config 3: 16 KiB pages")
    args = ap.parse_args()

    if args.config != 2:
        if args.gpus != 1:
            raise SystemExit("configs 1, 3 and 5 are single-GPU workloads")
        import bench_configs
        return {1: bench_configs.config1, 3: bench_configs.config3,
                5: bench_configs.config5}[args.config](args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    # The one JSON line must be alone on stdout, and libraries loaded below write there too (RCCL
    # prints a five-line version block at its first communicator): file descriptor 1 points at
    # stderr for the rest of the run, and the line goes out through a copy of the original.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from huffman_amd import api as hf

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # HF_BENCH_BACKEND=gloo is a rehearsal of the N-rank flow on a box with fewer GPUs than ranks:
    # ranks share the GPUs there are and the 256 counts are summed through host memory.  RCCL
    # itself wants one GPU per rank, which is what the default (nccl) asks for.
    backend = os.environ.get("HF_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU: no HIP device is visible")
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"--gpus {args.gpus} needs {args.gpus} GPUs, this box shows {ndev}")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # HF_BENCH_FORCE_DIST=1 takes the multi-rank code path (process group, all-reduce, barriers)
    # with a single rank, to rehearse it on a one-GPU box
    use_dist = world > 1 or os.environ.get("HF_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        kw = dict(rank=rank, world_size=world)
        if backend == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)

    def allreduce_counts(freq):
        if backend == "nccl":
            dist.all_reduce(freq)  # RCCL over xGMI, 2 KiB
        else:
            host = freq.cpu()
            dist.all_reduce(host)
            freq.copy_(host)

    def barrier():
        if backend == "nccl":
            dist.barrier(device_ids=[dev_index])
        else:
            dist.barrier()

    n = args.elems
    w = make_weights(torch, n, device)
    ctx = hf.Context(dev_index)
    system = hf.EntropyAwareHuffmanSystem(ctx=ctx)
    decoder = hf.ParrallelHuffmanDecoder(ctx=ctx)
    stat = hf.SymbolStat(ctx=ctx)

    nch = (n + hf.HF_CHUNK_SYMS - 1) // hf.HF_CHUNK_SYMS
    ctx.reserve(n)
    freq = torch.zeros(256, dtype=torch.int64, device=device)
    freq_pages = torch.empty(nch * 256, dtype=torch.int32, device=device)
    cb = hf.DeviceCodebook(ctx)
    side = torch.empty(n, dtype=torch.uint8, device=device)
    payload = torch.empty(n // 8 + nch * 192 + 64, dtype=torch.int32, device=device)  # 4 bits/sym
    chunk_off = torch.empty(((nch + 4) // 4) * 4, dtype=torch.int32, device=device)
    out = torch.empty(n, dtype=torch.bfloat16, device=device)

    def step(ev=None):
        if ev: ev[0].record()
        freq.zero_()
        system.histogram_pages(w, freq_chunks=freq_pages, freq256=freq)
        if ev: ev[1].record()
        if use_dist:
            allreduce_counts(freq)  # global codebook: identical tables on every rank
        system.codebook(freq, hf.HF_MODE_HUFFMAN, out=cb)
        if ev: ev[2].record()
        s = system.encode_with(w, cb, payload=payload, side=side, chunk_off=chunk_off, sync=False,
                               paged=True, chunk_freq=freq_pages)
        if ev: ev[3].record()
        decoder.decode(s, out=out, check_stream=False)
        if ev: ev[4].record()
        return s

    for _ in range(args.warmup):
        s = step()
    torch.cuda.synchronize()
    if use_dist:
        barrier()
    torch.cuda.synchronize()

    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        s = step(evs[k])
    torch.cuda.synchronize()
    if use_dist:
        barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel device times (ms), averaged over the timed steps
    phases = ["hist", "codebook", "encode", "decode"]
    per_step = {p: [evs[k][i].elapsed_time(evs[k][i + 1]) for k in range(args.steps)]
                for i, p in enumerate(phases)}
    ms = {p: sum(v) / len(v) for p, v in per_step.items()}
    ms_median = {p: sorted(v)[len(v) // 2] for p, v in per_step.items()}

    # ---- outside the timed region: verify and measure the stream -----------------------------
    ctx.sync()
    total_words = int(chunk_off[nch].item()) & 0xFFFFFFFF
    ok = bool(torch.equal(out.view(torch.int16), w.view(torch.int16)))
    code, lens, mode, ent = cb.download()
    f = freq.cpu().numpy().astype("float64")
    bits_ideal = float((f * lens).sum())
    nsym_codebook = float(f.sum())  # n * world when the histogram was all-reduced
    lbar = bits_ideal / max(1.0, nsym_codebook)
    comp_bytes = 4 * total_words + n + 4 * (nch + 1) + 256 * 3
    stream_ok = True
    # the first pages of the GPU's stream, kept for the CPU leg to compare with its own (the
    # buffers are reused below)
    sample_pages = min(nch, 64)
    sample_off = chunk_off[: sample_pages + 1].cpu()
    sample_words = payload[: int(sample_off[sample_pages].item()) & 0xFFFFFFFF].cpu()

    # the packed stream of format version 1 on the same tensor and codebook, outside the timed
    # region: histogram in one piece, look-back encoder, one wavefront per chunk in the decoder
    dense = None
    if rank == 0:
        dev_ms = []
        for _ in range(3):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
            freq.zero_()
            stat.accumulate(w, freq)
            e[1].record()
            sd = system.encode_with(w, cb, payload=payload, side=side, chunk_off=chunk_off, sync=False)
            e[2].record()
            decoder.decode(sd, out=out, check_stream=False)
            e[3].record()
            torch.cuda.synchronize()
            dev_ms.append([e[i].elapsed_time(e[i + 1]) for i in range(3)])
        ctx.sync()
        med = [sorted(x[i] for x in dev_ms)[1] for i in range(3)]
        dwords = int(chunk_off[nch].item()) & 0xFFFFFFFF
        dense = {"hist_ms": med[0], "encode_ms": med[1], "decode_ms": med[2],
                 "bits_per_exponent_stream": 32.0 * dwords / n,
                 "compression_ratio": 2.0 * n / (4 * dwords + n + 4 * (nch + 1) + 256 * 3),
                 "roundtrip_bit_exact": bool(torch.equal(out.view(torch.int16), w.view(torch.int16)))}
    # TreeQualityAnalyzer (tree_q_analyser.scala:21-47) on this codebook.  Its weighted sum is a
    # 32-bit wire, so the counts go in shifted right until sum(count x length) fits.
    fi = freq.cpu().numpy().astype("uint64")
    qshift = 0
    while int(((fi >> qshift) * lens.astype("uint64")).sum()) >= (1 << 32):
        qshift += 1
    try:
        q_avg, q_eff, q_ratio = hf.TreeQualityAnalyzer(256, 32, ctx=ctx)(fi >> qshift, lens)
        quality = {"avg_length_q8": q_avg, "efficiency_q8": q_eff, "compression_ratio_q8": q_ratio,
                   "counts_shifted_right_by": qshift}
    except hf.HuffhipError:
        quality = None
    ratio = 2.0 * n / comp_bytes

    raw_bytes = 2.0 * n
    # algorithmic bytes per launch: elements, side bytes, the codes at their ideal size (the page
    # headers, region lengths and per-page counts the kernels also move are not counted)
    alg = {
        "hist": 2.0 * n,
        "encode": n * (2.0 + 1.0 + lbar / 8.0) + 4.0 * nch,
        "decode": n * (1.0 + lbar / 8.0 + 2.0) + 4.0 * nch,
    }
    roof = {}
    lib_hash = library_hash()
    pmc, pmc_src = pmc_traffic(n, lib_hash)
    for p in ("hist", "encode", "decode"):
        ach = alg[p] / (ms[p] * 1e-3) / 1e9
        roof[p] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                   "frac": ach / HBM_PEAK_GBPS, "traffic": pmc.get(p), "ms": ms[p],
                   "ms_median": ms_median[p],
                   "algorithmic_bytes": alg[p], "traffic_source": pmc_src if p in pmc else None}
    dominant = "encode" if ms["encode"] >= ms["decode"] else "decode"

    if rank == 0:
        line = {
            "metric": METRIC,
            "value": raw_bytes * world * args.steps / elapsed / 1e9,
            "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": ("synthetic Gaussian bf16 weights N(0,0.02^2), "
                             f"{n} elements per GPU ({raw_bytes / 1e9:.2f} GB), exponent-byte "
                             "Huffman, forced Huffman mode, Lmax=16, one codebook, paged stream "
                             "of 16384-symbol pages without records"
                             + (", global codebook via RCCL all-reduce of the histogram"
                                if world > 1 and backend == "nccl" else "")
                             + (", REHEARSAL: ranks share GPUs, counts summed over gloo"
                                if backend != "nccl" else "")),
                "elements_per_gpu": n, "chunk_symbols": hf.HF_CHUNK_SYMS,
            },
            "encode_GBps": raw_bytes / ((ms["hist"] + ms["codebook"] + ms["encode"]) * 1e-3) / 1e9,
            "encode_kernel_GBps": raw_bytes / (ms["encode"] * 1e-3) / 1e9,
            "decode_GBps": raw_bytes / (ms["decode"] * 1e-3) / 1e9,
            "hist_GBps": raw_bytes / (ms["hist"] * 1e-3) / 1e9,
            "codebook_ms": ms["codebook"],
            "ms_per_phase_mean": ms, "ms_per_phase_median": ms_median,
            "compression_ratio": ratio,
            "bits_per_exponent_ideal": lbar,
            "bits_per_exponent_stream": 32.0 * total_words / n,
            "compression_mode": mode, "entropy_q16": ent,
            "tree_quality": quality,
            "roundtrip_bit_exact": ok,
            "dense_v1": dense,
            "library_sha256_16": lib_hash,
            "roofline": dict(roof[dominant], kernel=dominant),
            "roofline_all": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(torch, w, args.cpu_sample, code, lens, sample_words,
                                                sample_off)
            stream_ok = line["cpu_baseline"]["gpu_pages_equal_port"]
        print(json.dumps(line), file=json_out, flush=True)
    if use_dist:
        barrier()
        dist.destroy_process_group()
    if not ok or not stream_ok:
        raise SystemExit("round trip mismatch" if not ok else "pages differ from the CPU port's")


if __name__ == "__main__":
    main()
