"""GPU suite: the multi-rank path on the hardware there is.

One MI355X is what a test box has, so the pieces of the N-rank flow are each run for real in the
form one GPU allows: the RCCL all-reduce of the histogram inside a one-rank `nccl` process group
(the same collective call at every world size), the HIP histogram -> all-reduce -> HIP codebook
chain with two processes sharing the GPU and summing over gloo, one shard of the 8-way split of
the Llama-3-70B-sized stream at full size, and `bench.py --gpus 2` started the way the driver
starts it (the script launches its own ranks)."""
import ctypes as C
import json
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import gaussian_bf16

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLAMA3_70B_PARAMS = 70_553_706_496


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def kraft(lens):
    l = np.asarray(lens, dtype=np.int64)
    return float(np.sum(2.0 ** (-l[l > 0].astype(np.float64))))


@pytest.fixture()
def nccl_group():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        yield dist.group.WORLD
    finally:
        dist.destroy_process_group()


def test_sharded_codec_runs_the_rccl_all_reduce(hf, orc, nccl_group, monkeypatch):
    # ShardedCodec(global_codebook=True) under an nccl group: the histogram all-reduce is really
    # issued (counted here), the codebook is the oracle's for the same counts, the shard
    # round-trips and the shard directory comes out of the all-gather
    from huffman_amd import dist as hd
    from huffman_amd.dist import ShardedCodec, shard_range
    assert dist.get_backend() == "nccl"
    calls = []
    real = dist.all_reduce

    def counted(t, *a, **k):
        calls.append((t.device.type, tuple(t.shape), t.dtype))
        return real(t, *a, **k)

    monkeypatch.setattr(hd.dist, "all_reduce", counted)
    n = 9 * 16384 + 321
    w = gaussian_bf16(n, seed=77)
    lo, hi = shard_range(n, 0, 1)
    assert (lo, hi) == (0, n)
    t = torch.from_numpy(w.view(np.int16)).to(DEV).view(torch.bfloat16)
    codec = ShardedCodec(global_codebook=True, group=nccl_group)
    s = codec.encode(t[lo:hi], mode=hf.HF_MODE_HUFFMAN)
    assert calls == [("cuda", (256,), torch.int64)]
    sym, side = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
    freq = orc.aw_symbol_stat(sym, 256, 64)
    code, lens, mode, _ = s.codebook.download()
    ec, el = orc.rp_system_codebook(freq, 64, 0, 16)
    assert mode == 1 and np.array_equal(code, ec) and np.array_equal(lens, el)
    payload, off = orc.rp_encode(sym, ec, el)
    assert s.total_words == payload.size
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    out = codec.decode(s)
    assert torch.equal(out.view(torch.int16), t.view(torch.int16))
    d = codec.directory(s, lo)
    assert d.is_cuda and d.tolist() == [[lo, hi, s.total_words, s.container_bytes()]]


def test_sharded_codec_paged_shard(hf, orc, nccl_group):
    # the same flow with the shard stored as pages without records: one read of the data for the
    # per-page and the shard's counts, all-reduce, global codebook, exact page sizes
    from huffman_amd.dist import ShardedCodec
    n = 7 * 16384 + 999
    w = gaussian_bf16(n, seed=78)
    t = torch.from_numpy(w.view(np.int16)).to(DEV).view(torch.bfloat16)
    codec = ShardedCodec(global_codebook=True, group=nccl_group, paged=True)
    s = codec.encode(t, mode=hf.HF_MODE_HUFFMAN)
    sym, _ = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
    ec, el = orc.rp_system_codebook(orc.aw_symbol_stat(sym, 256, 64), 64, 0, 16)
    payload, off = orc.rp_encode_pages_shared(sym, ec, el)
    assert s.paged and s.total_words == payload.size
    assert np.array_equal(s.payload[: payload.size].cpu().numpy().view(np.uint32), payload)
    assert np.array_equal(s.chunk_off[: off.size].cpu().numpy().view(np.uint32), off)
    assert torch.equal(codec.decode(s).view(torch.int16), t.view(torch.int16))
    assert codec.directory(s, 0).tolist() == [[0, n, s.total_words, s.container_bytes()]]


def test_all_reduce_over_rccl_sums_device_counts(hf, nccl_group):
    # the collective itself on the HIP kernel's own output buffer: one rank, so the sum is the
    # input, and the call must leave the counters usable by the codebook kernel that follows on
    # the context's stream
    from huffman_amd.dist import allreduce_histogram
    sys_ = hf.EntropyAwareHuffmanSystem()
    t = torch.randint(0, 256, (3 * 16384 + 5,), dtype=torch.uint8, device=DEV)
    freq = sys_.histogram(t)
    before = freq.clone()
    allreduce_histogram(freq)
    torch.cuda.synchronize()
    assert torch.equal(freq, before) and int(freq.sum()) == t.numel()
    cb = sys_.codebook(freq, hf.HF_MODE_HUFFMAN)
    s = sys_.encode_with(t, cb)
    assert torch.equal(hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s), t)


def test_c_abi_all_reduce_on_a_one_rank_communicator(hf, orc):
    # hf_symbol_stat_allreduce through the C ABI with an ncclComm_t made by ncclCommInitAll(1):
    # histogram kernel -> all-reduce -> codebook kernel on the context's stream, no torch
    # collective anywhere; the tables must be the oracle's for the same counts
    L = hf.L
    lib = L.lib()
    rccl = C.CDLL("librccl.so.1")
    rccl.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    comm = C.c_void_p()
    dev0 = C.c_int(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, C.byref(dev0)) == 0 and comm.value
    try:
        ctx = hf.Context(0)
        n = 5 * 16384 + 77
        w = gaussian_bf16(n, seed=5)
        t = torch.from_numpy(w.view(np.int16)).to(DEV)
        freq = torch.zeros(256, dtype=torch.int64, device=DEV)
        vp = lambda x: C.c_void_p(x.data_ptr())
        assert lib.hf_symbol_stat_dev(ctx.handle, vp(t), n, L.HF_LAYOUT_BF16_EXP, vp(freq)) == 0
        assert lib.hf_symbol_stat_allreduce(ctx.handle, comm, vp(freq)) == 0
        cb = torch.zeros(L.HF_CODEBOOK_DEV_BYTES, dtype=torch.uint8, device=DEV)
        assert lib.hf_system_codebook_dev(ctx.handle, vp(freq), L.HF_MODE_HUFFMAN, 0, vp(cb)) == 0
        ctx.sync()
        sym, _ = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
        exp = orc.aw_symbol_stat(sym, 256, 64)
        assert np.array_equal(freq.cpu().numpy().astype(np.uint64), np.asarray(exp, np.uint64))
        code, lens, mode, _ = hf.DeviceCodebook(ctx, cb).download()
        ec, el = orc.rp_system_codebook(exp, 64, 0, 16)
        assert mode == 1 and np.array_equal(code, ec) and np.array_equal(lens, el)
        # argument checks come before anything touches RCCL
        assert lib.hf_symbol_stat_allreduce(ctx.handle, None, vp(freq)) == L.HF_ERR_ARG
        assert lib.hf_symbol_stat_allreduce(ctx.handle, comm, None) == L.HF_ERR_ARG
    finally:
        torch.cuda.synchronize()
        rccl.ncclCommDestroy(comm)


def test_plain_c_shard_driver_on_one_gpu():
    # the single-process driver (ncclCommInitAll, one context per GPU, plain C over the C ABI)
    # with the one GPU a test box has: shard = whole tensor, all-reduce over one rank
    exe = os.path.join(ROOT, "huffman_amd", "lib", "hf_shard_driver")
    assert os.path.exists(exe), "build it with make -C huffman_amd/csrc"
    n = 40 * 16384 + 123
    r = subprocess.run([exe, "--gpus", "1", "--elems", str(n)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"rank 0: elements \[0, (\d+)\) local_count_sum (\d+) global_count_sum (\d+) "
                  r"codebook_hash ([0-9a-f]{16}) mode 1 .* round_trip ok", r.stdout)
    assert m and [int(m.group(i)) for i in (1, 2, 3)] == [n, n, n], r.stdout
    assert "same_codebook_everywhere yes all_ok yes" in r.stdout


def test_config4_shard_full_size_properties(hf):
    # one of the 8 contiguous chunk-aligned shards of the Llama-3-70B-sized stream
    # (70 553 706 496 bf16 values): 8 819 228 672 elements, 17.64 GB, on one GPU -- histogram,
    # codebook, encode, decode; the size-independent properties of the stream and an exact round trip
    import bench
    from huffman_amd.dist import shard_range
    if torch.cuda.get_device_properties(0).total_memory < 100e9:
        pytest.skip("needs ~55 GB of device memory")
    sizes = [shard_range(LLAMA3_70B_PARAMS, r, 8) for r in range(8)]
    assert sizes[0][0] == 0 and sizes[-1][1] == LLAMA3_70B_PARAMS
    assert all(a[1] == b[0] and b[0] % 16384 == 0 for a, b in zip(sizes, sizes[1:]))
    lo, hi = sizes[0]  # the largest shard: 538 283 chunks (the others hold one chunk less)
    n = hi - lo
    assert n == 538_283 * 16384 and abs(n - LLAMA3_70B_PARAMS // 8) < 16384
    w = bench.make_weights(torch, n, DEV)
    sys_ = hf.EntropyAwareHuffmanSystem()
    freq = sys_.histogram(w)
    assert int(freq.sum()) == n
    cb = sys_.codebook(freq, hf.HF_MODE_HUFFMAN)
    s = sys_.encode_with(w, cb)
    nch = (n + 16383) // 16384
    off = s.chunk_off[: nch + 1].cpu().numpy().view(np.uint32).astype(np.int64)
    assert off[0] == 0 and off[-1] == s.total_words and np.all(np.diff(off) > 0)
    code, lens, mode, _ = cb.download()
    assert mode == 1 and lens.max() <= 16 and abs(kraft(lens) - 1.0) < 1e-12
    ideal_bits = float((freq.cpu().numpy().astype(np.float64) * lens).sum())
    assert ideal_bits <= 32.0 * s.total_words <= ideal_bits + nch * 64 * 96
    out = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s)
    assert torch.equal(out.view(torch.int16), w.view(torch.int16))
    assert 1.45 < s.ratio() < 1.52


def _two_rank_worker(rank, world, port, n, seed, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import huffman_amd.api as hf
        from huffman_amd.dist import shard_range
        from conftest import gaussian_bf16 as gen
        torch.cuda.set_device(0)
        w = gen(n, seed)
        lo, hi = shard_range(n, rank, world)
        t = torch.from_numpy(w[lo:hi].view(np.int16)).to("cuda:0").view(torch.bfloat16)
        sys_ = hf.EntropyAwareHuffmanSystem()
        freq = sys_.histogram(t)                 # HIP histogram of this rank's shard
        local = freq.cpu()
        host = local.clone()
        dist.all_reduce(host)                    # two ranks, one GPU: summed through host memory
        freq.copy_(host)
        cb = sys_.codebook(freq, hf.HF_MODE_HUFFMAN)   # HIP codebook of the summed counts
        s = sys_.encode_with(t, cb)
        code, lens, mode, _ = cb.download()
        back = hf.ParrallelHuffmanDecoder(ctx=sys_.ctx).decode(s)
        ok = bool(torch.equal(back.view(torch.int16), t.view(torch.int16)))
        q.put((rank, lo, hi, local.tolist(), host.tolist(), code.tolist(), lens.tolist(), mode,
               int(s.total_words), ok))
    finally:
        dist.destroy_process_group()


def test_hip_histograms_summed_over_two_ranks_give_the_oracles_global_codebook(orc):
    # two processes share the GPU; each histograms its own shard with the HIP kernel, the two
    # DIFFERENT count vectors are summed with a real two-rank all-reduce (gloo: RCCL wants one GPU
    # per rank), and each rank's HIP codebook kernel must give the oracle's table for the sum
    world, n, seed = 2, 11 * 16384 + 4321, 31
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, n, seed, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    w = gaussian_bf16(n, seed)
    sym, _ = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
    whole = orc.aw_symbol_stat(sym, 256, 64)
    ec, el = orc.rp_system_codebook(whole, 64, 0, 16)
    assert res[0][3] != res[1][3]                          # the per-rank counts differ
    for r in res:
        lo, hi = r[1], r[2]
        assert r[3] == orc.aw_symbol_stat(sym[lo:hi], 256, 64).tolist()
        assert r[4] == whole.tolist()
        assert r[5] == ec.tolist() and r[6] == el.tolist() and r[7] == 1
        payload, _ = orc.rp_encode(sym[lo:hi], ec, el)
        assert r[8] == payload.size and r[9]


def _run_bench(extra_env, args, timeout=600):
    env = dict(os.environ)
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    # `python bench.py --gpus 2` exactly as the driver calls it (no launcher around it): the script
    # starts two ranks itself and rank 0 prints the one line.  With a single GPU in the box the
    # ranks share it and sum their counts over gloo (HF_BENCH_BACKEND), everything else is the path
    # eight ranks on eight GPUs take.
    ndev = torch.cuda.device_count()
    env = {} if ndev >= 2 else {"HF_BENCH_BACKEND": "gloo"}
    line = _run_bench(env, ["--gpus", "2", "--steps", "2", "--warmup", "1",
                            "--elems", str(64 * 16384 * 16 + 77), "--no-cpu-baseline"])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["roundtrip_bit_exact"] is True
    assert line["scaling"] == "weak" and line["value"] > 0
    assert "roofline" in line and line["roofline"]["bound"] == "hbm"


def test_bench_one_rank_inside_an_rccl_group():
    # the N-rank code path (process group, RCCL all-reduce of the counts, barriers) with one rank
    line = _run_bench({"HF_BENCH_FORCE_DIST": "1", "MASTER_PORT": str(_free_port())},
                      ["--gpus", "1", "--steps", "2", "--warmup", "1",
                       "--elems", str(64 * 16384 * 16), "--no-cpu-baseline"])
    assert line["n_gpus"] == 1 and line["roundtrip_bit_exact"] is True
