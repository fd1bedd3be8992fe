"""CPU suite: the multi-rank host logic (contiguous chunk-aligned shards, histogram all-reduce
for the global codebook) under gloo with world_size 2.  The per-rank histograms come from the
oracle here; on GPUs they come from the HIP kernel and the same all-reduce runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_properties():
    from huffman_amd.dist import shard_range
    from huffman_amd import HF_CHUNK_SYMS as CH
    for n in (0, 1, CH - 1, CH, CH + 1, 7 * CH + 5, 1000 * CH, 8_030_261_248):
        for world in (1, 2, 3, 8):
            prev = 0
            sizes = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert lo == prev and lo <= hi <= n
                assert lo % CH == 0 or lo == n
                sizes.append(hi - lo)
                prev = hi
            assert prev == n
            full = [s for s in sizes if s]
            if len(full) > 1:
                assert max(full[:-1]) - min(full[:-1]) <= CH
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_assign_tensors_is_balanced_and_deterministic():
    from huffman_amd.dist import assign_tensors
    rng = np.random.default_rng(3)
    sizes = [int(x) for x in rng.integers(1, 1 << 20, 200)] + [1 << 26]
    for world in (1, 2, 8):
        place = assign_tensors(sizes, world, split_above=1 << 24)
        assert place == assign_tensors(list(sizes), world, split_above=1 << 24)
        assert place[-1] == (-1 if world >= 1 else 0)          # the oversized one is sharded
        assert all(0 <= p < world for p in place[:-1])
        load = [sum(s for s, p in zip(sizes, place) if p == r) for r in range(world)]
        assert max(load) - min(load) <= max(sizes[:-1])        # greedy bound: one tensor apart
    assert assign_tensors([5, 5, 5], 2) == [0, 1, 0]           # ties go to the lower rank
    with pytest.raises(ValueError):
        assign_tensors([1], 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, seed, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from huffman_amd.dist import allreduce_histogram, shard_directory, shard_range
        from oracle import oracle as orc
        from conftest import gaussian_bf16
        w = gaussian_bf16(n, seed)                       # every rank builds the same tensor
        sym, _ = orc.symbolize(w, orc.LAYOUT_BF16_EXP)
        lo, hi = shard_range(n, rank, world)
        local = orc.aw_symbol_stat(sym[lo:hi], 256, 64)
        freq = torch.from_numpy(local.astype(np.int64))
        allreduce_histogram(freq)
        glob = freq.numpy().astype(np.uint64)
        code, lens = orc.rp_system_codebook(glob, 64, 0, 16)
        # shard streams are independent: encode this rank's range with the global table
        payload, off = orc.rp_encode(sym[lo:hi], code, lens)
        back = orc.rp_decode(payload, off, hi - lo, code, lens)
        stored = orc.container_bytes(hi - lo, orc.LAYOUT_BF16_EXP, 0, int(payload.size))
        directory = shard_directory(lo, hi, int(payload.size), stored)
        q.put((rank, lo, hi, glob.tolist(), code.tolist(), lens.tolist(), int(payload.size),
               bool(np.array_equal(back, sym[lo:hi])), directory.tolist()))
    finally:
        dist.destroy_process_group()


def test_global_codebook_two_ranks_gloo(orc):
    world, n, seed = 2, 5 * 16384 + 1234, 21
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from conftest import gaussian_bf16
    sym, _ = orc.symbolize(gaussian_bf16(n, seed), orc.LAYOUT_BF16_EXP)
    whole = orc.aw_symbol_stat(sym, 256, 64)
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n
    assert res[0][1] % 16384 == 0 and res[1][1] % 16384 == 0
    for r in res:
        assert r[3] == whole.tolist()                  # all-reduced histogram = global histogram
        assert r[7]                                     # the shard round-trips
    assert res[0][4] == res[1][4] and res[0][5] == res[1][5]  # identical codebooks, no broadcast
    # every rank holds the same shard directory: ranges, payload words, stored bytes per shard
    assert res[0][8] == res[1][8] and len(res[0][8]) == world
    for r in res:
        assert res[0][8][r[0]][:3] == [r[1], r[2], r[6]] and res[0][8][r[0]][3] > 4 * r[6]
    code, lens = orc.rp_system_codebook(whole, 64, 0, 16)
    assert res[0][4] == code.tolist() and res[0][5] == lens.tolist()
