"""Multi-GPU use of the codec: one process per GPU, contiguous chunk-aligned shards.

The path shards without any data exchange: every rank histograms, encodes and decodes its own
range.  The only exchange step exists in *global-codebook* mode, where the 256 symbol counts
are summed over all ranks (one RCCL all-reduce of 2 KiB over xGMI, latency bound) so that the
deterministic integer codebook pipeline yields identical tables everywhere.  Bulk data never
crosses a link.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

from ._lib import HF_CHUNK_SYMS, HF_MODE_AUTO


def shard_range(n: int, rank: int, world: int, align: int = HF_CHUNK_SYMS) -> Tuple[int, int]:
    """Element range [lo, hi) of `rank`: contiguous, chunk aligned except for the global tail,
    sizes differing by at most one chunk, empty ranges only when there are fewer chunks than
    ranks."""
    if world < 1 or not (0 <= rank < world) or n < 0:
        raise ValueError("bad shard request")
    nchunks = (n + align - 1) // align
    base, extra = divmod(nchunks, world)
    c_lo = rank * base + min(rank, extra)
    c_hi = c_lo + base + (1 if rank < extra else 0)
    return min(n, c_lo * align), min(n, c_hi * align)


def assign_tensors(sizes, world: int, split_above: Optional[int] = None):
    """Place the tensors of a collection on `world` ranks: a tensor goes whole to one rank, the
    largest first to the rank with the least bytes so far (ties to the lower rank), except that
    a tensor of more than `split_above` elements is partitioned over all ranks with
    `shard_range`.  Returns one entry per tensor: the rank, or -1 for "sharded over all ranks".
    Deterministic, so every rank computes the same placement without talking."""
    if world < 1:
        raise ValueError("bad world size")
    load = [0] * world
    place = [0] * len(sizes)
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    for i in order:
        n = int(sizes[i])
        if split_above is not None and n > split_above:
            place[i] = -1
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                load[r] += hi - lo
            continue
        r = min(range(world), key=lambda k: (load[k], k))
        place[i] = r
        load[r] += n
    return place


def allreduce_histogram(freq256: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the per-rank symbol counts in place (int64[256]).  With the nccl backend this is the
    RCCL all-reduce over xGMI; with gloo (CPU tests) the same call on host tensors."""
    if freq256.dtype != torch.int64 or freq256.numel() != 256:
        raise ValueError("histogram must be int64[256]")
    # (run whenever a process group exists, a group of one rank included: the collective is the
    # same call at every world size, and a one-rank run then rehearses the very path N ranks take)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(freq256, op=dist.ReduceOp.SUM, group=group)
    return freq256


def shard_directory(lo: int, hi: int, payload_words: int, stored_bytes: int,
                    device: Optional[torch.device] = None, group=None) -> torch.Tensor:
    """The shard directory of a partitioned tensor: int64[world][4], row r = (first element, end
    element, payload words, stored bytes) of rank r's stream.  The streams are independent, so
    this is all a reader needs to find and decode any shard; it costs one all-gather of 32 bytes
    per rank (RCCL with the nccl backend, which wants `device` to be this rank's GPU)."""
    mine = torch.tensor([lo, hi, payload_words, stored_bytes], dtype=torch.int64,
                        device=device if device is not None else "cpu")
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return mine.view(1, 4)
    world = dist.get_world_size(group)
    out = torch.empty(world * 4, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.view(world, 4)


class ShardedCodec:
    """Encode / decode this rank's shard of a tensor that is partitioned over the ranks."""

    def __init__(self, ctx=None, global_codebook: bool = True, group=None, paged: bool = False):
        """``paged``: store the shard as pages without records (``hf_encode_pages_shared``): per-page
        and shard counts from one read of the data, the shard's counts summed over the ranks, every
        page sized exactly from its own counts under the global codebook."""
        from . import api
        self._api = api
        self.ctx = ctx if ctx is not None else api.default_context()
        self.system = api.EntropyAwareHuffmanSystem(ctx=self.ctx)
        self.decoder = api.ParrallelHuffmanDecoder(ctx=self.ctx)
        self.global_codebook = global_codebook
        self.group = group
        self.paged = paged

    def encode(self, shard: torch.Tensor, mode: int = HF_MODE_AUTO, layout: Optional[int] = None):
        shard = shard.contiguous()
        if self.paged:
            pages, freq = self.system.histogram_pages(shard, layout)
        else:
            pages, freq = None, self.system.histogram(shard, layout)
        if self.global_codebook:
            allreduce_histogram(freq, self.group)
        cb = self.system.codebook(freq, mode)
        return self.system.encode_with(shard, cb, layout, paged=self.paged, chunk_freq=pages)

    def decode(self, stream, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.decoder.decode(stream, out=out)

    def directory(self, stream, lo: int) -> torch.Tensor:
        """Shard directory (see `shard_directory`) for this rank's `stream` starting at element
        `lo` of the whole tensor; stored bytes are those of the stream's container."""
        return shard_directory(lo, lo + stream.n, stream.total_words, stream.container_bytes(),
                               device=stream.payload.device, group=self.group)
