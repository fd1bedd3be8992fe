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


def allreduce_histogram(freq256: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the per-rank symbol counts in place (int64[256]).  With the nccl backend this is the
    RCCL all-reduce over xGMI; with gloo (CPU tests) the same call on host tensors."""
    if freq256.dtype != torch.int64 or freq256.numel() != 256:
        raise ValueError("histogram must be int64[256]")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(freq256, op=dist.ReduceOp.SUM, group=group)
    return freq256


class ShardedCodec:
    """Encode / decode this rank's shard of a tensor that is partitioned over the ranks."""

    def __init__(self, ctx=None, global_codebook: bool = True, group=None):
        from . import api
        self._api = api
        self.ctx = ctx if ctx is not None else api.default_context()
        self.system = api.EntropyAwareHuffmanSystem(ctx=self.ctx)
        self.decoder = api.ParrallelHuffmanDecoder(ctx=self.ctx)
        self.global_codebook = global_codebook
        self.group = group

    def encode(self, shard: torch.Tensor, mode: int = HF_MODE_AUTO, layout: Optional[int] = None):
        shard = shard.contiguous()
        freq = self.system.histogram(shard, layout)
        if self.global_codebook:
            allreduce_histogram(freq, self.group)
        cb = self.system.codebook(freq, mode)
        return self.system.encode_with(shard, cb, layout)

    def decode(self, stream, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.decoder.decode(stream, out=out)
