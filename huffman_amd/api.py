"""Host-side mirror of the reference's module boundaries over the C ABI.

Class names, constructor parameters and port names follow the Chisel modules
(src/main/scala/huffman/encoder/*.scala, decoder/top_decoder.scala); each class is a thin
wrapper around one entry point of libhuffhip.so.  PyTorch is used only for device memory and
streams.  Nothing here computes on the CPU and nothing falls back to it: every call goes
through the HIP library or raises.
"""
from __future__ import annotations

import ctypes as C
import functools
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib as L
from ._lib import (HF_CHUNK_SYMS, HF_CODEBOOK_DEV_BYTES, HF_LAYOUT_BF16_EXP, HF_LAYOUT_FP8,
                   HF_LAYOUT_BITPLANE8, HF_LAYOUT_FP4_E2M1_EXP, HF_LAYOUT_FP8_E4M3_EXP, HF_LAYOUT_FP16_EXP, HF_LAYOUT_U8, HF_MODE_AUTO, HF_MODE_BEST, HF_MODE_BYPASS, HF_MODE_HUFFMAN, HuffhipError,
                   check, hf_entropy_config)

_u8p, _u16p, _u32p, _u64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_uint32),
                             C.POINTER(C.c_uint64))


def _np(a, dt) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=dt))


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(t)


def layout_of(t: torch.Tensor) -> int:
    """16-bit words are split into a coded byte and a raw byte, 8-bit tensors coded byte by byte.

    The 16-bit split (symbol = bits 14..7, raw byte = bit 15 and bits 6..0) puts the whole
    exponent of a bf16 word into the symbol; of an fp16 word it takes the 5 exponent bits and the
    3 leading mantissa bits, which costs nothing in size (those bits code at about 3 bits) and is
    exactly reversible, so fp16 and 16-bit integer tensors go through the same kernels.
    """
    if t.dtype in (torch.bfloat16, torch.float16, torch.int16):
        return HF_LAYOUT_BF16_EXP
    if t.dtype in (torch.uint8, torch.int8):
        return HF_LAYOUT_U8
    if t.dtype in (getattr(torch, "float8_e4m3fn", None), getattr(torch, "float8_e5m2", None)):
        return HF_LAYOUT_FP8
    raise TypeError(f"unsupported element type {t.dtype}")


# layouts that go through hf_fields_split / hf_fields_merge around the byte-symbol kernels
FIELD_LAYOUTS = (HF_LAYOUT_FP16_EXP, HF_LAYOUT_FP8_E4M3_EXP, HF_LAYOUT_FP4_E2M1_EXP,
                 HF_LAYOUT_BITPLANE8)


def element_bytes(layout: int) -> int:
    return 2 if layout in (HF_LAYOUT_BF16_EXP, HF_LAYOUT_FP16_EXP) else 1


def side_bytes(n: int, layout: int) -> int:
    """Bytes of the side stream a stream of n elements carries."""
    if layout == HF_LAYOUT_BF16_EXP:
        return n
    if layout in FIELD_LAYOUTS:  # (0 for the bit-plane transform)
        return int(L.lib().hf_fields_side_bytes(n, layout))
    return 0


def _dev_ptr(t: Optional[torch.Tensor]) -> int:
    """Device address of a tensor the library will read or write as one flat array."""
    if t is None:
        return 0
    if not t.is_contiguous():
        raise ValueError("the library walks memory from data_ptr(): pass a contiguous tensor "
                         "(inputs: .contiguous(); outputs cannot be strided views)")
    return t.data_ptr()


def _payload_words_default(n: int, nch: int) -> int:
    """Room for 8 bits per symbol plus the per-lane look-ahead words."""
    return max(4, n // 4 + nch * 192 + 64)


class Context:
    """One hf_ctx: a device plus the stream all of its launches go to."""

    def __init__(self, device: Optional[int] = None, stream: Optional[torch.cuda.Stream] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("huffman_amd needs a GPU: no HIP device is visible")
        self.device = torch.cuda.current_device() if device is None else int(device)
        with torch.cuda.device(self.device):
            self.stream = stream if stream is not None else torch.cuda.current_stream()
        h = C.c_void_p()
        check(L.lib().hf_ctx_create(C.byref(h), self.device, C.c_void_p(self.stream.cuda_stream)),
              "hf_ctx_create")
        self._h = h

    @property
    def handle(self):
        return self._h

    def flush(self) -> None:
        check(L.lib().hf_flush(self._h), "hf_flush")

    def sync(self) -> None:
        check(L.lib().hf_sync(self._h), "hf_sync")

    def reserve(self, max_n: int, flags: int = L.HF_RESERVE_ALL) -> None:
        """Make the context's workspaces now, for streams of up to max_n elements: afterwards no
        enqueue-only call with n <= max_n allocates, frees or synchronises (hf_ctx_reserve)."""
        check(L.lib().hf_ctx_reserve(self._h, int(max_n), int(flags)), "hf_ctx_reserve")

    def set_timing(self, on: bool) -> None:
        check(L.lib().hf_set_timing(self._h, 1 if on else 0), "hf_set_timing")

    def last_kernel_ms(self) -> float:
        ms = C.c_float(0)
        check(L.lib().hf_last_kernel_ms(self._h, C.byref(ms)), "hf_last_kernel_ms")
        return float(ms.value)

    def close(self) -> None:
        if getattr(self, "_h", None):
            L.lib().hf_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: dict = {}


def default_context() -> Context:
    dev = torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream().cuda_stream)
    if key not in _default_ctx:
        _default_ctx[key] = Context(dev)
    return _default_ctx[key]


class _Module:
    def __init__(self, ctx: Optional[Context]):
        self._ctx = ctx

    @property
    def ctx(self) -> Context:
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    def flush(self) -> None:
        """`flush` port: abort the block in flight and clear state."""
        self.ctx.flush()


# ---------------------------------------------------------------------------------------------
def _on_ctx_stream(method):
    """Run a method with the context's stream as torch's current stream, so that every tensor it
    allocates belongs to the stream the kernels write it on: the caching allocator then cannot hand
    a dropped temporary (or a result not yet consumed) to other current-stream work while a kernel
    enqueued on the context's stream still uses it."""
    @functools.wraps(method)
    def run(self, *a, **k):
        with torch.cuda.stream(self.ctx.stream):
            return method(self, *a, **k)
    return run


class SymbolStat(_Module):
    """symbol_stat.scala:6-58.  ``freq_out`` of one block of symbols."""

    def __init__(self, sym_width: int = 8, wtWidth: int = 32, depth: int = 256, ctx=None):
        super().__init__(ctx)
        self.sym_width, self.wtWidth, self.depth = sym_width, wtWidth, depth

    def __call__(self, data_in: torch.Tensor, layout: Optional[int] = None) -> np.ndarray:
        lay = layout_of(data_in) if layout is None else layout
        data_in = data_in.contiguous()
        freq = np.zeros(self.depth, np.uint64)
        check(L.lib().hf_symbol_stat(self.ctx.handle, C.c_void_p(_dev_ptr(data_in)),
                                     data_in.numel(), lay, self.depth, self.wtWidth,
                                     _ptr(freq, _u64p)), "hf_symbol_stat")
        return freq

    def accumulate(self, data_in: torch.Tensor, freq256: torch.Tensor,
                   layout: Optional[int] = None) -> None:
        """Device-side block: add the counts of ``data_in`` into ``freq256`` (int64[256])."""
        lay = layout_of(data_in) if layout is None else layout
        data_in = data_in.contiguous()
        assert freq256.dtype == torch.int64 and freq256.numel() == 256 and freq256.is_cuda
        check(L.lib().hf_symbol_stat_dev(self.ctx.handle, C.c_void_p(_dev_ptr(data_in)),
                                         data_in.numel(), lay, C.c_void_p(_dev_ptr(freq256))),
              "hf_symbol_stat_dev")


class SymbolSort(_Module):
    """symbol_sort.scala:82-148.  ``sorted_out`` plus the symbol identities."""

    def __init__(self, depth: int = 256, wtWidth: int = 32, ctx=None):
        super().__init__(ctx)
        self.depth, self.wtWidth = depth, wtWidth
        self.rank2sym: Optional[np.ndarray] = None

    def __call__(self, freq_in) -> np.ndarray:
        f = _np(freq_in, np.uint64)
        assert f.size == self.depth
        out = np.zeros(self.depth, np.uint64)
        ids = np.zeros(self.depth, np.uint16)
        check(L.lib().hf_symbol_sort(self.ctx.handle, _ptr(f, _u64p), self.depth,
                                     _ptr(out, _u64p), _ptr(ids, _u16p)), "hf_symbol_sort")
        self.rank2sym = ids
        return out


class ShannonEntropy(_Module):
    """shannon_entropy.scala:11-118 with adaptive_threshold.scala:11-87 inside."""

    def __init__(self, depth: int = 256, wtWidth: int = 32, ctx=None):
        super().__init__(ctx)
        self.depth, self.wtWidth = depth, wtWidth

    def __call__(self, freq_in):
        f = _np(freq_in, np.uint64)
        assert f.size == self.depth
        e = C.c_uint32(0)
        m = C.c_int(0)
        check(L.lib().hf_entropy(self.ctx.handle, _ptr(f, _u64p), self.depth, C.byref(e),
                                 C.byref(m)), "hf_entropy")
        return int(e.value), int(m.value)  # entropy_out, compression_mode


class AdaptiveThreshold:
    """adaptive_threshold.scala:11-87: the parameter registers chosen for an entropy value.

    The RTL exports only ``encoding_mode``; the other registers sit behind commented-out ports
    (:17-22).  Host arithmetic, needs no context.
    """

    def __init__(self, depth: int = 256, wtWidth: int = 32, fracBits: int = 16):
        assert fracBits == 16
        self.depth, self.wtWidth = depth, wtWidth

    def __call__(self, entropy_in: int) -> dict:
        cfg = L.hf_adaptive_config()
        check(L.lib().hf_adaptive_threshold(int(entropy_in), self.depth, self.wtWidth,
                                            C.byref(cfg)), "hf_adaptive_threshold")
        return dict(encoding_mode=int(cfg.encoding_mode),
                    compression_threshold=int(cfg.compression_threshold_q16),
                    tree_depth_limit=int(cfg.tree_depth_limit),
                    min_freq_threshold=int(cfg.min_freq_threshold))


class HuffmanTreeBuilder(_Module):
    """tree_build.scala:17-140."""

    def __init__(self, numLeaves: int = 256, wtWidth: int = 32, ctx=None):
        super().__init__(ctx)
        self.numLeaves, self.wtWidth = numLeaves, wtWidth

    def __call__(self, freqs) -> dict:
        f = _np(freqs, np.uint64)
        N = self.numLeaves
        assert f.size == N
        M = 2 * N - 1
        weights = np.zeros(M, np.uint64)
        lefts = np.zeros(M, np.uint16)
        rights = np.zeros(M, np.uint16)
        depths = np.zeros(N, np.uint16)
        root = C.c_uint16(0)
        W = 32 if self.wtWidth <= 32 else 64
        check(L.lib().hf_tree_build(self.ctx.handle, _ptr(f, _u64p), N, W, _ptr(weights, _u64p),
                                    _ptr(lefts, _u16p), _ptr(rights, _u16p), C.byref(root),
                                    _ptr(depths, _u16p)), "hf_tree_build")
        return dict(root=int(root.value), weights=weights, lefts=lefts, rights=rights,
                    depths=depths)


class tree_adjust(_Module):  # noqa: N801  (name as in tree_adjust.scala:6)
    """tree_adjust.scala:6-127."""

    def __init__(self, numLeaves: int = 256, maxDepth: int = 16, ctx=None):
        super().__init__(ctx)
        self.numLeaves, self.maxDepth = numLeaves, maxDepth

    def __call__(self, depths_in) -> np.ndarray:
        d = _np(depths_in, np.uint16)
        assert d.size == self.numLeaves
        out = np.zeros(self.numLeaves, np.uint16)
        check(L.lib().hf_tree_adjust(self.ctx.handle, _ptr(d, _u16p), self.numLeaves,
                                     self.maxDepth, _ptr(out, _u16p)), "hf_tree_adjust")
        return out


class EntropyAwareTreeBuilder(_Module):
    """entropy_aware_tree_builder.scala:6-169: ``freq_in`` -> ``code_out``, ``length_out``."""

    def __init__(self, depth: int = 256, wtWidth: int = 32, ctx=None):
        super().__init__(ctx)
        self.depth, self.wtWidth = depth, wtWidth

    def __call__(self, freq_in, encoding_mode: int = 1, tree_depth_limit: int = 0,
                 min_freq_threshold: int = 0):
        f = _np(freq_in, np.uint64)
        assert f.size == self.depth
        cfg = hf_entropy_config(encoding_mode, tree_depth_limit, min_freq_threshold)
        code = np.zeros(self.depth, np.uint16)
        lens = np.zeros(self.depth, np.uint8)
        check(L.lib().hf_codebook(self.ctx.handle, _ptr(f, _u64p), self.depth, C.byref(cfg),
                                  _ptr(code, _u16p), _ptr(lens, _u8p)), "hf_codebook")
        return code, lens


class TreeQualityAnalyzer(_Module):
    """tree_q_analyser.scala:10-47: (avg_length Q8, efficiency, compression_ratio)."""

    def __init__(self, depth: int = 256, wtWidth: int = 32, ctx=None):
        super().__init__(ctx)
        self.depth, self.wtWidth = depth, wtWidth

    def __call__(self, freq_in, code_length):
        f = _np(freq_in, np.uint64)
        l = _np(code_length, np.uint8)
        a, e, r = C.c_uint16(0), C.c_uint16(0), C.c_uint16(0)
        h = self._ctx.handle if self._ctx is not None else None
        check(L.lib().hf_quality(h, _ptr(f, _u64p), _ptr(l, _u8p), self.depth, self.wtWidth,
                                 C.byref(a), C.byref(e), C.byref(r)), "hf_quality")
        return int(a.value), int(e.value), int(r.value)


class DeviceCodebook:
    """A codebook resident in device memory (encoder table + decoder lookup tables)."""

    def __init__(self, ctx: Context, buf: Optional[torch.Tensor] = None):
        self.ctx = ctx
        self.buf = buf if buf is not None else torch.zeros(
            HF_CODEBOOK_DEV_BYTES, dtype=torch.uint8, device=f"cuda:{ctx.device}")

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr()

    def download(self):
        code = np.zeros(256, np.uint16)
        lens = np.zeros(256, np.uint8)
        mode = C.c_int(0)
        ent = C.c_uint32(0)
        check(L.lib().hf_codebook_download(self.ctx.handle, C.c_void_p(self.ptr),
                                           _ptr(code, _u16p), _ptr(lens, _u8p), C.byref(mode),
                                           C.byref(ent)), "hf_codebook_download")
        return code, lens, int(mode.value), int(ent.value)


class MemWrapper(_Module):
    """mem_wrapper.scala:6-87: the codebook RAM (write port, read port, bulk read, valid bits);
    ``to_device`` places the table where the encoder and decoder kernels read it."""

    def __init__(self, depth: int = 256, maxCodeLen: int = 16, addrWidth: int = 8, ctx=None):
        super().__init__(ctx)
        self.depth, self.maxCodeLen, self.addrWidth = depth, maxCodeLen, addrWidth
        self._code = np.zeros(depth, np.uint16)
        self._len = np.zeros(depth, np.uint8)
        self._valid = np.zeros(depth, bool)

    def write(self, write_addr: int, write_code: int, write_length: int) -> None:
        self._code[write_addr] = write_code
        self._len[write_addr] = write_length
        self._valid[write_addr] = True

    def read(self, read_addr: int):
        return int(self._code[read_addr]), int(self._len[read_addr]), bool(self._valid[read_addr])

    def bulk_read(self):
        """code_table, length_table, table_ready (all entries written)."""
        return self._code.copy(), self._len.copy(), bool(self._valid.all())

    def flush(self) -> None:
        self._valid[:] = False

    def to_device(self, mode: int = HF_MODE_HUFFMAN) -> DeviceCodebook:
        if self.depth != 256 or not self._valid.all():
            raise ValueError("table_ready is low: write all 256 entries first")
        cb = DeviceCodebook(self.ctx)
        check(L.lib().hf_codebook_upload(self.ctx.handle, _ptr(self._code, _u16p),
                                         _ptr(self._len, _u8p), mode, C.c_void_p(cb.ptr)),
              "hf_codebook_upload")
        return cb


@dataclass
class EncodedStream:
    """What the encoder leaves in device memory for one tensor."""
    n: int
    layout: int
    dtype: torch.dtype
    payload: torch.Tensor      # int32 words (capacity may exceed total_words)
    chunk_off: torch.Tensor    # int32[nchunks + 1]
    side: Optional[torch.Tensor]  # uint8[n] for bf16, else None
    codebook: Optional[DeviceCodebook]
    total_words: Optional[int] = None
    chunk_codebooks: Optional[torch.Tensor] = None  # per-block codebook records (uint8)
    chunk_freq: Optional[torch.Tensor] = None       # per-block counts, int32[nchunks * 256]
    # a slotted stream (hf_encode_chunks_slotted): chunk_off are slot starts, and these the words
    # each chunk filled of its slot; None for a dense (format-v1) payload
    chunk_words: Optional[torch.Tensor] = None
    # a paged stream (format version 2): chunk_off are page offsets.  With codebook = None
    # (hf_encode_pages) every page carries its own codebook record and the decoder needs nothing
    # but payload and offsets; with a codebook (hf_encode_pages_shared) the pages have no record
    # and are all coded with that one table
    paged: bool = False

    @property
    def nchunks(self) -> int:
        return (self.n + HF_CHUNK_SYMS - 1) // HF_CHUNK_SYMS

    def compressed_bytes(self) -> int:
        """payload + side stream + chunk offsets + codebooks.  A stream-wide codebook counts as
        256 x (16-bit code, 8-bit length); a per-block codebook as its record in the container
        (HF_CONTAINER_CHUNK_CODEBOOK_BYTES = 296 bytes per block: 256 symbol ids, 16 length
        counts, mode, entropy).  Section padding and the header are left to container_bytes()."""
        assert self.total_words is not None
        side = side_bytes(self.n, self.layout)
        if self.paged:  # the records are part of the pages, unless one table codes them all
            return 4 * self.total_words + side + 4 * (self.nchunks + 1) + \
                (256 * 3 if self.codebook is not None else 0)
        books = 256 * 3 if self.chunk_codebooks is None \
            else L.HF_CONTAINER_CHUNK_CODEBOOK_BYTES * self.nchunks
        index = 4 * (self.nchunks + 1) + (4 * self.nchunks if self.chunk_words is not None else 0)
        return 4 * self.total_words + side + index + books

    def ratio(self) -> float:
        return self.n * element_bytes(self.layout) / max(1, self.compressed_bytes())

    # -- container (wire format, include/huffhip.h) ------------------------------------------------
    @property
    def placement(self) -> int:
        return L.HF_PLACEMENT_PAGED if self.paged else (
            L.HF_PLACEMENT_SLOTTED if self.chunk_words is not None else L.HF_PLACEMENT_DENSE)

    def _plan(self):
        assert self.total_words is not None, "encode with sync=True first"
        info = L.hf_container_info()
        per_chunk = self.chunk_codebooks is not None or (self.paged and self.codebook is None)
        check(L.lib().hf_container_plan_placed(self.n, self.layout, 1 if per_chunk else 0,
                                               self.placement, self.total_words, C.byref(info)),
              "hf_container_plan_placed")
        return info, per_chunk

    def container_bytes(self) -> int:
        """Size of this stream's container (host arithmetic, nothing is packed)."""
        return int(self._plan()[0].total_bytes)

    def to_container(self, ctx: Optional["Context"] = None) -> torch.Tensor:
        """One self-describing uint8 device tensor: header, codebook(s), chunk offsets, the words
        array of a slotted stream, side bytes, payload.  Dense streams are written as wire format
        version 1, slotted and paged ones as version 2."""
        ctx = ctx if ctx is not None else default_context()
        info, per_chunk = self._plan()
        blob = torch.empty(int(info.total_bytes), dtype=torch.uint8, device=self.payload.device)
        if self.paged and per_chunk:
            cb_ptr = 0
        else:
            cb_ptr = self.chunk_codebooks.data_ptr() if per_chunk else self.codebook.ptr
        check(L.lib().hf_container_pack_placed(ctx.handle, C.byref(info), C.c_void_p(cb_ptr),
                                               C.c_void_p(self.chunk_off.data_ptr()),
                                               C.c_void_p(_dev_ptr(self.chunk_words)),
                                               C.c_void_p(_dev_ptr(self.side)),
                                               C.c_void_p(self.payload.data_ptr()),
                                               C.c_void_p(blob.data_ptr()), blob.numel()),
              "hf_container_pack_placed")
        return blob

    @staticmethod
    def from_container(blob: torch.Tensor, dtype: Optional[torch.dtype] = None,
                       ctx: Optional["Context"] = None) -> "EncodedStream":
        """Open a container in device memory.  The stream's payload, offsets and side bytes are
        views into ``blob`` (nothing is copied); only the codebook tables are rebuilt (a paged
        stream has none outside its pages)."""
        assert blob.dtype == torch.uint8 and blob.is_cuda and blob.is_contiguous()
        ctx = ctx if ctx is not None else default_context()
        info = L.hf_container_info()
        check(L.lib().hf_container_open(ctx.handle, C.c_void_p(blob.data_ptr()), blob.numel(),
                                        C.byref(info)), "hf_container_open")
        n, nch, words = int(info.n), int(info.nchunks), int(info.payload_words)
        paged = info.placement == L.HF_PLACEMENT_PAGED
        per_chunk = info.codebook_kind == L.HF_CODEBOOK_PER_CHUNK
        cbs, cb, cb_ptr = None, None, 0
        if paged and per_chunk:
            pass
        elif per_chunk:
            cbs = torch.empty(max(1, nch) * L.HF_CHUNK_CODEBOOK_BYTES, dtype=torch.uint8,
                              device=blob.device)
            cb_ptr = cbs.data_ptr()
        else:
            cb = DeviceCodebook(ctx)
            cb_ptr = cb.ptr
        check(L.lib().hf_container_codebook(ctx.handle, C.c_void_p(blob.data_ptr()),
                                            C.byref(info), C.c_void_p(cb_ptr)),
              "hf_container_codebook")
        o_off, o_side, o_pay = int(info.off_chunk_off), int(info.off_side), int(info.off_payload)
        chunk_off = blob[o_off:o_off + 4 * (nch + 1)].view(torch.int32)
        chunk_words = None
        if info.placement == L.HF_PLACEMENT_SLOTTED:
            o_w = int(info.off_chunk_words)
            # (padded to its 16-byte boundary: the decoders want aligned arrays, whatever nch is)
            chunk_words = blob[o_w:o_w + 4 * (((nch + 3) // 4) * 4)].view(torch.int32)
        nside = side_bytes(n, int(info.layout))
        side = blob[o_side:o_side + nside] if o_side else None
        payload = blob[o_pay:int(info.total_bytes)].view(torch.int32)  # padded to >= 4 words
        if dtype is None:
            dtype = {HF_LAYOUT_BF16_EXP: torch.bfloat16,
                     HF_LAYOUT_FP16_EXP: torch.float16}.get(int(info.layout), torch.uint8)
        return EncodedStream(n=n, layout=int(info.layout), dtype=dtype, payload=payload,
                             chunk_off=chunk_off, side=side, codebook=cb, total_words=words,
                             chunk_codebooks=cbs, chunk_words=chunk_words, paged=paged)


class EntropyAwareHuffmanSystem(_Module):
    """top_encoder.scala:6-184: data_in -> encoded_out, with the bit packer behind it.

    ``encode`` runs the whole pipeline on the device: SymbolStat -> SymbolSort ->
    ShannonEntropy decision -> EntropyAwareTreeBuilder -> lookup + packing.
    """

    def __init__(self, symWidth: int = 8, wtWidth: int = 32, depth: int = 256,
                 maxCodeLen: int = 16, ctx=None):
        super().__init__(ctx)
        if depth != 256 or symWidth != 8 or maxCodeLen != 16:
            raise ValueError("the packed stream is defined for symWidth=8, depth=256, maxCodeLen=16")
        self.symWidth, self.wtWidth, self.depth, self.maxCodeLen = symWidth, wtWidth, depth, maxCodeLen

    # -- stages --------------------------------------------------------------------------------
    def histogram(self, data_in: torch.Tensor, layout: Optional[int] = None,
                  freq256: Optional[torch.Tensor] = None) -> torch.Tensor:
        if freq256 is None:
            # cleared on the context's stream, where the kernel that adds into it runs
            with torch.cuda.stream(self.ctx.stream):
                freq256 = torch.zeros(256, dtype=torch.int64, device=data_in.device)
        SymbolStat(ctx=self.ctx).accumulate(data_in, freq256, layout)
        return freq256

    @_on_ctx_stream
    def histogram_pages(self, data_in: torch.Tensor, layout: Optional[int] = None,
                        freq_chunks: Optional[torch.Tensor] = None,
                        freq256: Optional[torch.Tensor] = None):
        """SymbolStat per 16384-symbol page and of the whole stream from one read of the data
        (``hf_symbol_stat_pages_dev``).
        Returns (int32[nchunks * 256], int64[256]); ``freq256`` given is added into."""
        lay = layout_of(data_in) if layout is None else layout
        data_in = data_in.contiguous()
        n = data_in.numel()
        nch = (n + HF_CHUNK_SYMS - 1) // HF_CHUNK_SYMS
        if freq_chunks is None:
            freq_chunks = torch.empty(max(1, nch) * 256, dtype=torch.int32, device=data_in.device)
        if freq256 is None:
            freq256 = torch.zeros(256, dtype=torch.int64, device=data_in.device)
        check(L.lib().hf_symbol_stat_pages_dev(self.ctx.handle, C.c_void_p(_dev_ptr(data_in)), n,
                                               lay, C.c_void_p(freq_chunks.data_ptr()),
                                               C.c_void_p(freq256.data_ptr())),
              "hf_symbol_stat_pages_dev")
        return freq_chunks, freq256

    def codebook(self, freq256: torch.Tensor, mode: int = HF_MODE_AUTO,
                 min_freq_threshold: int = 0,
                 out: Optional[DeviceCodebook] = None) -> DeviceCodebook:
        cb = out if out is not None else DeviceCodebook(self.ctx)
        check(L.lib().hf_system_codebook_dev(self.ctx.handle, C.c_void_p(freq256.data_ptr()), mode,
                                             min_freq_threshold, C.c_void_p(cb.ptr)),
              "hf_system_codebook_dev")
        return cb

    @_on_ctx_stream
    def codebook_of_block(self, data_in: torch.Tensor, mode: int = HF_MODE_AUTO,
                          min_freq_threshold: int = 0, layout: Optional[int] = None,
                          freq256: Optional[torch.Tensor] = None,
                          out: Optional[DeviceCodebook] = None):
        """``hf_system_codebook_block_dev``: SymbolStat and the codebook of one block of at most
        HF_BLOCK_MAX_SYMS symbols in one launch.  Returns (codebook, counts int64[256])."""
        lay = layout_of(data_in) if layout is None else layout
        data_in = data_in.contiguous()
        cb = out if out is not None else DeviceCodebook(self.ctx)
        if freq256 is None:
            freq256 = torch.empty(256, dtype=torch.int64, device=data_in.device)
        check(L.lib().hf_system_codebook_block_dev(
            self.ctx.handle, C.c_void_p(_dev_ptr(data_in)), data_in.numel(), lay, mode,
            min_freq_threshold, C.c_void_p(freq256.data_ptr()), C.c_void_p(cb.ptr)),
            "hf_system_codebook_block_dev")
        return cb, freq256

    @_on_ctx_stream
    def encode_with(self, data_in: torch.Tensor, cb: DeviceCodebook, layout: Optional[int] = None,
                    payload: Optional[torch.Tensor] = None, side: Optional[torch.Tensor] = None,
                    chunk_off: Optional[torch.Tensor] = None, sync: bool = True,
                    paged: bool = False,
                    chunk_freq: Optional[torch.Tensor] = None) -> EncodedStream:
        """Pack ``data_in`` with a codebook made elsewhere (another tensor's, a global one summed
        over ranks).  ``paged``: as pages without records (``hf_encode_pages_shared``), sized from
        the per-page counts ``chunk_freq`` (``histogram_pages``) when given, else placed while they
        are encoded (one read of the data, no counts; the same stream bit for bit)."""
        lay = layout_of(data_in) if layout is None else layout
        data_in = data_in.contiguous()
        n = data_in.numel()
        dev = data_in.device
        nch = (n + HF_CHUNK_SYMS - 1) // HF_CHUNK_SYMS
        if paged:
            return self._pack_pages_shared(data_in, lay, cb, chunk_freq, payload, side, chunk_off,
                                           sync)
        own_payload = payload is None
        if own_payload:
            # a codebook built from other counts than this tensor's (a global or foreign table,
            # a frequency threshold) can pass 8 bits per symbol: that is reported as
            # HF_ERR_OVERFLOW, never overrun, and answered below with the worst-case size
            payload = torch.empty(_payload_words_default(n, nch), dtype=torch.int32, device=dev)
        if chunk_off is None:
            chunk_off = torch.empty(((nch + 1 + 3) // 4) * 4, dtype=torch.int32, device=dev)
        if lay == HF_LAYOUT_BF16_EXP and side is None:
            side = torch.empty(max(16, n), dtype=torch.uint8, device=dev)
        total = C.c_uint64(0)

        def run(buf):
            return L.lib().hf_encode(self.ctx.handle, C.c_void_p(_dev_ptr(data_in)), n, lay,
                                     C.c_void_p(cb.ptr), C.c_void_p(_dev_ptr(side)),
                                     C.c_void_p(_dev_ptr(buf)), buf.numel(),
                                     C.c_void_p(_dev_ptr(chunk_off)),
                                     C.byref(total) if sync else None)
        rc = run(payload)
        if rc == L.HF_ERR_OVERFLOW and own_payload:
            payload = torch.empty(int(L.lib().hf_payload_bound_words(n)), dtype=torch.int32,
                                  device=dev)
            rc = run(payload)
        check(rc, "hf_encode")
        return EncodedStream(n=n, layout=lay, dtype=data_in.dtype, payload=payload,
                             chunk_off=chunk_off, side=side if lay == HF_LAYOUT_BF16_EXP else None,
                             codebook=cb, total_words=int(total.value) if sync else None)

    def _pack_pages_shared(self, data_in, lay, cb, freq_chunks, payload, side, chunk_off,
                           sync) -> EncodedStream:
        n, dev = data_in.numel(), data_in.device
        nch = (n + HF_CHUNK_SYMS - 1) // HF_CHUNK_SYMS
        if chunk_off is None:
            chunk_off = torch.empty(((nch + 1 + 3) // 4) * 4, dtype=torch.int32, device=dev)
        if lay == HF_LAYOUT_BF16_EXP and side is None:
            side = torch.empty(max(16, n), dtype=torch.uint8, device=dev)
        total = C.c_uint64(0)

        def run_pages(buf):
            return L.lib().hf_encode_pages_shared(
                self.ctx.handle, C.c_void_p(_dev_ptr(data_in)), n, lay, C.c_void_p(cb.ptr),
                C.c_void_p(_dev_ptr(freq_chunks)), C.c_void_p(_dev_ptr(side)),
                C.c_void_p(buf.data_ptr()), buf.numel(), C.c_void_p(chunk_off.data_ptr()),
                C.byref(total) if sync else None)
        own_payload = payload is None
        if own_payload:
            # (8 bits per symbol plus the page's header and region lengths; then the worst case)
            payload = torch.empty(max(16, nch * (HF_CHUNK_SYMS // 4 + 64)), dtype=torch.int32,
                                  device=dev)
        rc = run_pages(payload)
        if rc == L.HF_ERR_OVERFLOW and own_payload:
            payload = torch.empty(max(16, int(L.lib().hf_pages_bound_words(n))),
                                  dtype=torch.int32, device=dev)
            rc = run_pages(payload)
        check(rc, "hf_encode_pages_shared")
        return EncodedStream(n=n, layout=lay, dtype=data_in.dtype, payload=payload,
                             chunk_off=chunk_off, side=side if lay == HF_LAYOUT_BF16_EXP else None,
                             codebook=cb, total_words=int(total.value) if sync else None,
                             chunk_freq=freq_chunks, paged=True)

    # -- sub-field front end ---------------------------------------------------------------------
    def split_fields(self, data_in: torch.Tensor, layout: int):
        """``hf_fields_split``: (symbol bytes uint8[n], packed side stream uint8) of a tensor of
        fp16 (HF_LAYOUT_FP16_EXP), fp8 E4M3 (HF_LAYOUT_FP8_E4M3_EXP) or packed fp4 E2M1 pairs
        (HF_LAYOUT_FP4_E2M1_EXP); HF_LAYOUT_BITPLANE8: (the bytes transposed into bit planes,
        None)."""
        assert layout in FIELD_LAYOUTS and data_in.element_size() == element_bytes(layout)
        data_in = data_in.contiguous()
        n = data_in.numel()
        nside = side_bytes(n, layout)
        with torch.cuda.stream(self.ctx.stream):
            sym = torch.empty(max(16, n), dtype=torch.uint8, device=data_in.device)
            side = torch.empty(max(16, nside), dtype=torch.uint8, device=data_in.device) \
                if layout != HF_LAYOUT_BITPLANE8 else None
        check(L.lib().hf_fields_split(self.ctx.handle, C.c_void_p(_dev_ptr(data_in)), n, layout,
                                      C.c_void_p(sym.data_ptr()), C.c_void_p(_dev_ptr(side))),
              "hf_fields_split")
        return sym[:n], (side[:nside] if side is not None else None)

    # -- whole path ----------------------------------------------------------------------------
    def encode(self, data_in: torch.Tensor, mode: int = HF_MODE_AUTO,
               layout: Optional[int] = None, min_freq_threshold: int = 0,
               slotted: bool = False, paged: bool = False,
               page_counts: bool = True) -> EncodedStream:
        """``slotted``: one codebook for the stream, every chunk in a slot sized from the chunk's
        own counts (``hf_encode_slotted``): 1.5-3 % more words, written faster.

        ``paged``: one codebook for the stream, the chunks stored as pages without records
        (``hf_encode_pages_shared``): exact sizes, no look-ahead words; decoded by four
        sub-decoders per page from tables made once per call.  With ``page_counts`` (the faster
        way when the stream's counts have to be taken anyway) the pages are placed beforehand from
        per-page counts taken with the stream's in one read (``histogram_pages``), which the stream
        keeps in ``chunk_freq``; without, SymbolStat of the stream alone, and the encoder places
        the pages while it encodes them.  Both give the same stream."""
        data_in = data_in.contiguous()
        if layout in FIELD_LAYOUTS:
            sym, side = self.split_fields(data_in, layout)
            s = self.encode(sym, mode, HF_LAYOUT_U8, min_freq_threshold, slotted, paged, page_counts)
            s.layout, s.side, s.dtype = layout, side, data_in.dtype
            return s
        if paged and not page_counts:
            freq = self.histogram(data_in, layout)
            cb = self.codebook(freq, mode, min_freq_threshold)
            return self.encode_with(data_in, cb, layout, paged=True)
        if slotted or paged:
            return self._encode_slotted(data_in, mode, layout, min_freq_threshold, paged)
        # one chunk: counted where the codebook is built (larger blocks may still ask for it, but a
        # block of one repeated symbol serialises the single counter array: kept to one chunk here)
        if data_in.numel() <= HF_CHUNK_SYMS:
            cb, _ = self.codebook_of_block(data_in, mode, min_freq_threshold, layout)
            return self.encode_with(data_in, cb, layout)
        freq = self.histogram(data_in, layout)
        cb = self.codebook(freq, mode, min_freq_threshold)
        return self.encode_with(data_in, cb, layout)

    @_on_ctx_stream
    def _encode_slotted(self, data_in, mode, layout, min_freq_threshold,
                        paged: bool = False) -> EncodedStream:
        lay = layout_of(data_in) if layout is None else layout
        n, dev, h = data_in.numel(), data_in.device, self.ctx.handle
        nch = (n + HF_CHUNK_SYMS - 1) // HF_CHUNK_SYMS
        freq_chunks, freq = self.histogram_pages(data_in, lay)  # one read for both
        cb = self.codebook(freq, mode, min_freq_threshold)
        chunk_off = torch.empty(((nch + 1 + 3) // 4) * 4, dtype=torch.int32, device=dev)
        chunk_words = torch.empty(((nch + 3) // 4) * 4 + 4, dtype=torch.int32, device=dev)
        side = torch.empty(max(16, n), dtype=torch.uint8, device=dev) \
            if lay == HF_LAYOUT_BF16_EXP else None
        total = C.c_uint64(0)
        if paged:
            return self._pack_pages_shared(data_in, lay, cb, freq_chunks, None, side, chunk_off, True)

        def run(buf):
            return L.lib().hf_encode_slotted(
                h, C.c_void_p(_dev_ptr(data_in)), n, lay, C.c_void_p(cb.ptr),
                C.c_void_p(freq_chunks.data_ptr()), C.c_void_p(_dev_ptr(side)),
                C.c_void_p(buf.data_ptr()), buf.numel(), C.c_void_p(chunk_off.data_ptr()),
                C.c_void_p(chunk_words.data_ptr()), C.byref(total))
        # (a codebook built from the stream's own counts averages <= 8 bits per symbol over the
        # stream, not over every chunk: size for the average plus slack, then for the worst case)
        payload = torch.empty(max(16, nch * int(L.HF_SLOT_WORDS(HF_CHUNK_SYMS * 8))),
                              dtype=torch.int32, device=dev)
        rc = run(payload)
        if rc == L.HF_ERR_OVERFLOW:
            payload = torch.empty(max(16, int(L.lib().hf_slotted_bound_words(n))),
                                  dtype=torch.int32, device=dev)
            rc = run(payload)
        check(rc, "hf_encode_slotted")
        return EncodedStream(n=n, layout=lay, dtype=data_in.dtype, payload=payload,
                             chunk_off=chunk_off, side=side, codebook=cb,
                             total_words=int(total.value), chunk_freq=freq_chunks,
                             chunk_words=chunk_words)

    # -- per-block codebooks (one codebook per 16384-symbol chunk, e.g. a KV-cache page) --------
    @_on_ctx_stream
    def encode_blocks(self, data_in: torch.Tensor, mode: int = HF_MODE_AUTO,
                      layout: Optional[int] = None, min_freq_threshold: int = 0,
                      sync: bool = True, slotted: bool = False,
                      paged: bool = False) -> EncodedStream:
        """Every chunk is a block of its own: SymbolStat, SymbolSort, the entropy decision and
        EntropyAwareTreeBuilder run once per chunk, then the chunk is packed with its table.

        ``slotted``: every chunk goes to a slot sized from its histogram and codebook before the
        encoder runs (``hf_encode_chunks_slotted``): 2-4 % more words, written faster; the stream
        has ``chunk_words`` beside ``chunk_off`` and no container form.

        ``paged``: the paged stream (``hf_encode_pages``, format version 2): every page carries its
        codebook record and is sized exactly from its histogram, no look-back and no slack; decoded
        by ``hf_decode_pages``, four sub-decoders per page."""
        data_in = data_in.contiguous()
        if layout in FIELD_LAYOUTS:
            sym, side = self.split_fields(data_in, layout)
            s = self.encode_blocks(sym, mode, HF_LAYOUT_U8, min_freq_threshold, sync, slotted, paged)
            s.layout, s.side, s.dtype = layout, side, data_in.dtype
            return s
        lay = layout_of(data_in) if layout is None else layout
        n = data_in.numel()
        dev = data_in.device
        nch = (n + HF_CHUNK_SYMS - 1) // HF_CHUNK_SYMS
        freq = torch.empty(max(1, nch) * 256, dtype=torch.int32, device=dev)
        cbs = torch.empty(max(1, nch) * L.HF_CHUNK_CODEBOOK_BYTES, dtype=torch.uint8, device=dev)
        h = self.ctx.handle
        check(L.lib().hf_symbol_stat_chunks_dev(h, C.c_void_p(_dev_ptr(data_in)), n, lay,
                                                C.c_void_p(freq.data_ptr())),
              "hf_symbol_stat_chunks_dev")
        check(L.lib().hf_system_codebook_chunks_dev(h, C.c_void_p(freq.data_ptr()), nch, mode,
                                                    min_freq_threshold, C.c_void_p(cbs.data_ptr())),
              "hf_system_codebook_chunks_dev")
        chunk_off = torch.empty(((nch + 1 + 3) // 4) * 4, dtype=torch.int32, device=dev)
        side = torch.empty(max(16, n), dtype=torch.uint8, device=dev) \
            if lay == HF_LAYOUT_BF16_EXP else None
        total = C.c_uint64(0)
        if paged:
            def run_pages(buf):
                return L.lib().hf_encode_pages(
                    h, C.c_void_p(_dev_ptr(data_in)), n, lay, C.c_void_p(cbs.data_ptr()),
                    C.c_void_p(freq.data_ptr()), C.c_void_p(_dev_ptr(side)),
                    C.c_void_p(buf.data_ptr()), buf.numel(), C.c_void_p(chunk_off.data_ptr()),
                    C.byref(total) if sync else None)
            # (8 bits per symbol plus a page's record and region lengths; then the worst case)
            payload = torch.empty(max(16, nch * (HF_CHUNK_SYMS // 4 + 128)), dtype=torch.int32,
                                  device=dev)
            rc = run_pages(payload)
            if rc == L.HF_ERR_OVERFLOW:
                payload = torch.empty(max(16, int(L.lib().hf_pages_bound_words(n))),
                                      dtype=torch.int32, device=dev)
                rc = run_pages(payload)
            check(rc, "hf_encode_pages")
            return EncodedStream(n=n, layout=lay, dtype=data_in.dtype, payload=payload,
                                 chunk_off=chunk_off, side=side, codebook=None,
                                 total_words=int(total.value) if sync else None,
                                 chunk_codebooks=cbs, chunk_freq=freq, paged=True)
        if slotted:
            # (slots of the default size hold any block whose codes average <= 8 bits per symbol,
            # i.e. every block unless min_freq_threshold made its codebook from other counts)
            chunk_words = torch.empty(((nch + 3) // 4) * 4 + 4, dtype=torch.int32, device=dev)

            def run_slotted(buf):
                return L.lib().hf_encode_chunks_slotted(
                    h, C.c_void_p(_dev_ptr(data_in)), n, lay, C.c_void_p(cbs.data_ptr()),
                    C.c_void_p(freq.data_ptr()), C.c_void_p(_dev_ptr(side)),
                    C.c_void_p(buf.data_ptr()), buf.numel(), C.c_void_p(chunk_off.data_ptr()),
                    C.c_void_p(chunk_words.data_ptr()), C.byref(total) if sync else None)
            payload = torch.empty(max(16, nch * int(L.HF_SLOT_WORDS(HF_CHUNK_SYMS * 8))),
                                  dtype=torch.int32, device=dev)
            rc = run_slotted(payload)
            if rc == L.HF_ERR_OVERFLOW:
                payload = torch.empty(max(16, int(L.lib().hf_slotted_bound_words(n))),
                                      dtype=torch.int32, device=dev)
                rc = run_slotted(payload)
            check(rc, "hf_encode_chunks_slotted")
            return EncodedStream(n=n, layout=lay, dtype=data_in.dtype, payload=payload,
                                 chunk_off=chunk_off, side=side, codebook=None,
                                 total_words=int(total.value) if sync else None,
                                 chunk_codebooks=cbs, chunk_freq=freq, chunk_words=chunk_words)
        payload = torch.empty(_payload_words_default(n, nch), dtype=torch.int32, device=dev)

        def run(buf):
            return L.lib().hf_encode_chunks(h, C.c_void_p(_dev_ptr(data_in)), n, lay,
                                            C.c_void_p(cbs.data_ptr()), C.c_void_p(_dev_ptr(side)),
                                            C.c_void_p(buf.data_ptr()), buf.numel(),
                                            C.c_void_p(chunk_off.data_ptr()),
                                            C.byref(total) if sync else None)
        rc = run(payload)
        if rc == L.HF_ERR_OVERFLOW:  # (min_freq_threshold can push a block past 8 bits per symbol)
            payload = torch.empty(int(L.lib().hf_payload_bound_words(n)), dtype=torch.int32,
                                  device=dev)
            rc = run(payload)
        check(rc, "hf_encode_chunks")
        return EncodedStream(n=n, layout=lay, dtype=data_in.dtype, payload=payload,
                             chunk_off=chunk_off, side=side, codebook=None,
                             total_words=int(total.value) if sync else None,
                             chunk_codebooks=cbs, chunk_freq=freq)

    def block_codebook(self, s: EncodedStream, chunk: int):
        """code, length (by symbol), compression_mode and entropy of one block's codebook."""
        code = np.zeros(256, np.uint16)
        lens = np.zeros(256, np.uint8)
        mode = C.c_int(0)
        ent = C.c_uint32(0)
        if s.codebook is not None and s.chunk_codebooks is None:  # one table for every block
            return s.codebook.download()
        if s.paged:  # read back from the page's own record
            check(L.lib().hf_page_codebook_download(
                self.ctx.handle, C.c_void_p(s.payload.data_ptr()), s.payload.numel(),
                C.c_void_p(s.chunk_off.data_ptr()), chunk, _ptr(code, _u16p), _ptr(lens, _u8p),
                C.byref(mode), C.byref(ent)), "hf_page_codebook_download")
            return code, lens, int(mode.value), int(ent.value)
        check(L.lib().hf_chunk_codebook_download(self.ctx.handle,
                                                 C.c_void_p(s.chunk_codebooks.data_ptr()), chunk,
                                                 _ptr(code, _u16p), _ptr(lens, _u8p),
                                                 C.byref(mode), C.byref(ent)),
              "hf_chunk_codebook_download")
        return code, lens, int(mode.value), int(ent.value)

    @_on_ctx_stream
    def encode_pairs(self, data_in: torch.Tensor, cb: DeviceCodebook,
                     layout: Optional[int] = None) -> np.ndarray:
        """``encoded_out`` beats, one per symbol: columns code, length, compression_mode."""
        lay = layout_of(data_in) if layout is None else layout
        data_in = data_in.contiguous()
        n = data_in.numel()
        beats = torch.empty(max(1, n), dtype=torch.int32, device=data_in.device)
        check(L.lib().hf_encode_pairs(self.ctx.handle, C.c_void_p(_dev_ptr(data_in)), n, lay,
                                      C.c_void_p(cb.ptr), C.c_void_p(beats.data_ptr())),
              "hf_encode_pairs")
        self.ctx.sync()
        raw = beats[:n].cpu().numpy().view(np.uint32)
        return np.stack([raw & 0xFFFF, (raw >> 16) & 0xFF, raw >> 24], axis=1).astype(np.uint32)


class ParrallelHuffmanDecoder(_Module):  # spelling as in decoder/top_decoder.scala:6
    """decoder/top_decoder.scala:6-10 (constructor only in the reference)."""

    def __init__(self, symWidth: int = 8, wtWidth: int = 32, depth: int = 256,
                 maxCodeLen: int = 16, ctx=None):
        super().__init__(ctx)
        if depth != 256 or symWidth != 8 or maxCodeLen != 16:
            raise ValueError("the packed stream is defined for symWidth=8, depth=256, maxCodeLen=16")

    @_on_ctx_stream
    def _merge_fields(self, s: EncodedStream, sym: torch.Tensor,
                      out: Optional[torch.Tensor]) -> torch.Tensor:
        """``hf_fields_merge``: decoded symbol bytes + the stream's side fields -> elements."""
        if out is None:
            out = torch.empty(max(s.n, 16), dtype=s.dtype, device=sym.device)
        check(L.lib().hf_fields_merge(self.ctx.handle, C.c_void_p(_dev_ptr(sym)),
                                      C.c_void_p(_dev_ptr(s.side)), s.n, s.layout,
                                      C.c_void_p(_dev_ptr(out))), "hf_fields_merge")
        self.ctx.sync()
        return out[:s.n]

    @staticmethod
    def _symbol_view(s: EncodedStream) -> EncodedStream:
        """The same stream as the byte-symbol kernels see it (sub-field layouts)."""
        import dataclasses
        return dataclasses.replace(s, layout=HF_LAYOUT_U8, dtype=torch.uint8, side=None)

    @_on_ctx_stream
    def decode_blocks(self, s: EncodedStream, chunk_ids: Optional[torch.Tensor] = None,
                      out: Optional[torch.Tensor] = None, check_stream: bool = True) -> torch.Tensor:
        """Decode a stream with per-block codebooks.  With ``chunk_ids`` (int32 device tensor)
        this is decode-on-read: the listed chunks, in that order, land back to back in ``out``."""
        assert s.chunk_codebooks is not None or s.paged
        if s.layout in FIELD_LAYOUTS:
            if chunk_ids is not None:
                raise ValueError("decode-on-read of a sub-field stream: decode the symbols with "
                                 "the HF_LAYOUT_U8 view and join the fields of the pages wanted")
            sym = self.decode_blocks(self._symbol_view(s), None, None, check_stream)
            return self._merge_fields(s, sym, out)
        if chunk_ids is None:
            n_out, n_ids = s.n, 0
        else:
            assert chunk_ids.dtype == torch.int32 and chunk_ids.is_cuda
            n_ids = chunk_ids.numel()
            n_out = n_ids * HF_CHUNK_SYMS
        if out is None:
            out = torch.empty(max(n_out, 16), dtype=s.dtype, device=s.payload.device)
        if s.paged and s.codebook is not None:
            check(L.lib().hf_decode_pages_shared(
                self.ctx.handle, C.c_void_p(s.payload.data_ptr()), s.payload.numel(),
                C.c_void_p(s.chunk_off.data_ptr()), C.c_void_p(_dev_ptr(s.side)), s.n, s.layout,
                C.c_void_p(s.codebook.ptr), C.c_void_p(_dev_ptr(chunk_ids)), None, n_ids,
                C.c_void_p(_dev_ptr(out)), 0, 1 if check_stream else 0), "hf_decode_pages_shared")
            return out[:n_out]
        if s.paged:
            check(L.lib().hf_decode_pages(
                self.ctx.handle, C.c_void_p(s.payload.data_ptr()), s.payload.numel(),
                C.c_void_p(s.chunk_off.data_ptr()), C.c_void_p(_dev_ptr(s.side)), s.n, s.layout,
                C.c_void_p(_dev_ptr(chunk_ids)), None, n_ids, C.c_void_p(_dev_ptr(out)), 0,
                1 if check_stream else 0), "hf_decode_pages")
            return out[:n_out]
        if s.chunk_words is not None:
            check(L.lib().hf_decode_chunks_slotted(
                self.ctx.handle, C.c_void_p(s.payload.data_ptr()), s.payload.numel(),
                C.c_void_p(s.chunk_off.data_ptr()), C.c_void_p(s.chunk_words.data_ptr()),
                C.c_void_p(_dev_ptr(s.side)), s.n, s.layout,
                C.c_void_p(s.chunk_codebooks.data_ptr()), C.c_void_p(_dev_ptr(chunk_ids)), None,
                n_ids, C.c_void_p(_dev_ptr(out)), 0, 1 if check_stream else 0),
                "hf_decode_chunks_slotted")
            return out[:n_out]
        check(L.lib().hf_decode_chunks(self.ctx.handle, C.c_void_p(s.payload.data_ptr()),
                                       s.payload.numel(), C.c_void_p(s.chunk_off.data_ptr()),
                                       C.c_void_p(_dev_ptr(s.side)), s.n, s.layout,
                                       C.c_void_p(s.chunk_codebooks.data_ptr()),
                                       C.c_void_p(_dev_ptr(chunk_ids)), n_ids,
                                       C.c_void_p(_dev_ptr(out)), 1 if check_stream else 0),
              "hf_decode_chunks")
        return out[:n_out]

    @_on_ctx_stream
    def decode_blocks_to(self, s: EncodedStream, chunk_ids: torch.Tensor, out_slots: torch.Tensor,
                         page_pool: torch.Tensor, check_stream: bool = True) -> torch.Tensor:
        """Decode-on-read into a paged buffer: chunk ``chunk_ids[k]`` lands in page
        ``out_slots[k]`` of ``page_pool`` (a contiguous tensor of whole 16384-element pages)."""
        assert s.chunk_codebooks is not None or s.paged
        for t in (chunk_ids, out_slots):
            assert t.dtype == torch.int32 and t.is_cuda and t.is_contiguous()
        assert chunk_ids.numel() == out_slots.numel()
        assert page_pool.is_cuda and page_pool.is_contiguous() and page_pool.dtype == s.dtype
        assert page_pool.numel() % HF_CHUNK_SYMS == 0
        if s.paged and s.codebook is not None:
            check(L.lib().hf_decode_pages_shared(
                self.ctx.handle, C.c_void_p(s.payload.data_ptr()), s.payload.numel(),
                C.c_void_p(s.chunk_off.data_ptr()), C.c_void_p(_dev_ptr(s.side)), s.n, s.layout,
                C.c_void_p(s.codebook.ptr), C.c_void_p(chunk_ids.data_ptr()),
                C.c_void_p(out_slots.data_ptr()), chunk_ids.numel(),
                C.c_void_p(page_pool.data_ptr()), page_pool.numel() // HF_CHUNK_SYMS,
                1 if check_stream else 0), "hf_decode_pages_shared")
            return page_pool
        if s.paged:
            check(L.lib().hf_decode_pages(
                self.ctx.handle, C.c_void_p(s.payload.data_ptr()), s.payload.numel(),
                C.c_void_p(s.chunk_off.data_ptr()), C.c_void_p(_dev_ptr(s.side)), s.n, s.layout,
                C.c_void_p(chunk_ids.data_ptr()), C.c_void_p(out_slots.data_ptr()),
                chunk_ids.numel(), C.c_void_p(page_pool.data_ptr()),
                page_pool.numel() // HF_CHUNK_SYMS, 1 if check_stream else 0), "hf_decode_pages")
            return page_pool
        if s.chunk_words is not None:
            check(L.lib().hf_decode_chunks_slotted(
                self.ctx.handle, C.c_void_p(s.payload.data_ptr()), s.payload.numel(),
                C.c_void_p(s.chunk_off.data_ptr()), C.c_void_p(s.chunk_words.data_ptr()),
                C.c_void_p(_dev_ptr(s.side)), s.n, s.layout,
                C.c_void_p(s.chunk_codebooks.data_ptr()), C.c_void_p(chunk_ids.data_ptr()),
                C.c_void_p(out_slots.data_ptr()), chunk_ids.numel(),
                C.c_void_p(page_pool.data_ptr()), page_pool.numel() // HF_CHUNK_SYMS,
                1 if check_stream else 0), "hf_decode_chunks_slotted")
            return page_pool
        check(L.lib().hf_decode_chunks_to(self.ctx.handle, C.c_void_p(s.payload.data_ptr()),
                                          s.payload.numel(), C.c_void_p(s.chunk_off.data_ptr()),
                                          C.c_void_p(_dev_ptr(s.side)), s.n, s.layout,
                                          C.c_void_p(s.chunk_codebooks.data_ptr()),
                                          C.c_void_p(chunk_ids.data_ptr()),
                                          C.c_void_p(out_slots.data_ptr()), chunk_ids.numel(),
                                          C.c_void_p(page_pool.data_ptr()),
                                          page_pool.numel() // HF_CHUNK_SYMS,
                                          1 if check_stream else 0), "hf_decode_chunks_to")
        return page_pool

    @_on_ctx_stream
    def decode(self, s: EncodedStream, out: Optional[torch.Tensor] = None,
               check_stream: bool = True) -> torch.Tensor:
        if s.chunk_codebooks is not None or s.paged:
            return self.decode_blocks(s, None, out, check_stream)
        if s.layout in FIELD_LAYOUTS:
            sym = self.decode(self._symbol_view(s), None, check_stream)
            return self._merge_fields(s, sym, out)
        if out is None:
            out = torch.empty(max(s.n, 16), dtype=s.dtype, device=s.payload.device)
        if s.chunk_words is not None:
            check(L.lib().hf_decode_slotted(
                self.ctx.handle, C.c_void_p(s.payload.data_ptr()), s.payload.numel(),
                C.c_void_p(s.chunk_off.data_ptr()), C.c_void_p(s.chunk_words.data_ptr()),
                C.c_void_p(_dev_ptr(s.side)), s.n, s.layout, C.c_void_p(s.codebook.ptr),
                C.c_void_p(_dev_ptr(out)), 1 if check_stream else 0), "hf_decode_slotted")
            return out[:s.n]
        check(L.lib().hf_decode(self.ctx.handle, C.c_void_p(s.payload.data_ptr()),
                                s.payload.numel(), C.c_void_p(s.chunk_off.data_ptr()),
                                C.c_void_p(_dev_ptr(s.side)), s.n, s.layout,
                                C.c_void_p(s.codebook.ptr), C.c_void_p(_dev_ptr(out)),
                                1 if check_stream else 0), "hf_decode")
        return out[:s.n]
