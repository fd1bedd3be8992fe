// JNA binding of libhuffhip.so (include/huffhip.h) with software twins of the reference's ten live
// Chisel modules (SymbolStat, SymbolSort, ShannonEntropy, AdaptiveThreshold, HuffmanTreeBuilder,
// tree_adjust, EntropyAwareTreeBuilder, MemWrapper, TreeQualityAnalyzer, EntropyAwareHuffmanSystem)
// and of the decoder it only names (ParrallelHuffmanDecoder): same class names, same constructor
// parameters, functions instead of ports.
//
// SOURCE ONLY: the image this library is built in has no JVM, sbt or JNA jar, so this file has
// never been compiled.  It needs net.java.dev.jna:jna (5.x) on the class path and
// libhuffhip.so on jna.library.path.  Device buffers are raw device addresses wrapped in
// com.sun.jna.Pointer (from any HIP binding the JVM side already uses).
package huffman.hip

import com.sun.jna.{Library, Native, Pointer, Structure}
import com.sun.jna.ptr.{FloatByReference, IntByReference, LongByReference, PointerByReference,
  ShortByReference}

trait HuffHip extends Library {
  def hf_abi_version(): Int
  def hf_ctx_create(out: PointerByReference, device: Int, stream: Pointer): Int
  def hf_ctx_destroy(ctx: Pointer): Int
  def hf_flush(ctx: Pointer): Int                                   // every module's `flush`
  def hf_sync(ctx: Pointer): Int     // wait for the stream; returns and clears kept kernel errors
  def hf_error_string(code: Int): String
  def hf_last_hip_error(ctx: Pointer): Int
  // make the context's lazily sized workspaces now (Reserve.ENCODE | CODEBOOKS | PLACED), for
  // streams of up to maxN elements: later enqueue-only calls allocate and synchronise nothing
  def hf_ctx_reserve(ctx: Pointer, maxN: Long, flags: Int): Int
  // capturable clear (a kernel, not a memset node) for the counters the stat calls add into
  def hf_zero_dev(ctx: Pointer, dPtr: Pointer, nwords: Long): Int
  def hf_num_chunks(n: Long): Long
  def hf_payload_bound_words(n: Long): Long                         // worst case over all codebooks
  def hf_set_timing(ctx: Pointer, enabled: Int): Int
  def hf_last_kernel_ms(ctx: Pointer, ms: FloatByReference): Int    // device time of the last call
  // SymbolStat.io: data_in -> freq_out (symbol_stat.scala:7-13)
  def hf_symbol_stat(ctx: Pointer, dData: Pointer, n: Long, layout: Int, depth: Int, wtWidth: Int,
                     hFreq: Array[Long]): Int
  def hf_symbol_stat_dev(ctx: Pointer, dData: Pointer, n: Long, layout: Int, dFreq256: Pointer): Int
  // shards on several GPUs: sum the 256 counts over the ranks of an RCCL communicator (ncclComm_t),
  // in place, on the context's stream; every rank then builds the same codebook
  def hf_symbol_stat_allreduce(ctx: Pointer, ncclComm: Pointer, dFreq256: Pointer): Int
  // front ends (reports/spec.md:22: FP16 / FP8 / FP4 operands, bit-plane compression): exponent
  // field -> symbol byte, the other bits -> packed side stream (Layout.FP16_EXP, FP8_E4M3_EXP,
  // FP4_E2M1_EXP), or bytes -> bit planes (Layout.BITPLANE8: no side stream, n a multiple of 8)
  def hf_fields_side_bytes(n: Long, layout: Int): Long
  def hf_fields_split(ctx: Pointer, dData: Pointer, n: Long, layout: Int, dSym: Pointer,
                      dSide: Pointer): Int
  def hf_fields_merge(ctx: Pointer, dSym: Pointer, dSide: Pointer, n: Long, layout: Int,
                      dOut: Pointer): Int
  // SymbolSort.io: freq_in -> sorted_out (symbol_sort.scala:84-90)
  def hf_symbol_sort(ctx: Pointer, hFreq: Array[Long], depth: Int, hSorted: Array[Long],
                     hRank2Sym: Array[Short]): Int
  // ShannonEntropy.io: freq_in -> entropy_out, compression_mode (shannon_entropy.scala:12-26)
  def hf_entropy(ctx: Pointer, hFreq: Array[Long], depth: Int, entropyQ16: IntByReference,
                 mode: IntByReference): Int
  // HuffmanTreeBuilder.io (tree_build.scala:22-32)
  def hf_tree_build(ctx: Pointer, hFreqs: Array[Long], numLeaves: Int, wtWidth: Int,
                    hWeights: Array[Long], hLefts: Array[Short], hRights: Array[Short],
                    hRoot: ShortByReference, hDepths: Array[Short]): Int
  // tree_adjust.io (tree_adjust.scala:9-14)
  def hf_tree_adjust(ctx: Pointer, hDepthsIn: Array[Short], numLeaves: Int, maxDepth: Int,
                     hDepthsOut: Array[Short]): Int
  // EntropyAwareTreeBuilder.io (entropy_aware_tree_builder.scala:7-19)
  def hf_codebook(ctx: Pointer, hFreq: Array[Long], depth: Int, cfg: EntropyConfig,
                  hCode: Array[Short], hLen: Array[Byte]): Int
  // the same from one block of at most 65536 symbols, counted in the same launch
  def hf_system_codebook_block_dev(ctx: Pointer, dData: Pointer, n: Long, layout: Int, mode: Int,
                                   minFreq: Long, dFreq256: Pointer, dCodebook: Pointer): Int
  def hf_system_codebook_dev(ctx: Pointer, dFreq256: Pointer, mode: Int, minFreq: Long,
                             dCodebook: Pointer): Int
  // MemWrapper.io (mem_wrapper.scala:7-26)
  def hf_codebook_upload(ctx: Pointer, hCode: Array[Short], hLen: Array[Byte], mode: Int,
                         dCodebook: Pointer): Int
  def hf_codebook_download(ctx: Pointer, dCodebook: Pointer, hCode: Array[Short],
                           hLen: Array[Byte], mode: IntByReference, entropy: IntByReference): Int
  // EntropyAwareHuffmanSystem.io: data_in -> encoded_out (top_encoder.scala:12-27)
  def hf_encode_pairs(ctx: Pointer, dData: Pointer, n: Long, layout: Int, dCodebook: Pointer,
                      dBeats: Pointer): Int
  def hf_encode(ctx: Pointer, dData: Pointer, n: Long, layout: Int, dCodebook: Pointer,
                dSide: Pointer, dPayload: Pointer, payloadCapWords: Long, dChunkOff: Pointer,
                hTotalWords: LongByReference): Int
  // ParrallelHuffmanDecoder (decoder/top_decoder.scala:6)
  def hf_decode(ctx: Pointer, dPayload: Pointer, payloadWords: Long, dChunkOff: Pointer,
                dSide: Pointer, n: Long, layout: Int, dCodebook: Pointer, dOut: Pointer,
                check: Int): Int
  // per-block codebooks: one EntropyAwareTreeBuilder codebook per 16384-symbol block
  // (entropy_aware_tree_builder.scala:7-19 applied block by block; KV-cache pages)
  def hf_symbol_stat_pages_dev(ctx: Pointer, dData: Pointer, n: Long, layout: Int,
                               dFreqChunks: Pointer, dFreq256: Pointer): Int
  def hf_symbol_stat_chunks_dev(ctx: Pointer, dData: Pointer, n: Long, layout: Int,
                                dFreqChunks: Pointer): Int              // u32[nchunks][256]
  def hf_system_codebook_chunks_dev(ctx: Pointer, dFreqChunks: Pointer, nchunks: Long, mode: Int,
                                    minFreq: Long, dChunkCodebooks: Pointer): Int
  // the same records without any allocation on the call path (at most 4096 blocks per call; the
  // batch scratch is a static device array): for graph capture and serving loops
  def hf_system_codebook_chunks_static(ctx: Pointer, dFreqChunks: Pointer, nchunks: Long, mode: Int,
                                       minFreq: Long, dChunkCodebooks: Pointer): Int
  def hf_encode_chunks(ctx: Pointer, dData: Pointer, n: Long, layout: Int, dChunkCodebooks: Pointer,
                       dSide: Pointer, dPayload: Pointer, payloadCapWords: Long, dChunkOff: Pointer,
                       hTotalWords: LongByReference): Int
  def hf_decode_chunks(ctx: Pointer, dPayload: Pointer, payloadWords: Long, dChunkOff: Pointer,
                       dSide: Pointer, n: Long, layout: Int, dChunkCodebooks: Pointer,
                       dChunkIds: Pointer, nIds: Long, dOut: Pointer, check: Int): Int
  // decode-on-read into a page pool: chunk dChunkIds(k) -> page dOutSlots(k) of dOut (nSlots pages)
  def hf_decode_chunks_to(ctx: Pointer, dPayload: Pointer, payloadWords: Long, dChunkOff: Pointer,
                          dSide: Pointer, n: Long, layout: Int, dChunkCodebooks: Pointer,
                          dChunkIds: Pointer, dOutSlots: Pointer, nIds: Long, dOut: Pointer,
                          nSlots: Long, check: Int): Int
  // slotted placement: chunk c owns words dChunkOff(c) .. dChunkOff(c + 1) and fills the first
  // dChunkWords(c) of them; faster to write, 2-4 % larger, no container form
  def hf_slotted_bound_words(n: Long): Long
  def hf_encode_chunks_slotted(ctx: Pointer, dData: Pointer, n: Long, layout: Int,
                               dChunkCodebooks: Pointer, dFreqChunks: Pointer, dSide: Pointer,
                               dPayload: Pointer, payloadCapWords: Long, dChunkOff: Pointer,
                               dChunkWords: Pointer, hTotalWords: LongByReference): Int
  def hf_decode_chunks_slotted(ctx: Pointer, dPayload: Pointer, payloadWords: Long,
                               dChunkOff: Pointer, dChunkWords: Pointer, dSide: Pointer, n: Long,
                               layout: Int, dChunkCodebooks: Pointer, dChunkIds: Pointer,
                               dOutSlots: Pointer, nIds: Long, dOut: Pointer, nSlots: Long,
                               check: Int): Int
  // the same placement with one codebook for the stream: count per chunk, fold, build, encode
  def hf_symbol_stat_fold_dev(ctx: Pointer, dFreqChunks: Pointer, nChunks: Long, dFreq256: Pointer): Int
  def hf_encode_slotted(ctx: Pointer, dData: Pointer, n: Long, layout: Int, dCodebook: Pointer,
                        dFreqChunks: Pointer, dSide: Pointer, dPayload: Pointer,
                        payloadCapWords: Long, dChunkOff: Pointer, dChunkWords: Pointer,
                        hTotalWords: LongByReference): Int
  def hf_decode_slotted(ctx: Pointer, dPayload: Pointer, payloadWords: Long, dChunkOff: Pointer,
                        dChunkWords: Pointer, dSide: Pointer, n: Long, layout: Int,
                        dCodebook: Pointer, dOut: Pointer, check: Int): Int
  // paged stream (format version 2): every page carries its codebook record and its exact size
  // follows from its histogram; four sub-decoders per page (reports/figs/decoder.svg)
  def hf_pages_bound_words(n: Long): Long
  def hf_encode_pages(ctx: Pointer, dData: Pointer, n: Long, layout: Int, dChunkCodebooks: Pointer,
                      dFreqChunks: Pointer, dSide: Pointer, dPayload: Pointer, payloadCapWords: Long,
                      dPageOff: Pointer, hTotalWords: LongByReference): Int
  def hf_decode_pages(ctx: Pointer, dPayload: Pointer, payloadWords: Long, dPageOff: Pointer,
                      dSide: Pointer, n: Long, layout: Int, dPageIds: Pointer, dOutSlots: Pointer,
                      nIds: Long, dOut: Pointer, nSlots: Long, check: Int): Int
  // the same pages without records, all coded with the stream's one codebook (dFreqChunks may be
  // null: the pages are then placed while they are encoded, from one read of the data)
  def hf_encode_pages_shared(ctx: Pointer, dData: Pointer, n: Long, layout: Int, dCodebook: Pointer,
                             dFreqChunks: Pointer, dSide: Pointer, dPayload: Pointer,
                             payloadCapWords: Long, dPageOff: Pointer,
                             hTotalWords: LongByReference): Int
  def hf_decode_pages_shared(ctx: Pointer, dPayload: Pointer, payloadWords: Long, dPageOff: Pointer,
                             dSide: Pointer, n: Long, layout: Int, dCodebook: Pointer,
                             dPageIds: Pointer, dOutSlots: Pointer, nIds: Long, dOut: Pointer,
                             nSlots: Long, check: Int): Int
  def hf_page_codebook_download(ctx: Pointer, dPayload: Pointer, payloadWords: Long,
                                dPageOff: Pointer, page: Long, hCode: Array[Short],
                                hLen: Array[Byte], mode: IntByReference,
                                entropy: IntByReference): Int
  def hf_chunk_codebook_download(ctx: Pointer, dChunkCodebooks: Pointer, chunk: Long,
                                 hCode: Array[Short], hLen: Array[Byte], mode: IntByReference,
                                 entropy: IntByReference): Int
  // container (stored form; no counterpart in the reference)
  def hf_container_plan(n: Long, layout: Int, codebookKind: Int, payloadWords: Long,
                        info: ContainerInfo): Int
  def hf_container_plan_placed(n: Long, layout: Int, codebookKind: Int, placement: Int,
                               payloadWords: Long, info: ContainerInfo): Int
  def hf_container_pack(ctx: Pointer, info: ContainerInfo, dCodebook: Pointer, dChunkOff: Pointer,
                        dSide: Pointer, dPayload: Pointer, dBlob: Pointer, blobCap: Long): Int
  def hf_container_pack_placed(ctx: Pointer, info: ContainerInfo, dCodebook: Pointer,
                               dChunkOff: Pointer, dChunkWords: Pointer, dSide: Pointer,
                               dPayload: Pointer, dBlob: Pointer, blobCap: Long): Int
  def hf_container_open(ctx: Pointer, dBlob: Pointer, blobBytes: Long, info: ContainerInfo): Int
  def hf_container_codebook(ctx: Pointer, dBlob: Pointer, info: ContainerInfo,
                            dCodebookOut: Pointer): Int
  // AdaptiveThreshold's register sets (adaptive_threshold.scala:41-79); out = {mode, threshold Q16,
  // depth limit, reserved: 4 x u32, then min_freq_threshold: u64}
  def hf_adaptive_threshold(entropyQ16: Int, depth: Int, wtWidth: Int, out: Pointer): Int
  // TreeQualityAnalyzer.io (tree_q_analyser.scala:11-19)
  def hf_quality(ctx: Pointer, hFreq: Array[Long], hLen: Array[Byte], depth: Int, wtWidth: Int,
                 avgLenQ8: ShortByReference, efficiency: ShortByReference,
                 ratio: ShortByReference): Int
}

@Structure.FieldOrder(Array("version", "layout", "codebook_kind", "chunk_syms", "max_code_len",
  "mode", "entropy_q16", "placement", "n", "nchunks", "payload_words", "off_codebook",
  "off_chunk_off", "off_side", "off_payload", "total_bytes", "off_chunk_words"))
class ContainerInfo extends Structure {
  var version, layout, codebook_kind, chunk_syms, max_code_len, mode, entropy_q16, placement: Int = 0
  var n, nchunks, payload_words, off_codebook, off_chunk_off, off_side, off_payload,
      total_bytes, off_chunk_words: Long = 0L
}

@Structure.FieldOrder(Array("encoding_mode", "tree_depth_limit", "min_freq_threshold"))
class EntropyConfig extends Structure {
  var encoding_mode: Int = 1; var tree_depth_limit: Int = 0; var min_freq_threshold: Long = 0L
}

object HuffHip { lazy val lib: HuffHip = Native.load("huffhip", classOf[HuffHip]) }

// HF_RESERVE_*, HF_LAYOUT_* and HF_MODE_* of include/huffhip.h
object Placement { val DENSE = 0; val SLOTTED = 1; val PAGED = 2 }
object Reserve {
  val ENCODE = 1; val CODEBOOKS = 2; val PLACED = 4; val ALL = 7
}
object Layout {
  val U8 = 0; val BF16_EXP = 1; val FP8 = 2; val FP16_EXP = 3; val FP8_E4M3_EXP = 4
  val FP4_E2M1_EXP = 5; val BITPLANE8 = 6
}
object Mode { val BYPASS = 0; val HUFFMAN = 1; val AUTO = 2; val BEST = 3 }

/** Software twin of the Chisel SymbolStat: same constructor, `freq_out` of one block. */
class SymbolStat(val sym_width: Int, val wtWidth: Int, val depth: Int)(implicit ctx: Pointer) {
  def apply(dData: Pointer, n: Long, layout: Int): Array[Long] = {
    val f = new Array[Long](depth)
    val rc = HuffHip.lib.hf_symbol_stat(ctx, dData, n, layout, depth, wtWidth, f)
    require(rc == 0, s"hf_symbol_stat: $rc"); f
  }
}

/** Software twin of SymbolSort (symbol_sort.scala:82-148): sorted counts and the rank -> symbol map. */
class SymbolSort(val depth: Int, val wtWidth: Int)(implicit ctx: Pointer) {
  def apply(freq: Array[Long]): (Array[Long], Array[Short]) = {
    val sorted = new Array[Long](depth); val rank2sym = new Array[Short](depth)
    val rc = HuffHip.lib.hf_symbol_sort(ctx, freq, depth, sorted, rank2sym)
    require(rc == 0, s"hf_symbol_sort: $rc"); (sorted, rank2sym)
  }
}

/** Software twin of ShannonEntropy (shannon_entropy.scala:11-118) with AdaptiveThreshold inside:
  * `entropy_out` (Q16.16, the RTL's arithmetic) and `compression_mode`. */
class ShannonEntropy(val depth: Int, val wtWidth: Int)(implicit ctx: Pointer) {
  def apply(freq: Array[Long]): (Int, Int) = {
    val e = new IntByReference(); val m = new IntByReference()
    val rc = HuffHip.lib.hf_entropy(ctx, freq, depth, e, m)
    require(rc == 0, s"hf_entropy: $rc"); (e.getValue, m.getValue)
  }
}

/** Software twin of AdaptiveThreshold (adaptive_threshold.scala:11-87): the register set chosen
  * for an entropy value, as (encoding_mode, compression_threshold Q16, tree_depth_limit,
  * min_freq_threshold).  Host arithmetic, needs no context. */
class AdaptiveThreshold(val depth: Int, val wtWidth: Int, val fracBits: Int = 16) {
  def apply(entropyQ16: Int): (Int, Int, Int, Long) = {
    val out = new com.sun.jna.Memory(24)
    val rc = HuffHip.lib.hf_adaptive_threshold(entropyQ16, depth, wtWidth, out)
    require(rc == 0, s"hf_adaptive_threshold: $rc")
    (out.getInt(0), out.getInt(4), out.getInt(8), out.getLong(16))
  }
}

/** What HuffmanTreeBuilder's ports hold when `done` rises (tree_build.scala:22-32); `depths` are
  * true leaf depths (the RTL's port of that name is never driven). */
case class HuffmanTree(root: Int, weights: Array[Long], lefts: Array[Short], rights: Array[Short],
                       depths: Array[Short])

/** Software twin of HuffmanTreeBuilder (tree_build.scala:17-140): merge priority (weight, index). */
class HuffmanTreeBuilder(val numLeaves: Int, val wtWidth: Int)(implicit ctx: Pointer) {
  def apply(freqs: Array[Long]): HuffmanTree = {
    val m = 2 * numLeaves - 1
    val w = new Array[Long](m); val l = new Array[Short](m); val r = new Array[Short](m)
    val d = new Array[Short](numLeaves); val root = new ShortByReference()
    val rc = HuffHip.lib.hf_tree_build(ctx, freqs, numLeaves, if (wtWidth <= 32) 32 else 64, w, l, r,
                                       root, d)
    require(rc == 0, s"hf_tree_build: $rc")
    HuffmanTree(root.getValue & 0xFFFF, w, l, r, d)
  }
}

/** Software twin of tree_adjust (tree_adjust.scala:6-127; lower-case name as there): the depth
  * limiter with assign-by-rank. */
class tree_adjust(val numLeaves: Int, val maxDepth: Int)(implicit ctx: Pointer) {
  def apply(depthsIn: Array[Short]): Array[Short] = {
    val out = new Array[Short](numLeaves)
    val rc = HuffHip.lib.hf_tree_adjust(ctx, depthsIn, numLeaves, maxDepth, out)
    require(rc == 0, s"hf_tree_adjust: $rc"); out
  }
}

/** Software twin of MemWrapper (mem_wrapper.scala:6-87): write port, read port, bulk read and
  * valid bits on the host; `toDevice` places the table where the encoder and decoder kernels
  * read it (dCodebook: HF_CODEBOOK_DEV_BYTES of device memory). */
class MemWrapper(val depth: Int = 256, val maxCodeLen: Int = 16, val addrWidth: Int = 8)
                (implicit ctx: Pointer) {
  private val code = new Array[Short](depth); private val len = new Array[Byte](depth)
  private val valid = new Array[Boolean](depth)
  def write(addr: Int, c: Int, l: Int): Unit = {
    code(addr) = c.toShort; len(addr) = l.toByte; valid(addr) = true
  }
  def read(addr: Int): (Int, Int, Boolean) = (code(addr) & 0xFFFF, len(addr) & 0xFF, valid(addr))
  def tableReady: Boolean = valid.forall(identity)
  def bulkRead: (Array[Short], Array[Byte], Boolean) = (code.clone, len.clone, tableReady)
  def flush(): Unit = java.util.Arrays.fill(valid, false)
  def toDevice(dCodebook: Pointer, mode: Int = 1): Unit = {
    require(depth == 256 && tableReady, "table_ready is low: write all 256 entries first")
    val rc = HuffHip.lib.hf_codebook_upload(ctx, code, len, mode, dCodebook)
    require(rc == 0, s"hf_codebook_upload: $rc")
  }
}

/** Software twin of TreeQualityAnalyzer (tree_q_analyser.scala:10-47):
  * (avg_length Q8, efficiency, compression_ratio) with the RTL's widths. */
class TreeQualityAnalyzer(val depth: Int, val wtWidth: Int) {
  def apply(freq: Array[Long], codeLength: Array[Byte]): (Int, Int, Int) = {
    val a = new ShortByReference(); val e = new ShortByReference(); val r = new ShortByReference()
    val rc = HuffHip.lib.hf_quality(Pointer.NULL, freq, codeLength, depth, wtWidth, a, e, r)
    require(rc == 0, s"hf_quality: $rc")
    (a.getValue & 0xFFFF, e.getValue & 0xFFFF, r.getValue & 0xFFFF)
  }
}

/** Software twin of EntropyAwareTreeBuilder (entropy_aware_tree_builder.scala:6-169). */
class EntropyAwareTreeBuilder(val depth: Int, val wtWidth: Int)(implicit ctx: Pointer) {
  def apply(freq: Array[Long], minFreqThreshold: Long = 0L): (Array[Short], Array[Byte]) = {
    val cfg = new EntropyConfig; cfg.min_freq_threshold = minFreqThreshold
    val code = new Array[Short](depth); val len = new Array[Byte](depth)
    val rc = HuffHip.lib.hf_codebook(ctx, freq, depth, cfg, code, len)
    require(rc == 0, s"hf_codebook: $rc"); (code, len)
  }
}

/** Software twin of EntropyAwareHuffmanSystem (top_encoder.scala:6-184) with the bit packer:
  * device tensor in, packed stream (payload, chunk offsets, side bytes) out. */
class EntropyAwareHuffmanSystem(val symWidth: Int = 8, val wtWidth: Int = 32, val depth: Int = 256,
                                val maxCodeLen: Int = 16)(implicit ctx: Pointer) {
  /** dFreq256: zeroed uint64[256]; dCodebook: HF_CODEBOOK_DEV_BYTES; returns the payload words. */
  def encode(dData: Pointer, n: Long, layout: Int, mode: Int, dFreq256: Pointer, dCodebook: Pointer,
             dSide: Pointer, dPayload: Pointer, payloadCapWords: Long, dChunkOff: Pointer): Long = {
    val lib = HuffHip.lib
    require(lib.hf_symbol_stat_dev(ctx, dData, n, layout, dFreq256) == 0)
    require(lib.hf_system_codebook_dev(ctx, dFreq256, mode, 0L, dCodebook) == 0)
    val words = new LongByReference()
    val rc = lib.hf_encode(ctx, dData, n, layout, dCodebook, dSide, dPayload, payloadCapWords,
                           dChunkOff, words)
    require(rc == 0, s"hf_encode: $rc"); words.getValue
  }
}

/** Software twin of ParrallelHuffmanDecoder (decoder/top_decoder.scala:6; spelling as there). */
class ParrallelHuffmanDecoder(val symWidth: Int = 8, val wtWidth: Int = 32, val depth: Int = 256,
                              val maxCodeLen: Int = 16)(implicit ctx: Pointer) {
  def decode(dPayload: Pointer, payloadWords: Long, dChunkOff: Pointer, dSide: Pointer, n: Long,
             layout: Int, dCodebook: Pointer, dOut: Pointer): Unit = {
    val rc = HuffHip.lib.hf_decode(ctx, dPayload, payloadWords, dChunkOff, dSide, n, layout,
                                   dCodebook, dOut, 1)
    require(rc == 0, s"hf_decode: $rc")
  }
}
