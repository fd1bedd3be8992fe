/*
 * hf_oracle.h -- CPU restatement of the liuyuxivapor/huffman encoder pipeline.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or calls this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Two layers, kept apart by prefix:
 *
 *   orc_aw_*   "as written": literal restatements of the reference modules that are sound
 *              (SymbolStat, SymbolSort, HuffmanTreeBuilder, the codeStart recurrence, the
 *              ShannonEntropy/AdaptiveThreshold arithmetic, the encoder lookup mux, the
 *              TreeQualityAnalyzer arithmetic).  Pinned by the reference's own known-answer
 *              tests where they exist (t_symbol_stat.scala, t_symbol_sort.scala) and by
 *              source reading elsewhere.
 *
 *   orc_rp_*   "repaired": what the reference intends but does not deliver (true leaf depths,
 *              a Kraft-correct length limiter, per-symbol canonical codes, the rank->symbol
 *              permutation, a packed bitstream and a decoder).  PARITY UNPINNED: the reference
 *              has no working counterpart, so these are pinned only by invariants
 *              (Kraft, max length, round trip) and by the golden vectors this oracle produced.
 *
 * All arithmetic is integer.  "N" is the alphabet size (`depth` / `numLeaves` in the
 * reference), "W" the counter width (`wtWidth`).
 */
#ifndef HF_ORACLE_H
#define HF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- stream geometry of the packed format (repaired layer; see orc_rp_encode) ---------- */
#define ORC_CHUNK_SYMS 16384u /* symbols per chunk = 64 lanes x 256 symbols             */
#define ORC_LANES 64u
#define ORC_GROUP 8u          /* consecutive symbols a lane takes per round              */
#define ORC_ROUNDS 32u        /* rounds per chunk                                        */

enum {
    ORC_LAYOUT_U8 = 0,
    ORC_LAYOUT_BF16_EXP = 1,
    ORC_LAYOUT_FP8 = 2,
    ORC_LAYOUT_FP16_EXP = 3,     /* symbol = 5 exponent bits, side field = sign | 10 mantissa bits */
    ORC_LAYOUT_FP8_E4M3_EXP = 4, /* symbol = sign | 4 exponent bits, side field = 3 mantissa bits */
    ORC_LAYOUT_FP4_E2M1_EXP = 5, /* a byte = two fp4: symbol = their sign | exponent fields, side = 2 mantissa bits */
    ORC_LAYOUT_BITPLANE8 = 6,    /* bytes transposed into bit planes, no side stream; n % 8 == 0 */
};
#define ORC_BITPLANE_BLOCK (8u * ORC_CHUNK_SYMS) /* bytes transposed together: 8 chunks */

/* ---- front end ------------------------------------------------------------------------- */
/* Split `n` elements into an 8-bit symbol stream and (BF16_EXP only) a raw side byte stream.
 * BF16_EXP: symbol = (w >> 7) & 0xFF (the exponent), side = ((w >> 8) & 0x80) | (w & 0x7F).
 * U8 / FP8: symbol = the byte, no side stream (side may be NULL). */
/* FP16_EXP / FP8_E4M3_EXP (sub-field layouts; reports/spec.md:22 lists FP16 / FP8 / FP4): the
 * symbol is the exponent field (with the sign for fp8), the side stream the dense little-endian
 * bit-packing of the remaining 11 / 3 bits per element in element order, orc_side_bytes() long.
 * FP4_E2M1_EXP: an element is a byte holding two fp4 values (s ee m, the first in the low
 * nibble); symbol = (see of the first) | (see of the second) << 3, side field = the two mantissa
 * bits (first in bit 0), 2 bytes per 8 elements.
 * BITPLANE8 (bitplane/bitplane.scala is an empty file, reports/spec.md:22 names bit-plane
 * compression): the bytes are taken in blocks of ORC_BITPLANE_BLOCK (the last one shorter, n a
 * multiple of 8); a block of m bytes is G = m / 8 groups of 8 consecutive bytes, and symbol
 * block + p * G + g holds bit p of each byte of group g, byte i of the group in bit i.  With
 * full blocks chunk 8k + p of the symbol stream is bit plane p of data chunks 8k .. 8k + 7, so
 * per-chunk codebooks see one plane each.  No side stream. */
uint64_t orc_side_bytes(uint64_t n, int layout);
void orc_symbolize(const void* data, size_t n, int layout, uint8_t* sym, uint8_t* side);
void orc_desymbolize(const uint8_t* sym, const uint8_t* side, size_t n, int layout, void* out);

/* ---- as written ------------------------------------------------------------------------ */
/* SymbolStat (symbol_stat.scala:25-49): freq[sym mod N] += 1, counters wrap at 2^W.
 * N must be a power of two (the RTL indexes with the low log2(N) bits). */
void orc_aw_symbol_stat(const uint8_t* sym, size_t n, int N, int W, uint64_t* freq);

/* SymbolSort (symbol_sort.scala:125-142): 2N passes of odd-even transposition, first pass on
 * odd pairs, swap when a < b (strict) => descending and stable.  `ids` (optional) rides along
 * with the values: the payload extension that carries symbol identities. */
void orc_aw_symbol_sort(const uint64_t* in, int N, uint64_t* sorted, uint16_t* ids);

/* HuffmanTreeBuilder (tree_build.scala:47-72, 90-131).  Arrays of 2N-1 entries.
 * `heights` (optional, 2N-1) is depthArr as the RTL computes it (subtree height); the RTL's
 * `depths` output is its first N entries, which are never written after init => all zero. */
void orc_aw_tree_build(const uint64_t* freqs, int N, int W, uint64_t* weights, uint16_t* lefts,
                       uint16_t* rights, uint16_t* root, uint16_t* heights);

/* codeStart recurrence (entropy_aware_tree_builder.scala:118-127): count[d] over d = 0..16,
 * codeStart[0] = 0, codeStart[d] = (codeStart[d-1] + count[d-1]) << 1 in 16 bits.
 * count[0] participates exactly as written. */
void orc_aw_code_start(const uint16_t* len, int N, uint16_t count[17], uint16_t code_start[17]);

/* The per-symbol code formula exactly as written (entropy_aware_tree_builder.scala:133-155):
 * code(i) = codeStart[len_i] + prefixSum[len_i].  Not prefix free; kept for the record. */
void orc_aw_codes_as_written(const uint16_t* len, int N, uint16_t* code);

/* ShannonEntropy (shannon_entropy.scala:52-54, 58-109) with the loop run exactly N times,
 * W <= 32.  Returns finalEntropy (Q16.16, 32 bits). */
uint32_t orc_aw_entropy_q16(const uint64_t* freq, int N, int W);

/* AdaptiveThreshold decision (adaptive_threshold.scala:26-28, 65-79): 1 = Huffman, 0 = bypass. */
int orc_aw_adaptive_mode(uint32_t entropy_q16);

/* The full register set AdaptiveThreshold selects (adaptive_threshold.scala:41-79); the RTL
 * exports only the mode.  Returns -1 where the Scala literal does not elaborate. */
int orc_aw_adaptive_config(uint32_t entropy_q16, int depth, int W, uint32_t* mode,
                           uint32_t* thr_q16, uint32_t* depth_limit, uint64_t* min_freq);

/* EntropyAwareTreeBuilder.sFilter (entropy_aware_tree_builder.scala:61-66). */
void orc_aw_filter(const uint64_t* freq, int N, uint64_t min_freq_threshold, uint64_t* out);

/* Encoder lookup mux (top_encoder.scala:124-162): one {code,length,mode} triple per symbol.
 * mode 0 => {sym, sym_width, 0}. out is n entries of 4 bytes: code lo, code hi, length, mode. */
void orc_aw_encode_pairs(const uint8_t* sym, size_t n, const uint16_t* code, const uint8_t* len,
                         int mode, int sym_width, uint8_t* out4);

/* TreeQualityAnalyzer (tree_q_analyser.scala:21-47).  Returns 0, or -1 if a divisor is zero
 * (the RTL result is then undefined). */
int orc_aw_quality(const uint64_t* freq, const uint8_t* len, int N, int W, uint16_t* avg_len_q8,
                   uint16_t* efficiency, uint16_t* compression_ratio);

/* ---- repaired -------------------------------------------------------------------------- */
/* True leaf depths from lefts/rights/root (repairs tree_build.scala:81). */
void orc_rp_leaf_depths(const uint16_t* lefts, const uint16_t* rights, int root, int N,
                        uint16_t* depths);

/* Length limiter keeping tree_adjust.scala's four phases (BuildHist, Migrate, FillHoles,
 * Assign-by-rank, tree_adjust.scala:44-119) with Kraft-correct arithmetic.
 * Returns the number of Migrate steps taken, or -1 if 2^lmax < N. */
int orc_rp_tree_adjust(const uint16_t* depths_in, int N, int lmax, uint16_t* depths_out);

/* Canonical codes in index order: code(i) = codeStart[len_i] + #{j < i : len_j == len_i},
 * count[0] forced to 0 (the intent at entropy_aware_tree_builder.scala:105-111). */
void orc_rp_canonical_codes(const uint16_t* len, int N, uint16_t* code);

/* EntropyAwareTreeBuilder, repaired, at module level: indices are the caller's indices
 * (filter -> tree -> depths -> adjust(lmax) -> canonical codes).  Returns 0 / -1. */
int orc_rp_codebook(const uint64_t* freq, int N, int W, uint64_t min_freq_threshold, int lmax,
                    uint16_t* code, uint8_t* len);

/* The whole front half of EntropyAwareHuffmanSystem, repaired: per-symbol counts ->
 * (count desc, symbol asc) sort -> codebook over ranks -> rank->symbol permutation.
 * Outputs are indexed by raw symbol. */
int orc_rp_system_codebook(const uint64_t* freq_by_symbol, int N, int W,
                           uint64_t min_freq_threshold, int lmax, uint16_t* code, uint8_t* len);

/* Entropy for counters wider than 32 bits: if the total needs more than 32 bits every count
 * is shifted right by the excess first, then orc_aw_entropy_q16(W = 32) applies. */
uint32_t orc_rp_entropy_q16_u64(const uint64_t* freq, int N);

/* Packed stream.  Chunks of ORC_CHUNK_SYMS symbols; inside a chunk lane l of 64 owns the
 * symbol groups g = r*64 + l (r = round), 8 symbols each, and concatenates their codes MSB
 * first into a private bit string cut into 32-bit words.  The chunk payload is those words
 * of all 64 lanes in the order a lock-step decoder asks for them: before each pair of symbols
 * every lane holding fewer than 32 buffered bits takes the next word, lanes in ascending order.
 * chunk_off[c] = first payload word of chunk c, chunk_off[nchunks] = total words.
 * Returns total words written (payload may be NULL to size only). */
uint64_t orc_rp_encode(const uint8_t* sym, size_t n, const uint16_t* code, const uint8_t* len,
                       uint32_t* payload, uint32_t* chunk_off);
/* Returns 0, or -1 if the payload does not decode (ran past a chunk / no matching code). */
int orc_rp_decode(const uint32_t* payload, const uint32_t* chunk_off, size_t n,
                  const uint16_t* code, const uint8_t* len, uint8_t* sym_out);

/* Per-block codebooks: every chunk of ORC_CHUNK_SYMS symbols is a block of its own (the
 * reference clears its counters on each `start`, symbol_stat.scala:26-33): SymbolStat, the
 * repaired system codebook with Huffman forced, and the packed stream, chunk by chunk.
 * code / len: nchunks x 256 entries by raw symbol.  Returns total words (payload may be NULL). */
uint64_t orc_rp_encode_blocks(const uint8_t* sym, size_t n, uint64_t min_freq_threshold,
                              uint16_t* code, uint8_t* len, uint32_t* payload,
                              uint32_t* chunk_off);
/* Slotted placement of a per-block stream (an alternative to the back-to-back one above, for
 * encoders that want every chunk's place before they start): chunk c gets a slot of
 * ORC_SLOT_WORDS(B_c) = ceil(B_c / 32) + 126 words, B_c = the bits of the chunk's codes
 * (sum over its symbols of len[c][symbol]) -- no chunk needs more: a lane asks for a word only
 * while it holds fewer than 32 bits, so it ends with at most 62 unread ones and takes at most
 * floor((T + 62) / 32) words for the T bits of its codes; 64 lanes: floor(B_c / 32) + 124.  The chunk's words are the
 * same words in the same order; slot_off[c] = sum of the earlier slots, slot_off[nchunks] = their
 * total, words[c] = the words the chunk really has, and the rest of each slot is zero.
 * len_stride: 256 for per-block codebooks (len = nchunks x 256), 0 for one codebook (len = 256
 * entries; the back-to-back stream is then orc_rp_encode's).
 * `payload` / `chunk_off`: the back-to-back stream of orc_rp_encode_blocks.  Returns the total
 * slot words (slot_payload may be NULL to size only), UINT64_MAX if a chunk were larger than its
 * slot. */
#define ORC_SLOT_WORDS(bits) (((bits) + 31u) / 32u + 126u)
uint64_t orc_rp_slot_blocks(const uint8_t* sym, size_t n, const uint8_t* len, size_t len_stride,
                            const uint32_t* payload, const uint32_t* chunk_off,
                            uint32_t* slot_payload, uint32_t* slot_off, uint32_t* words);
int orc_rp_decode_blocks(const uint32_t* payload, const uint32_t* chunk_off, size_t n,
                         const uint16_t* code, const uint8_t* len, uint8_t* sym_out);

/* Paged stream (format version 2): the second stored form of a per-block stream, in which every
 * page of ORC_CHUNK_SYMS symbols carries its own codebook record, has a size that follows from its
 * histogram and codebook alone, and is decoded by four sub-decoders at once (the reference's only
 * decoder intent, reports/figs/decoder.svg: a scheduler feeding parallel sub-decoders; its module,
 * decoder/top_decoder.scala:6-10, is empty).  Layout: the comment above orc_rp_page_head in
 * hf_oracle.c.  PARITY UNPINNED, as the packed stream above. */
/* ranks the record lists explicitly: the smallest p such that rank2sym[p .. 255] ascends */
uint32_t orc_rp_page_head(const uint8_t* rank2sym);
/* words of a page whose record lists `head` ranks and whose codes take `bits` bits */
uint64_t orc_rp_page_words(uint32_t head, uint64_t bits);
uint64_t orc_rp_pages_bound_words(size_t n);
uint64_t orc_rp_encode_page(const uint8_t* sym, size_t m, const uint16_t* code, const uint8_t* len,
                            const uint8_t* rank2sym, const uint8_t* len_by_rank, uint32_t mode,
                            uint32_t entropy_q16, uint32_t* out);
/* mode_req: 0 bypass, 1 Huffman forced, 2 the entropy decision, 3 best (orc_rp_best_mode) */
uint64_t orc_rp_encode_pages(const uint8_t* sym, size_t n, uint64_t min_freq_threshold, int mode_req,
                             uint16_t* code_out, uint8_t* len_out, uint32_t* payload,
                             uint32_t* page_off);
int orc_rp_page_codebook(const uint32_t* page, uint64_t words, uint16_t* code, uint8_t* len,
                         uint32_t* mode, uint32_t* entropy_q16);
int orc_rp_decode_pages(const uint32_t* payload, const uint32_t* page_off, size_t n,
                        uint8_t* sym_out);
/* paged stream with one codebook for the stream: pages without a record (word 0 = B << 10 | 1 << 29) */
uint64_t orc_rp_shared_page_words(uint64_t bits);
uint64_t orc_rp_encode_pages_shared(const uint8_t* sym, size_t n, const uint16_t* code,
                                    const uint8_t* len, uint32_t* payload, uint32_t* page_off);
int orc_rp_decode_pages_shared(const uint32_t* payload, const uint32_t* page_off, size_t n,
                               const uint16_t* code, const uint8_t* len, uint8_t* sym_out);

size_t orc_num_chunks(size_t n);
/* Upper bound on payload words for n symbols under any codebook with lengths <= 16. */
uint64_t orc_payload_bound_words(size_t n);

/* Mode choice "best": 1 (Huffman) if the table saves bits on the counts it was built from,
 * sum(freq x len) < 8 x sum(freq), else 0 (bypass).  Not in the reference, whose entropy
 * decision (shannon_entropy.scala:78-91) is no measure of compressibility. */
int orc_rp_best_mode(const uint64_t* freq_by_symbol, const uint8_t* len_by_symbol, int N);

/* Container, wire format version 1 (the reference defines no stored form, top_encoder.scala:17-21
 * ends at the {code, length, mode} beat; the format is this build's): a 128-byte little-endian
 * header (magic "HUFHIP1\0", version, header size, layout, codebook kind, chunk symbols, maximum
 * code length, mode, entropy, n, nchunks, payload words, four section offsets, total size), then
 * the codebook section, chunk offsets, side bytes (bf16 exponent layout only) and payload, each
 * starting on a 16-byte boundary and zero padded; the payload section holds at least 16 bytes.
 * kind 0: one codebook, code u16[256] + length u8[256].  kind 1: one 296-byte record per chunk:
 * rank2sym u8[256], u16[16] number of ranks per code length 1..16, u32 mode, u32 entropy. */
uint64_t orc_container_bytes(uint64_t n, int layout, int kind, uint64_t payload_words);
/* Writes the container into out (orc_container_bytes() bytes) and returns its size.  kind 0 uses
 * mode, entropy_q16, code, len; kind 1 uses rank2sym, len_by_rank (nchunks x 256 each), modes,
 * entropies (nchunks each). */
uint64_t orc_container_pack(uint8_t* out, uint64_t n, int layout, int kind, uint64_t payload_words,
                            uint32_t mode, uint32_t entropy_q16, const uint16_t* code,
                            const uint8_t* len, const uint8_t* rank2sym,
                            const uint8_t* len_by_rank, const uint32_t* modes,
                            const uint32_t* entropies, const uint32_t* chunk_off,
                            const uint8_t* side, const uint32_t* payload);

/* Container, wire format version 2: version 1 plus a placement field (header byte 104: 1 slotted,
 * 2 paged) and, for slotted streams, the words array (its offset at header byte 112, the section
 * between the chunk offsets and the side bytes).  A paged stream has no codebook section (offset 0):
 * its pages carry the records.  placement 0 writes the version-1 container. */
uint64_t orc_container2_bytes(uint64_t n, int layout, int kind, int placement, uint64_t payload_words);
uint64_t orc_container2_pack(uint8_t* out, uint64_t n, int layout, int kind, int placement,
                             uint64_t payload_words, uint32_t mode, uint32_t entropy_q16,
                             const uint16_t* code, const uint8_t* len, const uint8_t* rank2sym,
                             const uint8_t* len_by_rank, const uint32_t* modes,
                             const uint32_t* entropies, const uint32_t* chunk_off,
                             const uint32_t* chunk_words, const uint8_t* side,
                             const uint32_t* payload);

#ifdef __cplusplus
}
#endif
#endif
