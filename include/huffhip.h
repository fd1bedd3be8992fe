/*
 * huffhip.h -- C ABI of libhuffhip.so, the MI355X (gfx950) Huffman codec.
 *
 * The reference (liuyuxivapor/huffman) is Chisel RTL and has no software FFI; its interface
 * contract is the IO bundle of each hardware module.  Every entry point below replaces one of
 * those module boundaries and cites it.  Conventions shared by all calls:
 *
 *   - one opaque hf_ctx per (device, stream); a ctx is not thread safe, distinct ctxs are;
 *   - every call returns 0 or a negative HF_ERR_* code and never throws; the RTL itself has no
 *     error outputs (counters wrap silently), so anything it would silently mangle is an error;
 *   - the caller owns all buffers.  Bulk data (symbols, payload, side bytes, chunk offsets)
 *     are DEVICE pointers, 16-byte aligned; tables of at most a few KiB are HOST pointers
 *     unless the name ends in _dev;
 *   - calls taking host tables synchronise the ctx stream before returning; calls whose
 *     outputs are all device-side only enqueue work (ordering = stream order);
 *   - N ("depth"/"numLeaves" in the reference) is the alphabet size, 2 <= N <= 256.
 */
#ifndef HUFFHIP_H
#define HUFFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HF_ABI_VERSION 3

/* error codes */
#define HF_OK 0
#define HF_ERR_ARG (-1)      /* null pointer, bad size, bad enum                          */
#define HF_ERR_ALIGN (-2)    /* device pointer not 16-byte aligned                         */
#define HF_ERR_RANGE (-3)    /* value outside what the arithmetic is defined for           */
#define HF_ERR_HIP (-4)      /* a HIP runtime call failed (see hf_last_hip_error)          */
#define HF_ERR_OVERFLOW (-5) /* payload buffer too small                                   */
#define HF_ERR_TIMEOUT (-6)  /* an in-kernel bounded wait ran out                          */
#define HF_ERR_CORRUPT (-7)  /* payload / offsets inconsistent                             */
#define HF_ERR_NOMEM (-8)
#define HF_ERR_RCCL (-9)     /* RCCL could not be loaded, or the collective failed (the
                                ncclResult_t is left in hf_last_hip_error)                  */

/* element layouts accepted by the front ends */
#define HF_LAYOUT_U8 0       /* symbol = byte                                              */
#define HF_LAYOUT_BF16_EXP 1 /* symbol = bf16 exponent (w>>7)&0xFF, side byte = sign|mantissa */
#define HF_LAYOUT_FP8 2      /* fp8 E4M3/E5M2 byte taken whole, same as U8                  */
/* sub-field layouts: split by hf_fields_split into a symbol stream (coded as HF_LAYOUT_U8) and a
 * packed side stream, joined again by hf_fields_merge; the container carries them as such */
#define HF_LAYOUT_FP16_EXP 3     /* fp16: symbol = 5 exponent bits, side field = sign | mantissa */
#define HF_LAYOUT_FP8_E4M3_EXP 4 /* fp8 E4M3: symbol = sign | exponent, side field = 3 mantissa bits */
#define HF_LAYOUT_FP4_E2M1_EXP 5 /* a byte of two fp4 E2M1: symbol = their sign | exponent fields, side
                                  * field = the 2 mantissa bits */
#define HF_LAYOUT_BITPLANE8 6    /* bytes transposed into bit planes, HF_BITPLANE_BLOCK bytes at a
                                  * time; no side stream; n must be a multiple of 8 */
#define HF_BITPLANE_BLOCK 131072u /* 8 chunks: with full blocks chunk 8k + p of the symbol stream is
                                   * bit plane p of data chunks 8k .. 8k + 7 */

/* packed-stream geometry */
#define HF_CHUNK_SYMS 16384u /* symbols per chunk (one wavefront: 64 lanes x 256 symbols)  */
#define HF_MAX_CODE_LEN 16

/* compression_mode values (top_encoder.scala:20, adaptive_threshold.scala:18) */
#define HF_MODE_BYPASS 0
#define HF_MODE_HUFFMAN 1
#define HF_MODE_AUTO 2 /* let the entropy decision choose (system calls only) */
#define HF_MODE_BEST 3 /* build the Huffman table, keep it only if sum(count x length) is below
                          8 bits per symbol on the counts it was built from, else bypass (system
                          calls only; the as-written entropy decision of HF_MODE_AUTO is no
                          measure of compressibility, see shannon_entropy.scala:78-91) */

typedef struct hf_ctx hf_ctx;

/* Device-resident codebook: encoder table, decoder lookup tables, mode.  Opaque; size and
 * alignment are fixed so callers can place it in their own device allocations. */
#define HF_CODEBOOK_DEV_BYTES 36864u

/* entropy_config bundle of EntropyAwareTreeBuilder (entropy_aware_tree_builder.scala:9-13).
 * As in the RTL only min_freq_threshold is honoured by the tree stage; tree_depth_limit = 0
 * selects the RTL's hard-wired 16 (entropy_aware_tree_builder.scala:23). */
typedef struct hf_entropy_config {
    uint32_t encoding_mode;
    uint32_t tree_depth_limit;
    uint64_t min_freq_threshold;
} hf_entropy_config;

/* one beat of EntropyAwareHuffmanSystem.io.encoded_out (top_encoder.scala:17-21) */
typedef struct hf_code_beat {
    uint16_t code;
    uint8_t length;
    uint8_t compression_mode;
} hf_code_beat;

/* ---- context ----------------------------------------------------------------------------- */
int hf_abi_version(void);
/* `stream` is a hipStream_t (NULL = the device's default stream). */
int hf_ctx_create(hf_ctx** out, int device, void* stream);
int hf_ctx_destroy(hf_ctx* ctx);
/* `flush` input of every module (symbol_stat.scala:51-57, symbol_sort.scala:107-110,
 * entropy_aware_tree_builder.scala:166-168, top_encoder.scala:178-183): drop any in-flight
 * block state; here: wait for the stream and clear the ctx's accumulators and error word. */
int hf_flush(hf_ctx* ctx);
/* Wait for the stream.  In-kernel errors (HF_ERR_OVERFLOW, HF_ERR_CORRUPT, HF_ERR_TIMEOUT) are
 * sticky: a call that only enqueues work leaves them in the ctx, and the next call that
 * synchronises -- hf_sync, or an encode / decode call asked to report -- returns and clears them. */
int hf_sync(hf_ctx* ctx);
const char* hf_error_string(int code);
int hf_last_hip_error(hf_ctx* ctx);
/* The context's device workspaces are sized on demand: the tile descriptors and the ~100 MB spill
 * area of hf_encode / hf_encode_chunks (HF_RESERVE_ENCODE), the 68 MB batch scratch of
 * hf_system_codebook_chunks_dev (HF_RESERVE_CODEBOOKS), the offset-scan workspace of the slotted
 * and paged encoders and the page descriptors of hf_encode_pages_shared without counts
 * (HF_RESERVE_PLACED).  A call that needs one that is missing or too small
 * synchronises the stream, frees and allocates -- which a stream under capture, or a loop that
 * must not allocate, cannot have.  hf_ctx_reserve makes the named workspaces now, sized for streams
 * of up to max_n elements; afterwards every enqueue-only call on the context with n <= max_n
 * allocates, frees and synchronises nothing, whatever was called before.  No counterpart in the
 * reference (its modules own their registers). */
#define HF_RESERVE_ENCODE 1u
#define HF_RESERVE_CODEBOOKS 2u
#define HF_RESERVE_PLACED 4u
#define HF_RESERVE_ALL 7u
int hf_ctx_reserve(hf_ctx* ctx, uint64_t max_n, unsigned flags);
/* Clear nwords 32-bit words at d_ptr (4-byte aligned) on the context's stream with a kernel: the
 * clear to use in front of the accumulating calls (hf_symbol_stat_dev, hf_symbol_stat_fold_dev)
 * in a sequence that is captured into a HIP graph.  Small hipMemsetAsync nodes captured inside
 * this library's pipelines were seen to replay with another node's fill pattern on ROCm 7.2; the
 * minimal program tools/memset_graph_repro.hip (two memset nodes and a kernel) replays correctly
 * there, so the cause is not pinned down, and a kernel node avoids the question.  No counterpart
 * in the reference (its counters are cleared by the flush port, symbol_stat.scala:41-46).
 * Enqueue only. */
int hf_zero_dev(hf_ctx* ctx, void* d_ptr, uint64_t nwords);

/* ---- SymbolStat (symbol_stat.scala:6-58) -------------------------------------------------- */
/* start + stream of n beats + done: histogram of one block.  freq[s mod N] counts; the
 * internal counters are 64 bit, the returned values are reduced mod 2^wt_width
 * (wt_width in 1..64) as the RTL's W-bit registers would hold them.  N must be a power of 2. */
int hf_symbol_stat(hf_ctx* ctx, const void* d_data, uint64_t n, int layout, int N, int wt_width,
                   uint64_t* h_freq);
/* Same block, counts left on the device: d_freq256 (uint64[256], by raw symbol) is
 * ACCUMULATED into (zero it for a fresh block); no synchronisation. */
int hf_symbol_stat_dev(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                       uint64_t* d_freq256);

/* The one exchange step of a tensor sharded over several GPUs (one context per GPU, every shard
 * counted by hf_symbol_stat_dev on its own GPU): the 256 counts are summed over all ranks of the
 * communicator, in place, on the context's stream -- one ncclAllReduce(sum, uint64, 256 elements,
 * 2 KiB: latency bound, nowhere near a link's bandwidth).  Every rank then runs
 * hf_system_codebook_dev on the same sums and gets the same tables; encode and decode stay local.
 * The reference counts one stream in one module (symbol_stat.scala:34-38); this is that module's
 * counter array summed across devices.  nccl_comm is an ncclComm_t of the RCCL the process uses
 * (librccl.so.1, loaded on the first call -- libhuffhip.so itself does not link it; HF_RCCL_LIB
 * names another file); with several communicators driven by one thread the calls go between
 * ncclGroupStart and ncclGroupEnd as usual.  Enqueue only. */
int hf_symbol_stat_allreduce(hf_ctx* ctx, void* nccl_comm, uint64_t* d_freq256);

/* ---- sub-field front ends -------------------------------------------------------------------
 * The reference's spec lists FP16 / FP8 / FP4 operands (reports/spec.md:22) but its RTL takes 8-bit
 * symbols only (symbol_stat.scala:6-13) and its bit-plane package is an empty file
 * (bitplane/bitplane.scala:1-4).  These two calls are that front end: hf_fields_split writes one
 * symbol byte per element (the exponent field: 5 bits of an fp16 value; sign and exponent of an
 * fp8 E4M3 value; the two sign-and-exponent fields of a byte of two fp4 E2M1 values) to d_sym and
 * the remaining bits (sign and mantissa, 11 bits; mantissa, 3 bits; the two mantissa bits) to
 * d_side, bit-packed little endian in element order -- 11, 3 or 2 bytes per 8 elements,
 * hf_fields_side_bytes(n, layout) in all.  The symbol stream then goes through SymbolStat, the
 * codebook and the encoder as HF_LAYOUT_U8; hf_fields_merge undoes the split after decoding.
 * HF_LAYOUT_BITPLANE8 goes through the same two calls: d_sym receives the n bytes transposed into
 * bit planes (a block of m <= HF_BITPLANE_BLOCK bytes is G = m / 8 groups of 8 consecutive bytes;
 * symbol block + p * G + g holds bit p of each byte of group g, byte i of the group in bit i),
 * d_side is not used (NULL) and n must be a multiple of 8 (HF_ERR_ARG otherwise).
 * Device pointers, 16-byte aligned; enqueue only. */
uint64_t hf_fields_side_bytes(uint64_t n, int layout);
int hf_fields_split(hf_ctx* ctx, const void* d_data, uint64_t n, int layout, uint8_t* d_sym,
                    uint8_t* d_side);
int hf_fields_merge(hf_ctx* ctx, const uint8_t* d_sym, const uint8_t* d_side, uint64_t n, int layout,
                    void* d_out);

/* ---- SymbolSort (symbol_sort.scala:82-148) ------------------------------------------------ */
/* sorted = counts in descending order (stable).  rank2sym (optional) carries the symbol
 * identities the RTL drops: rank2sym[p] = index of the p-th largest count,
 * total order (count desc, index asc). */
int hf_symbol_sort(hf_ctx* ctx, const uint64_t* h_freq, int N, uint64_t* h_sorted,
                   uint16_t* h_rank2sym);

/* ---- ShannonEntropy + AdaptiveThreshold (shannon_entropy.scala:11-118,
 *      adaptive_threshold.scala:11-87) ------------------------------------------------------- */
/* entropy_out (Q16.16) with the RTL's arithmetic at W = 32 and the loop run N times;
 * compression_mode = 1 (Huffman) / 0 (bypass).  Totals above 32 bits are scaled down by a
 * right shift of every count first. */
int hf_entropy(hf_ctx* ctx, const uint64_t* h_freq, int N, uint32_t* entropy_q16,
               int* compression_mode);

/* The whole parameter set AdaptiveThreshold keeps in registers for an entropy value
 * (adaptive_threshold.scala:41-79): the RTL drives only encoding_mode out, the other three sit
 * behind commented-out ports (:17-22, :82-86).  Values as the registers hold them:
 * compression_threshold in Q0.16 (0.3 / 0.6 / 0.9), tree_depth_limit = depth/4, depth/2, depth/8
 * truncated to log2ceil(depth) bits, min_freq_threshold = 1, 2, 2^(wt_width-2) truncated to
 * wt_width bits, for entropy below 0.5, in between, above 6.0 bits.  Host arithmetic, no
 * context.  depth in 2..256, wt_width in 2..64. */
typedef struct hf_adaptive_config {
    uint32_t encoding_mode;
    uint32_t compression_threshold_q16;
    uint32_t tree_depth_limit;
    uint32_t reserved;
    uint64_t min_freq_threshold;
} hf_adaptive_config;
int hf_adaptive_threshold(uint32_t entropy_q16, int depth, int wt_width, hf_adaptive_config* out);

/* ---- HuffmanTreeBuilder (tree_build.scala:17-140) ----------------------------------------- */
/* freqs[N] in the caller's index order (the system passes them sorted descending).
 * weights/lefts/rights have 2N-1 entries, entries below N as the RTL initialises them
 * (weight = freq, children 0).  Merge priority (weight, index) as findTwoMin
 * (tree_build.scala:47-72).  depths[N] are true leaf depths (the RTL's `depths` port is dead).
 * wt_width: 32 or 64.  The sum of all freqs must stay below 2^wt_width - 1 (the RTL wraps;
 * this call returns HF_ERR_RANGE instead). */
int hf_tree_build(hf_ctx* ctx, const uint64_t* h_freqs, int N, int wt_width, uint64_t* h_weights,
                  uint16_t* h_lefts, uint16_t* h_rights, uint16_t* h_root, uint16_t* h_depths);

/* ---- tree_adjust (tree_adjust.scala:6-127) ------------------------------------------------- */
/* Length limiter: BuildHist / Migrate / FillHoles / Assign-by-index with Kraft-correct moves.
 * max_depth in 1..16 and 2^max_depth >= N. */
int hf_tree_adjust(hf_ctx* ctx, const uint16_t* h_depths_in, int N, int max_depth,
                   uint16_t* h_depths_out);

/* ---- EntropyAwareTreeBuilder (entropy_aware_tree_builder.scala:6-169) ---------------------- */
/* freq_in[N] -> code_out[N], length_out[N]; indices are the caller's (filter, tree, adjust,
 * canonical codes in index order). */
int hf_codebook(hf_ctx* ctx, const uint64_t* h_freq, int N, const hf_entropy_config* cfg,
                uint16_t* h_code, uint8_t* h_len);

/* ---- the front half of EntropyAwareHuffmanSystem (top_encoder.scala:29-139) ---------------- */
/* Device pipeline over a 256-bin histogram by raw symbol: sort -> entropy decision ->
 * codebook over ranks -> rank->symbol scatter -> device codebook with decode tables.
 * mode: HF_MODE_BYPASS / HF_MODE_HUFFMAN force the choice, HF_MODE_AUTO follows the entropy
 * decision.  No synchronisation. */
int hf_system_codebook_dev(hf_ctx* ctx, const uint64_t* d_freq256, int mode,
                           uint64_t min_freq_threshold, void* d_codebook);

/* The same front half fed with ONE block of symbols instead of its counts, in one launch
 * (top_encoder.scala:29-41 puts SymbolStat in front of the sort in the same module): counts the
 * n <= HF_BLOCK_MAX_SYMS elements at d_data, leaves the counts in d_freq256 (overwritten, not
 * added to) and builds the codebook from them.  For small blocks the two launches of
 * hf_symbol_stat_dev + hf_system_codebook_dev cost more than the work; larger inputs take those.
 * No synchronisation. */
#define HF_BLOCK_MAX_SYMS 65536u
int hf_system_codebook_block_dev(hf_ctx* ctx, const void* d_data, uint64_t n, int layout, int mode,
                                 uint64_t min_freq_threshold, uint64_t* d_freq256,
                                 void* d_codebook);

/* ---- MemWrapper (mem_wrapper.scala:6-87): the codebook store ------------------------------- */
/* write port, all entries: build a device codebook from host tables indexed by symbol. */
int hf_codebook_upload(hf_ctx* ctx, const uint16_t* h_code, const uint8_t* h_len, int mode,
                       void* d_codebook);
/* bulk read port: code_table / length_table / mode / entropy of a device codebook. */
int hf_codebook_download(hf_ctx* ctx, const void* d_codebook, uint16_t* h_code, uint8_t* h_len,
                         int* mode, uint32_t* entropy_q16);

/* ---- encoder (top_encoder.scala:124-162) --------------------------------------------------- */
/* One {code, length, compression_mode} beat per symbol, unpacked, as the RTL emits them. */
int hf_encode_pairs(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                    const void* d_codebook, hf_code_beat* d_beats);

/* Packed stream.  d_side (BF16_EXP only) receives n raw side bytes.  d_payload receives the
 * payload words, d_chunk_off[c] the first word of chunk c and d_chunk_off[nchunks] the total
 * (nchunks = ceil(n / HF_CHUNK_SYMS)).  payload_cap_words bounds what may be written.
 * h_total_words == NULL: enqueue only.  Otherwise synchronise, return the total and report
 * in-kernel errors. */
int hf_encode(hf_ctx* ctx, const void* d_data, uint64_t n, int layout, const void* d_codebook,
              uint8_t* d_side, uint32_t* d_payload, uint64_t payload_cap_words,
              uint32_t* d_chunk_off, uint64_t* h_total_words);
uint64_t hf_num_chunks(uint64_t n);
/* words of a chunk's slot in a slotted stream (hf_encode_chunks_slotted), from the bits of its codes */
#define HF_SLOT_WORDS(bits) (((bits) + 31u) / 32u + 126u)
/* worst case over all codebooks with lengths <= 16 */
uint64_t hf_payload_bound_words(uint64_t n);

/* ---- ParrallelHuffmanDecoder (decoder/top_decoder.scala:6-10, constructor only) ------------ */
/* Rebuilds n elements of `layout` into d_out.  payload_words = size of the d_payload buffer,
 * at least 4 (offsets pointing past it are reported, never followed).  `check` != 0:
 * synchronise and report in-kernel errors. */
int hf_decode(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
              const uint32_t* d_chunk_off, const uint8_t* d_side, uint64_t n, int layout,
              const void* d_codebook, void* d_out, int check);

/* ---- per-block codebooks (one EntropyAwareTreeBuilder run per 16384-symbol chunk) ---------- *
 * The reference builds one codebook per block it is fed (symbol_stat.scala:26-33 clears the
 * counters on every `start`); for KV-cache pages a block is one chunk.  All four calls are
 * device to device and only enqueue work unless stated.                                       */
#define HF_CHUNK_CODEBOOK_BYTES 1552u /* one record per chunk, opaque */
/* SymbolStat per chunk: d_freq_chunks = uint32[nchunks][256], overwritten. */
int hf_symbol_stat_chunks_dev(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                              uint32_t* d_freq_chunks);
/* SymbolStat per chunk AND of the whole stream from one read of the data: d_freq_chunks as
 * above, and the stream's 256 counts ADDED into d_freq256 (uint64[256]; may be NULL) -- what a
 * stream with one codebook and pages sized from their own counts needs (hf_encode_pages_shared,
 * hf_encode_slotted).  The reference counts one stream in one module (symbol_stat.scala:34-38). */
int hf_symbol_stat_pages_dev(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                             uint32_t* d_freq_chunks, uint64_t* d_freq256);
/* sort -> entropy decision -> tree -> adjust -> canonical codes -> rank->symbol scatter, once
 * per chunk; d_chunk_codebooks = nchunks records of HF_CHUNK_CODEBOOK_BYTES. */
int hf_system_codebook_chunks_dev(hf_ctx* ctx, const uint32_t* d_freq_chunks, uint64_t nchunks,
                                  int mode, uint64_t min_freq_threshold, void* d_chunk_codebooks);
/* hf_system_codebook_chunks_dev for callers that cannot allocate: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.
 * The scratch the three codebook kernels pass a batch through sits in a static device array of
 * HF_STATIC_CB_SLOTS slices of HF_STATIC_CB_MAX_CHUNKS x 1040 bytes (16.25 MiB in all, held by
 * every GPU the library runs on until the process ends), so the call allocates and frees nothing;
 * same arguments, same records, bit for bit.  nchunks above HF_STATIC_CB_MAX_CHUNKS is refused
 * with HF_ERR_RANGE.  A context takes one slice of its GPU's array at its first call here (host
 * bookkeeping only) and gives it back in hf_ctx_destroy, so contexts on different streams of one
 * GPU never share scratch and calls of one context are ordered by its stream; the call of a
 * context that finds all HF_STATIC_CB_SLOTS slices of its GPU taken returns HF_ERR_NOMEM and
 * launches nothing. */
#define HF_STATIC_CB_MAX_CHUNKS 4096u
#define HF_STATIC_CB_SLOTS 4u
int hf_system_codebook_chunks_static(hf_ctx* ctx, const uint32_t* d_freq_chunks, uint64_t nchunks,
                                     int mode, uint64_t min_freq_threshold,
                                     void* d_chunk_codebooks);
/* as hf_encode, chunk c coded with record c */
int hf_encode_chunks(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                     const void* d_chunk_codebooks, uint8_t* d_side, uint32_t* d_payload,
                     uint64_t payload_cap_words, uint32_t* d_chunk_off, uint64_t* h_total_words);
/* as hf_decode.  d_chunk_ids == NULL: every chunk to its own place in d_out (n elements).
 * Otherwise decode-on-read: the n_ids listed chunks, in that order, to d_out + k * 16384
 * elements (all listed chunks must be full chunks; `n` still describes the encoded stream).
 * A chunk whose record is not a canonical order (lengths above 16, decreasing with the rank, or
 * filling more than the code space), whose offsets are out of order or past the payload, or whose
 * id is out of range is left undecoded and HF_ERR_CORRUPT is returned (with check != 0); the
 * other chunks of the call are decoded all the same. */
int hf_decode_chunks(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                     const uint32_t* d_chunk_off, const uint8_t* d_side, uint64_t n, int layout,
                     const void* d_chunk_codebooks, const uint32_t* d_chunk_ids, uint64_t n_ids,
                     void* d_out, int check);
/* decode-on-read into a paged buffer: listed chunk k (d_chunk_ids[k], a full chunk) goes to page
 * d_out_slots[k] of d_out, a buffer of n_slots pages of 16384 elements; a slot >= n_slots is
 * reported as HF_ERR_CORRUPT and never written.  Two items naming the same slot race.  No
 * counterpart in the reference (its decoder is an empty module, decoder/top_decoder.scala:6-10);
 * meant for KV-cache pages decoded straight into the page pool of a paged-attention runtime. */
int hf_decode_chunks_to(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                        const uint32_t* d_chunk_off, const uint8_t* d_side, uint64_t n, int layout,
                        const void* d_chunk_codebooks, const uint32_t* d_chunk_ids,
                        const uint32_t* d_out_slots, uint64_t n_ids, void* d_out, uint64_t n_slots,
                        int check);
/* Slotted placement of a per-block stream.  Chunk c is encoded with its own codebook exactly as by
 * hf_encode_chunks -- the same words in the same order -- but its place is fixed before the
 * encoder runs: it owns a slot of HF_SLOT_WORDS(B_c) words, B_c = the bits of its codes (from
 * d_freq_chunks, the histograms the codebooks were built from, and the codebooks' lengths), which
 * no chunk can outgrow.  d_chunk_off[c] = start of slot c (prefix sum of the slot sizes,
 * nchunks + 1 values, the last one the stream's words), d_chunk_words[c] = words the chunk filled;
 * the rest of a slot is zero.  The payload is 1.5-3 % larger than the dense one (about 54 words per
 * full chunk) and is written without staging or look-back, at 1.1-1.45x the dense encoder's rate.
 * It is not a dense payload: it is decoded by hf_decode_chunks_slotted only, and stored as a
 * version-2 container with its words array (hf_container_pack_placed).  payload_cap_words >= hf_slotted_bound_words(n) always fits; a
 * smaller buffer that the slots do not fit gives HF_ERR_OVERFLOW and nothing is written.
 * Histograms that are not this data's give HF_ERR_CORRUPT (a chunk outgrew its slot; nothing is
 * written outside the slot).  No counterpart in the reference (its top level stops at the
 * codebook, top.scala:60-67); meant for KV-cache pages, where encode rate matters more than the
 * last 3 % of ratio. */
uint64_t hf_slotted_bound_words(uint64_t n);
int hf_encode_chunks_slotted(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                             const void* d_chunk_codebooks, const uint32_t* d_freq_chunks,
                             uint8_t* d_side, uint32_t* d_payload, uint64_t payload_cap_words,
                             uint32_t* d_chunk_off, uint32_t* d_chunk_words,
                             uint64_t* h_total_words);
/* The same placement for a stream with ONE codebook (hf_system_codebook_dev): the slot sizes
 * still come from per-chunk histograms, so the caller counts per chunk
 * (hf_symbol_stat_chunks_dev) and folds those counts into the stream's 256 (hf_symbol_stat_fold_dev
 * adds them into d_freq256, as hf_symbol_stat_dev adds a tensor's) instead of counting the stream
 * in one piece.  The chunk words are those hf_encode writes; decoded by hf_decode_slotted only.
 * No counterpart in the reference, as above. */
int hf_symbol_stat_fold_dev(hf_ctx* ctx, const uint32_t* d_freq_chunks, uint64_t nchunks,
                            uint64_t* d_freq256);
int hf_encode_slotted(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                      const void* d_codebook, const uint32_t* d_freq_chunks, uint8_t* d_side,
                      uint32_t* d_payload, uint64_t payload_cap_words, uint32_t* d_chunk_off,
                      uint32_t* d_chunk_words, uint64_t* h_total_words);
int hf_decode_slotted(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                      const uint32_t* d_chunk_off, const uint32_t* d_chunk_words,
                      const uint8_t* d_side, uint64_t n, int layout, const void* d_codebook,
                      void* d_out, int check);
/* decodes a slotted stream: all chunks (d_chunk_ids NULL), listed chunks to consecutive pages
 * (d_out_slots NULL), or listed chunks to named pages -- as hf_decode_chunks / hf_decode_chunks_to */
int hf_decode_chunks_slotted(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                             const uint32_t* d_chunk_off, const uint32_t* d_chunk_words,
                             const uint8_t* d_side, uint64_t n, int layout,
                             const void* d_chunk_codebooks, const uint32_t* d_chunk_ids,
                             const uint32_t* d_out_slots, uint64_t n_ids, void* d_out,
                             uint64_t n_slots, int check);
/* ---- paged stream (format version 2): per-block codebooks, every page self-contained ----------
 * The reference has no stored form and no decoder (top_encoder.scala:17-21 ends at the {code,
 * length, mode} beat; decoder/top_decoder.scala:6-10 is an empty bundle; its figure
 * reports/figs/decoder.svg shows a scheduler feeding parallel sub-decoders).  This is the stored
 * form for per-block streams that are written once and read page by page (KV-cache pages): a page
 * is decoded by a workgroup whose four wavefronts are four sub-decoders, its tables built once by
 * all of them; a page's size follows from its histogram and codebook alone, so every page's place
 * is known before the encoder starts (no look-back, no slack); and a page carries its own codebook
 * record, so a reader fetches one contiguous piece per page.
 *
 * A page of HF_CHUNK_SYMS symbols has 256 strings of 64 symbols: string (q, l) = the symbols lane l
 * owns in rounds 8q .. 8q+7 (symbol index ((r * 64 + l) * 8 + j), round r, j = 0..7), their codes
 * MSB first, symbols past the end of a last, partial page left out.  Strings lie bit-contiguous in
 * 128 regions: region r = qf * 64 + l (qf = 0, 1) holds string (qf, l) read forward from the
 * region's first bit and string (qf + 2, l) read backward from its last bit.  Stream bit i of the
 * strings area is bit 31 - i % 32 of word i / 32.  Page layout in 32-bit words:
 *     0       head | mode << 9 | B << 10    B = bits of all codes of the page
 *     1       entropy_q16
 *     2..9    u16[16]  number of ranks with code length 1 .. 16 (lengths never decrease with the
 *             rank, codes are canonical in rank order)
 *     10..    the first `head` symbols in rank order (count desc, symbol asc), a byte each, zero
 *             padded to a word; head is the smallest number such that the remaining ranks are
 *             the remaining symbols in ascending order
 *     then    HF_PAGE_GAP_WORDS words: the region lengths R_0 .. R_127, 12 bits each, field r at
 *             bits [12 r, 12 r + 12) of the table (bit b of the table = bit b % 32 of word b / 32)
 *     then    ceil(B / 32) words of strings
 * d_page_off[c] = first word of page c, d_page_off[nchunks] = the stream's words.
 *
 * hf_encode_pages: d_chunk_codebooks and d_freq_chunks as for hf_encode_chunks_slotted (the
 * records of hf_system_codebook_chunks_dev and the histograms they were built from; a histogram
 * that is not its page's gives HF_ERR_CORRUPT and that page is not written).  payload_cap_words >=
 * hf_pages_bound_words(n) always fits; a buffer the pages do not fit gives HF_ERR_OVERFLOW and
 * nothing is written.  h_total_words as for hf_encode.
 * hf_decode_pages: all pages to their own places (d_page_ids NULL), the n_ids listed pages to
 * consecutive pages of d_out (d_out_slots NULL), or to the named pages of a pool of n_slots pages
 * -- as hf_decode_chunks / hf_decode_chunks_to; listed pages must be full pages.  A damaged page
 * (sizes that do not match, a record that is not a canonical order, region lengths that do not add
 * up, strings that do not meet) is reported as HF_ERR_CORRUPT with check != 0; reads and writes
 * stay inside the page and its output whatever it holds. */
#define HF_PAGE_HDR_WORDS 10u
#define HF_PAGE_GAP_WORDS 48u
uint64_t hf_pages_bound_words(uint64_t n);
int hf_encode_pages(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                    const void* d_chunk_codebooks, const uint32_t* d_freq_chunks, uint8_t* d_side,
                    uint32_t* d_payload, uint64_t payload_cap_words, uint32_t* d_page_off,
                    uint64_t* h_total_words);
int hf_decode_pages(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                    const uint32_t* d_page_off, const uint8_t* d_side, uint64_t n, int layout,
                    const uint32_t* d_page_ids, const uint32_t* d_out_slots, uint64_t n_ids,
                    void* d_out, uint64_t n_slots, int check);
/* bulk read of a stored page's record: code/length by symbol, mode, entropy (synchronises) */
int hf_page_codebook_download(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                              const uint32_t* d_page_off, uint64_t page, uint16_t* h_code,
                              uint8_t* h_len, int* mode, uint32_t* entropy_q16);

/* Paged stream with ONE codebook for the whole stream (hf_system_codebook_dev on the stream's
 * counts -- or on counts summed over GPUs, hf_symbol_stat_allreduce): the pages carry no record.
 *   word 0   B << 10 | 1 << 29   (bit 29: "no record, the codes are the stream's"; bits 0..9 zero)
 *   then     the 48 words of region lengths and the ceil(B / 32) words of strings, as above
 * A page takes 49 + ceil(B / 32) words: no look-ahead words, no per-lane padding, 0.10 bits per
 * symbol of page overhead.  d_freq_chunks are the per-page counts (hf_symbol_stat_chunks_dev;
 * hf_symbol_stat_fold_dev sums them into the stream's 256 counts, so the data is read once for
 * both); d_codebook must be a canonical codebook, which every codebook this library builds is
 * (an uploaded table of another shape makes every page report HF_ERR_CORRUPT).  The decoder copies
 * tables made once per call from the codebook instead of building them per page.  Everything
 * else as hf_encode_pages / hf_decode_pages; hf_pages_bound_words bounds these pages too.  In a
 * container: HF_PLACEMENT_PAGED with HF_CODEBOOK_STREAM.
 * d_freq_chunks == NULL: no per-page counts.  The encoder then places the pages while it encodes
 * them -- every page publishes its size once its codes are looked up and sums the sizes in front
 * of it (a walk back over one 8-byte descriptor per page, kept in the context: hf_ctx_reserve
 * with HF_RESERVE_PLACED makes them up front).  The stream is the same bit for bit; the call
 * reads the data once and needs no counts, but waits on the pages in front and is slower than
 * counting per page and placing beforehand (1.9 against 0.8 + 1.2 ms per 4.3 GB of bf16, of
 * which 0.6 ms is the stream's histogram either way).  A payload buffer that is too small is
 * reported as HF_ERR_OVERFLOW after the pages that fit have been written (with counts nothing
 * is written); a walk that does not end within a second gives HF_ERR_TIMEOUT. */
int hf_encode_pages_shared(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                           const void* d_codebook, const uint32_t* d_freq_chunks, uint8_t* d_side,
                           uint32_t* d_payload, uint64_t payload_cap_words, uint32_t* d_page_off,
                           uint64_t* h_total_words);
int hf_decode_pages_shared(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                           const uint32_t* d_page_off, const uint8_t* d_side, uint64_t n, int layout,
                           const void* d_codebook, const uint32_t* d_page_ids,
                           const uint32_t* d_out_slots, uint64_t n_ids, void* d_out,
                           uint64_t n_slots, int check);

/* bulk read of one record: code/length by symbol, mode, entropy (synchronises) */
int hf_chunk_codebook_download(hf_ctx* ctx, const void* d_chunk_codebooks, uint64_t chunk,
                               uint16_t* h_code, uint8_t* h_len, int* mode, uint32_t* entropy_q16);

/* ---- container: one self-describing blob per encoded stream --------------------------------
 * The reference stops at one {code, length, mode} beat per symbol (top_encoder.scala:17-21) and
 * defines no stored form; this is the wire format of this library (little endian).  Version 1
 * holds dense streams (hf_encode, hf_encode_chunks); version 2 adds the other two placements, the
 * slotted one (hf_encode_slotted, hf_encode_chunks_slotted) and the paged one (hf_encode_pages).
 * A dense stream is always written as version 1; both versions are read.
 *
 *   header, HF_CONTAINER_HEADER_BYTES = 128:
 *     0 magic "HUFHIP1\0" | 8 u32 version | 12 u32 header_bytes | 16 u32 layout |
 *     20 u32 codebook_kind | 24 u32 chunk_syms | 28 u32 max_code_len | 32 u32 mode |
 *     36 u32 entropy_q16 | 40 u64 n | 48 u64 nchunks | 56 u64 payload_words |
 *     64 u64 off_codebook | 72 u64 off_chunk_off | 80 u64 off_side | 88 u64 off_payload |
 *     96 u64 total_bytes | version 2: 104 u32 placement | 108 zero | 112 u64 off_chunk_words |
 *     otherwise zero up to 127
 *   sections, each starting on a 16-byte boundary, zero padded, in this order:
 *     codebook   HF_CODEBOOK_STREAM: code u16[256], length u8[256] (768 bytes);
 *                HF_CODEBOOK_PER_CHUNK: nchunks records of HF_CONTAINER_CHUNK_CODEBOOK_BYTES:
 *                rank2sym u8[256] (symbols in (count desc, symbol asc) order), u16[16] number of
 *                ranks with code length 1..16 (lengths never decrease with rank, codes are
 *                canonical in rank order), u32 mode, u32 entropy_q16;
 *                absent (off_codebook = 0) for HF_PLACEMENT_PAGED with HF_CODEBOOK_PER_CHUNK:
 *                the records are in the pages
 *     chunk_off  u32[nchunks + 1]: first words of the chunks / slots / pages, then the total
 *     chunk_words  HF_PLACEMENT_SLOTTED only: u32[nchunks], the words each chunk fills of its slot
 *     side       HF_LAYOUT_BF16_EXP: u8[n]; HF_LAYOUT_FP16_EXP, _FP8_E4M3_EXP, _FP4_E2M1_EXP: the packed
 *                side fields, hf_fields_side_bytes(n, layout) bytes (the payload then codes the
 *                symbol bytes of hf_fields_split); absent otherwise
 *     payload    u32[payload_words], padded to at least 16 bytes
 * Every section can be handed to the decoders in place (d_blob + offset), so a stored stream is
 * decoded, whole or block by block, without unpacking it. */
#define HF_CONTAINER_VERSION 2
#define HF_CONTAINER_HEADER_BYTES 128
#define HF_CONTAINER_CHUNK_CODEBOOK_BYTES 296
#define HF_CODEBOOK_STREAM 0
#define HF_CODEBOOK_PER_CHUNK 1
#define HF_PLACEMENT_DENSE 0   /* chunks back to back (hf_encode, hf_encode_chunks)          */
#define HF_PLACEMENT_SLOTTED 1 /* chunks in slots, with chunk_words                          */
#define HF_PLACEMENT_PAGED 2   /* pages: with their records (hf_encode_pages, HF_CODEBOOK_PER_CHUNK)
                                  or without (hf_encode_pages_shared, HF_CODEBOOK_STREAM) */

typedef struct hf_container_info {
    uint32_t version;
    uint32_t layout;
    uint32_t codebook_kind;
    uint32_t chunk_syms;
    uint32_t max_code_len;
    uint32_t mode;        /* stream-wide codebook: compression_mode; per chunk: in the records */
    uint32_t entropy_q16; /* stream-wide codebook only */
    uint32_t placement;   /* HF_PLACEMENT_* */
    uint64_t n;
    uint64_t nchunks;
    uint64_t payload_words;
    uint64_t off_codebook; /* 0 when the pages carry the records */
    uint64_t off_chunk_off;
    uint64_t off_side; /* 0 when there is no side stream */
    uint64_t off_payload;
    uint64_t total_bytes;
    uint64_t off_chunk_words; /* 0 unless slotted */
} hf_container_info;

/* Section offsets and the total size for a stream of n symbols (host arithmetic only);
 * hf_container_plan is hf_container_plan_placed with HF_PLACEMENT_DENSE. */
int hf_container_plan(uint64_t n, int layout, int codebook_kind, uint64_t payload_words,
                      hf_container_info* info);
int hf_container_plan_placed(uint64_t n, int layout, int codebook_kind, int placement,
                             uint64_t payload_words, hf_container_info* info);
/* Write the blob: header from `info` (as filled by a plan call; mode / entropy are taken from the
 * device codebook), codebook(s) in wire form, offsets, side bytes, payload.  All pointers are
 * device pointers; d_codebook is an HF_CODEBOOK_DEV_BYTES codebook or an array of
 * HF_CHUNK_CODEBOOK_BYTES records according to info->codebook_kind (not used, may be NULL, for a
 * paged stream); d_chunk_words only for a slotted one.  hf_container_pack is the same without
 * d_chunk_words (dense and paged streams).  Synchronises. */
int hf_container_pack(hf_ctx* ctx, hf_container_info* info, const void* d_codebook,
                      const uint32_t* d_chunk_off, const uint8_t* d_side,
                      const uint32_t* d_payload, void* d_blob, uint64_t blob_cap_bytes);
int hf_container_pack_placed(hf_ctx* ctx, hf_container_info* info, const void* d_codebook,
                             const uint32_t* d_chunk_off, const uint32_t* d_chunk_words,
                             const uint8_t* d_side, const uint32_t* d_payload, void* d_blob,
                             uint64_t blob_cap_bytes);
/* Read and validate the header of a blob in device memory (HF_ERR_CORRUPT if it is not a
 * version-1 or version-2 container that fits blob_bytes).  Synchronises. */
int hf_container_open(hf_ctx* ctx, const void* d_blob, uint64_t blob_bytes,
                      hf_container_info* info);
/* Rebuild the device codebook(s) of an opened blob into d_codebook_out (HF_CODEBOOK_DEV_BYTES,
 * or nchunks * HF_CHUNK_CODEBOOK_BYTES for per-chunk codebooks; nothing to do, and
 * d_codebook_out may be NULL, for a paged stream).  Synchronises. */
int hf_container_codebook(hf_ctx* ctx, const void* d_blob, const hf_container_info* info,
                          void* d_codebook_out);

/* ---- TreeQualityAnalyzer (tree_q_analyser.scala:10-47) ------------------------------------- */
int hf_quality(hf_ctx* ctx, const uint64_t* h_freq, const uint8_t* h_len, int N, int wt_width,
               uint16_t* avg_len_q8, uint16_t* efficiency, uint16_t* compression_ratio);

/* ---- per-kernel device timing of the last call (milliseconds, for the bench) --------------- */
int hf_last_kernel_ms(hf_ctx* ctx, float* ms);
int hf_set_timing(hf_ctx* ctx, int enabled);

#ifdef __cplusplus
}
#endif
#endif /* HUFFHIP_H */
