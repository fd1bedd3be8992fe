// Shared declarations of libhuffhip's translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/huffhip.h"

#define HF_LANES 64
#define HF_GROUP 8          // consecutive symbols a lane takes per round
#define HF_ROUNDS 32        // rounds per chunk
#define HF_LANE_SYMS 256    // HF_GROUP * HF_ROUNDS
#define HF_LUT_BITS 12      // first-level decode table width
#define HF_LUT_SIZE (1 << HF_LUT_BITS)

// Device-resident codebook (HF_CODEBOOK_DEV_BYTES reserves room for it).
struct hf_cb_dev {
    uint32_t enc[256];           // code | len << 16, by raw symbol
    uint16_t lut[HF_LUT_SIZE];   // sym << 8 | len for codes of <= HF_LUT_BITS bits;
                                 // len == 0: escape, the sym field selects a second-level table
    uint16_t sub[256 * 16];      // second level: 16 entries per distinct 12-bit prefix of a
                                 // longer code, indexed by the next 4 bits; sym << 8 | len
    uint16_t code[256];
    uint8_t len[256];
    uint32_t n_sub;              // second-level tables in use
    uint32_t mode;               // HF_MODE_BYPASS / HF_MODE_HUFFMAN actually in force
    uint32_t entropy_q16;
    uint32_t entropy_mode;       // what the entropy decision alone says
    uint32_t max_len;
    uint32_t chunk_words_hint;   // payload words a full chunk is expected to take (symbol counts
                                 // x code lengths, schedule overhead included); 0 = unknown
    uint32_t pair_hits;          // entries of lut2 that resolve two symbols (of HF_LUT_SIZE)
    uint32_t pad[1];
    // Multi-symbol first level, by the next 12 bits.  Byte 0: bits consumed by what the entry
    // resolves; byte 1: first symbol; byte 2: second symbol; byte 3: length of the first code,
    // bit 7 set when the entry resolves two symbols (both codes lie inside the 12 bits).  An
    // escape (first code longer than 12 bits) has bytes 0 and 3 zero and the second-level table's
    // number in byte 1, as in `lut`.
    uint32_t lut2[HF_LUT_SIZE];
};
// a stream is decoded two symbols per lookup when at least this many of the 4096 entries resolve
// two symbols: with the code's own distribution (p = 2^-len) that is the probability that a pair
// of symbols hits, and all 64 lanes of a wavefront must hit for the short path to be taken
#define HF_PAIR_HITS_MIN 4053u   // 0.9895 ^ 64 = 0.51
static_assert(sizeof(hf_cb_dev) <= HF_CODEBOOK_DEV_BYTES, "codebook struct outgrew its slot");

// Per-chunk codebook record (HF_CHUNK_CODEBOOK_BYTES each): what the encoder needs by symbol
// and what the decoder needs in canonical (rank) order.  Codes of one chunk are canonical in
// rank order, so (rank2sym, len_by_rank) determines them completely.
struct hf_chunk_cb {
    uint32_t enc[256];        // code | len << 16, by raw symbol
    uint8_t rank2sym[256];    // symbol of rank p, ranks ordered (count desc, symbol asc)
    uint8_t len_by_rank[256]; // code length of rank p (non-decreasing in p)
    uint32_t mode;
    uint32_t entropy_q16;
    uint32_t pad[2];
};
static_assert(sizeof(hf_chunk_cb) == HF_CHUNK_CODEBOOK_BYTES, "per-chunk codebook record size");

// Stored form of a per-chunk codebook inside a container (HF_CONTAINER_CHUNK_CODEBOOK_BYTES).
struct hf_wire_chunk_cb {
    uint8_t rank2sym[256];  // symbols in (count desc, symbol asc) order
    uint16_t nlen[16];      // number of ranks with code length 1 .. 16
    uint32_t mode;
    uint32_t entropy_q16;
};
static_assert(sizeof(hf_wire_chunk_cb) == HF_CONTAINER_CHUNK_CODEBOOK_BYTES, "wire record size");

// Global-memory I/O block of the single-wavefront table stages.
struct hf_stage_io {
    uint64_t freq[256];
    uint64_t sorted[256];
    uint64_t weights[511];
    uint64_t pad0;
    uint16_t rank2sym[256];
    uint16_t lefts[512];
    uint16_t rights[512];
    uint16_t depths[256];
    uint16_t adj[256];
    uint16_t code[256];
    uint8_t len[256];
    uint32_t root;
    uint32_t entropy_q16;
    uint32_t mode;
    int32_t status;
    uint32_t migrate_steps;
    uint32_t pad1[3];
};

enum hf_stage_op {
    HF_OP_SORT = 1,
    HF_OP_ENTROPY = 2,
    HF_OP_TREE = 3,
    HF_OP_ADJUST = 4,
    HF_OP_CODEBOOK = 5,
};

// in-kernel error bits accumulated in the ctx error word
#define HF_KERR_TIMEOUT 1u
#define HF_KERR_OVERFLOW 2u
#define HF_KERR_CORRUPT 4u

// launchers (one per translation unit)
void hf_launch_hist(const void* d_data, uint64_t n, int layout, unsigned long long* d_freq256,
                    int num_cu, hipStream_t s);
void hf_launch_stage(hf_stage_io* io, int N, int op, int lmax, unsigned long long thr,
                     hipStream_t s);
void hf_launch_system(const unsigned long long* d_freq256, int mode, unsigned long long thr,
                      hf_cb_dev* cb, hipStream_t s);
void hf_launch_tables(hf_cb_dev* cb, hipStream_t s);
void hf_launch_pairs(const void* d_data, uint64_t n, int layout, const hf_cb_dev* cb,
                     hf_code_beat* out, hipStream_t s);
void hf_launch_encode(const void* d_data, uint64_t n, int layout, const hf_cb_dev* cb,
                      uint8_t* d_side, uint32_t* d_payload, uint64_t cap_words,
                      uint32_t* d_chunk_off, unsigned long long* d_desc, uint32_t* d_ticket,
                      uint32_t* d_spill, uint32_t* d_err, int num_cu, hipStream_t s);
size_t hf_encode_spill_bytes(int num_cu);
bool hf_encode_single_tile(uint64_t nchunks);
size_t hf_scan_aux_bytes(uint64_t m);
size_t hf_slot_dump_bytes();
void hf_launch_encode_chunks_slotted(const void* d_data, uint64_t n, int layout,
                                     const uint32_t* d_enc, uint32_t enc_stride,
                                     const uint32_t* d_freq_chunks,
                                     uint8_t* d_side, uint32_t* d_payload, uint64_t cap_words,
                                     uint32_t* d_slot_off, uint32_t* d_chunk_words,
                                     unsigned long long* d_scan_aux, uint32_t* d_dump,
                                     uint32_t* d_err, int num_cu, hipStream_t s);
void hf_launch_system_block(const void* d_data, uint32_t n, int layout, int mode,
                            unsigned long long thr, unsigned long long* d_freq256, hf_cb_dev* cb,
                            hipStream_t s);
void hf_launch_fold_chunks(const uint32_t* d_freq_chunks, uint64_t nchunks,
                           unsigned long long* d_freq256, hipStream_t s);
void hf_launch_hist_pages(const void* d_data, uint64_t n, int layout, uint32_t* d_freq_chunks,
                          unsigned long long* d_freq256, int num_cu, hipStream_t s);
void hf_launch_hist_chunks(const void* d_data, uint64_t n, int layout, uint32_t* d_freq_chunks,
                           hipStream_t s);
#define HF_CB_BATCH 65536u  // per-block codebooks built per pass through the scratch block
#define HF_CB_SCRATCH_PER_CHUNK 1040u  // bytes of scratch per codebook in flight
size_t hf_chunk_cb_scratch_bytes(void);
// `batch` = codebooks the scratch block at d_scratch holds (HF_CB_SCRATCH_PER_CHUNK bytes each)
void hf_launch_system_chunks(const uint32_t* d_freq_chunks, uint64_t nchunks, int mode,
                             unsigned long long thr, hf_chunk_cb* cbs, void* d_scratch,
                             uint32_t batch, hipStream_t s);
void hf_launch_encode_chunks(const void* d_data, uint64_t n, int layout, const hf_chunk_cb* cbs,
                             uint8_t* d_side, uint32_t* d_payload, uint64_t cap_words,
                             uint32_t* d_chunk_off, unsigned long long* d_desc,
                             uint32_t* d_ticket, uint32_t* d_spill, uint32_t* d_err, int num_cu,
                             hipStream_t s);
void hf_launch_decode_chunks(const uint32_t* d_payload, uint64_t payload_words,
                             const uint32_t* d_chunk_off, const uint32_t* d_chunk_words,
                             const uint8_t* d_side, uint64_t n,
                             int layout, const hf_chunk_cb* cbs, const uint32_t* d_chunk_ids,
                             const uint32_t* d_out_slots, uint64_t n_slots, uint64_t n_ids,
                             void* d_out, uint32_t* d_err, int num_cu, hipStream_t s);
void hf_launch_cb_to_wire(const hf_chunk_cb* cbs, uint64_t nchunks, hf_wire_chunk_cb* out,
                          hipStream_t s);
void hf_launch_wire_to_cb(const hf_wire_chunk_cb* in, uint64_t nchunks, hf_chunk_cb* cbs,
                          uint32_t* d_err, hipStream_t s);
// Paged stream (format version 2; layout: include/huffhip.h).
// (HF_PAGE_HDR_WORDS: head | mode | bits, entropy, 16 length counts; HF_PAGE_GAP_WORDS: 128 region
// lengths of 12 bits)
#define HF_PAGE_MAX_WORDS (HF_PAGE_HDR_WORDS + 64u + HF_PAGE_GAP_WORDS + HF_CHUNK_SYMS * 16u / 32u)
void hf_launch_scan_u32(uint32_t* d_a, uint64_t m, unsigned long long* d_aux,
                        unsigned long long limit, uint32_t* d_err, hipStream_t s);
void hf_launch_encode_pages(const void* d_data, uint64_t n, int layout, const hf_chunk_cb* cbs,
                            const hf_cb_dev* shared_cb, const uint32_t* d_freq_chunks, uint8_t* d_side,
                            uint32_t* d_payload, uint64_t cap_words, uint32_t* d_page_off,
                            unsigned long long* d_scan_aux, uint32_t* d_err, hipStream_t s);
void hf_launch_encode_pages_once(const void* d_data, uint64_t n, int layout, const hf_cb_dev* shared_cb,
                                 uint8_t* d_side, uint32_t* d_payload, uint64_t cap_words,
                                 uint32_t* d_page_off, unsigned long long* d_desc, uint32_t* d_err,
                                 hipStream_t s);
size_t hf_page_tab_image_bytes(void);
void hf_launch_decode_pages(const uint32_t* d_payload, uint64_t payload_words,
                            const uint32_t* d_page_off, const uint8_t* d_side, uint64_t n,
                            int layout, const hf_cb_dev* shared_cb, void* d_tab_image,
                            const uint32_t* d_page_ids, const uint32_t* d_out_slots,
                            uint64_t n_slots, uint64_t n_ids, void* d_out, uint32_t* d_err,
                            hipStream_t s);
uint64_t hf_fields_side_bytes_impl(uint64_t n, int layout);
void hf_launch_fields_split(const void* d_data, uint64_t n, int layout, uint8_t* d_sym,
                            uint8_t* d_side, hipStream_t s);
void hf_launch_fields_merge(const uint8_t* d_sym, const uint8_t* d_side, uint64_t n, int layout,
                            void* d_out, hipStream_t s);
void hf_launch_decode(const uint32_t* d_payload, uint64_t payload_words,
                      const uint32_t* d_chunk_off, const uint32_t* d_chunk_words,
                      const uint8_t* d_side, uint64_t n, int layout,
                      const hf_cb_dev* cb, void* d_out, uint32_t* d_err, int num_cu,
                      hipStream_t s);

#ifdef __HIPCC__
__device__ __forceinline__ int hf_lane() { return (int)__lane_id(); }

// wave-wide sum of a 32-bit value (all 64 lanes must call)
__device__ __forceinline__ uint32_t hf_wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ unsigned long long hf_wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// number of set bits of `m` below this lane
__device__ __forceinline__ uint32_t hf_mbcnt(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Global-memory (address space 1) views used by the streaming kernels: with a chunk's base held
// in scalar registers, a per-round scalar offset and a constant per-lane vector offset, an access
// needs no vector address arithmetic.
typedef __attribute__((address_space(1))) uint8_t hf_gmem_u8;
typedef uint32_t hf_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t hf_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) hf_u32x2 hf_gmem_u32x2;
typedef __attribute__((address_space(1))) hf_u32x4 hf_gmem_u32x4;

// A wave-uniform global pointer moved into scalar registers.  The builtin returns int: both
// halves go through uint32_t so that the low one is not sign-extended over the high one.
__device__ __forceinline__ hf_gmem_u8* hf_uniform_ptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    return (hf_gmem_u8*)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}
#endif
