// Paged stream: the second stored form of a per-block stream (format version 2).
//
// The reference has neither a packed stream nor a decoder: top_encoder.scala:17-21 ends at the
// {code, length, mode} beat, decoder/top_decoder.scala:6-10 is an empty bundle, and its only
// decoder design intent, reports/figs/decoder.svg, shows a scheduler feeding parallel sub-decoders
// whose outputs a combiner puts back in order.  Here a page (16384 symbols with one codebook,
// entropy_aware_tree_builder.scala:6-169) is the unit of a WORKGROUP: its four wavefronts are the
// four sub-decoders, the workgroup is the scheduler, and the tables are built once per page by all
// 256 threads.
//
// Format (include/huffhip.h has the byte layout).  The 64 symbols lane l owns in rounds 8q .. 8q+7
// (the ownership of the packed stream, hf_encode.hip) form string (q, l); their codes are
// concatenated MSB first without padding.  Strings are stored bit-contiguous in 128 regions: region
// r = qf * 64 + l holds string (qf, l) read forward from its first bit and string (qf + 2, l) read
// backward from its last bit, so one 12-bit length per region places all 256 strings, and the
// direction is the same for a whole wavefront.  No look-ahead words, no per-lane padding: a page
// takes ceil(B / 32) words for the B bits of its codes, B = sum(count x length) is known from the
// page's histogram before the encoder starts, and so is every page's place (a prefix sum) -- no
// look-back, no staging of unknown size.  The page also carries its codebook record: the symbols
// in rank order as far as the order is not plain ascending, and the number of ranks per code
// length.  Everything a reader needs for one page lies in one contiguous piece of the payload.
//
// A stream whose pages share ONE codebook (the stream-wide one of hf_system_codebook_dev) stores
// pages without a record: word 0 = B << 10 | 1 << 29, the region lengths, the strings.  Its decoder
// copies ready-made tables (built once per call from the codebook) into LDS instead of building
// them per page.
//
// HBM traffic per byte symbol: encode 1 B read + L/8 B written (+ 2.3 KB per page of counts and
// codebook read for the sizes), decode L/8 B read + 1 B written.
#include "hf_common.h"
#include "hf_dec_tables.h"

namespace {

constexpr uint32_t kHdr = HF_PAGE_HDR_WORDS;
constexpr uint32_t kGap = HF_PAGE_GAP_WORDS;
// A page's image in LDS: header, record, region lengths and strings, with the alignment phase of
// its place in the payload (image word a0 + k = page word k, a0 = first word's index mod 4), so
// that 16-byte pieces of the image are 16-byte pieces of the payload.  Pages of up to 8.1 bits per
// symbol fit (a bypass page, 8 bits, does); larger ones -- Huffman forced on data it does not
// compress -- go word by word through global memory.
constexpr int kImgWords = 4224;
constexpr int kImgSlack = 40;  // reads of a lane's window may run this far past its strings

__host__ __device__ __forceinline__ uint32_t page_words(const uint32_t head, const uint32_t bits) {
    return kHdr + (head + 3u) / 4u + kGap + (bits + 31u) / 32u;
}

constexpr uint32_t kSharedFlag = 1u << 29;  // word 0 of a page without a record
__host__ __device__ __forceinline__ uint32_t page_words_shared(const uint32_t bits) {
    return 1u + kGap + (bits + 31u) / 32u;
}

// ranks a record lists explicitly: the largest p with rank2sym[p-1] >= rank2sym[p], else 0 (from
// there on the symbols ascend, and "all symbols not listed, ascending" says the same).  Lane l
// holds ranks 4l .. 4l+3 in r4.
__device__ __forceinline__ uint32_t head_of_wave(const uint32_t r4, const int lane) {
    const uint32_t prev = __shfl_up(r4, 1) >> 24;
    const uint32_t b0 = r4 & 0xFFu, b1 = (r4 >> 8) & 0xFFu, b2 = (r4 >> 16) & 0xFFu, b3 = r4 >> 24;
    const bool d[4] = {lane > 0 && prev >= b0, b0 >= b1, b1 >= b2, b2 >= b3};
    uint32_t head = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u64 m = __ballot(d[i]);
        if (m) {
            const uint32_t cand = 4u * (63u - (uint32_t)__builtin_clzll(m)) + (uint32_t)i;
            head = cand > head ? cand : head;
        }
    }
    return head;
}

// size[c] = words of page c from its counts and its codebook; size[nchunks] = 0 (the scan turns
// the array into the page offsets and the total).
__global__ __launch_bounds__(256) void hf_k_page_sizes(const uint32_t* __restrict__ freq_chunks,
                                                       const hf_chunk_cb* __restrict__ cbs,
                                                       const uint32_t nchunks,
                                                       uint32_t* __restrict__ size) {
    const int lane = hf_lane();
    const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (c > nchunks) return;
    if (c == nchunks) {
        if (lane == 0) size[c] = 0;
        return;
    }
    const hf_chunk_cb* cb = cbs + c;
    const uint4 f = reinterpret_cast<const uint4*>(freq_chunks + (size_t)c * 256)[lane];
    const uint4 e = reinterpret_cast<const uint4*>(cb->enc)[lane];
    uint32_t bits = f.x * (e.x >> 16) + f.y * (e.y >> 16) + f.z * (e.z >> 16) + f.w * (e.w >> 16);
    bits = hf_wave_sum_u32(bits);
    // counts that are not a page's (more than 16384 symbols, or a wrapped sum) are clamped: the
    // 32-bit sums of the scan cannot wrap and the encoder finds the page's size wrong
    const uint32_t most = HF_CHUNK_SYMS * (uint32_t)HF_MAX_CODE_LEN;
    bits = bits < most ? bits : most;
    const uint32_t head = head_of_wave(reinterpret_cast<const uint32_t*>(cb->rank2sym)[lane], lane);
    if (lane == 0) size[c] = page_words(head, bits);
}

// the same for pages without a record, all coded with the stream's codebook
__global__ __launch_bounds__(256) void hf_k_page_sizes_shared(const uint32_t* __restrict__ freq_chunks,
                                                              const hf_cb_dev* __restrict__ cb,
                                                              const uint32_t nchunks,
                                                              uint32_t* __restrict__ size) {
    const int lane = hf_lane();
    const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (c > nchunks) return;
    if (c == nchunks) {
        if (lane == 0) size[c] = 0;
        return;
    }
    const uint4 f = reinterpret_cast<const uint4*>(freq_chunks + (size_t)c * 256)[lane];
    const uint4 e = reinterpret_cast<const uint4*>(cb->enc)[lane];
    uint32_t bits = f.x * (e.x >> 16) + f.y * (e.y >> 16) + f.z * (e.z >> 16) + f.w * (e.w >> 16);
    bits = hf_wave_sum_u32(bits);
    const uint32_t most = HF_CHUNK_SYMS * (uint32_t)HF_MAX_CODE_LEN;
    bits = bits < most ? bits : most;
    if (lane == 0) size[c] = page_words_shared(bits);
}

// ---- encoder ---------------------------------------------------------------------------------

// the 64 symbols of string (q, lane): sy[2 rr], sy[2 rr + 1] hold the 8 symbols of round 8q + rr,
// one per byte; bf16 elements leave their side bytes in the side stream on the way
template <int LAYOUT, bool FULL>
__device__ __forceinline__ void load_string(const void* chunk_in, uint8_t* side_chunk,
                                            const uint32_t q, const int lane, const uint32_t nvalid,
                                            uint32_t (&sy)[16]) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const uint32_t g = (q * 8u + (uint32_t)rr) * 64u + (uint32_t)lane;
        if (LAYOUT == HF_LAYOUT_BF16_EXP) {
            uint32_t raw[4] = {0, 0, 0, 0};
            const bool whole = FULL || g * 8u + 8u <= nvalid;
            if (whole) {
                const hf_u32x4 v =
                    __builtin_nontemporal_load(reinterpret_cast<const hf_u32x4*>(chunk_in) + g);
                raw[0] = v.x, raw[1] = v.y, raw[2] = v.z, raw[3] = v.w;
            } else {
                const uint16_t* p = reinterpret_cast<const uint16_t*>(chunk_in) + (size_t)g * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (g * 8u + (uint32_t)i < nvalid) raw[i >> 1] |= (uint32_t)p[i] << (16 * (i & 1));
            }
            uint32_t sd[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // exponents: bits 7..14 of each half; side byte: sign over the mantissa
                sy[2 * rr + h] = __builtin_amdgcn_perm(raw[2 * h + 1] >> 7, raw[2 * h] >> 7, 0x06040200u);
                const uint32_t lo = __builtin_amdgcn_perm(raw[2 * h + 1], raw[2 * h], 0x06040200u);
                const uint32_t hi = __builtin_amdgcn_perm(raw[2 * h + 1], raw[2 * h], 0x07050301u);
                sd[h] = (lo & 0x7F7F7F7Fu) | (hi & 0x80808080u);
            }
            if (whole) {
                __builtin_nontemporal_store(hf_u32x2{sd[0], sd[1]},
                                            reinterpret_cast<hf_u32x2*>(side_chunk) + g);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (g * 8u + (uint32_t)i < nvalid)
                        side_chunk[(size_t)g * 8 + i] = (uint8_t)(sd[i >> 2] >> (8 * (i & 3)));
            }
        } else {
            if (FULL || g * 8u + 8u <= nvalid) {
                const hf_u32x2 v =
                    __builtin_nontemporal_load(reinterpret_cast<const hf_u32x2*>(chunk_in) + g);
                sy[2 * rr] = v.x, sy[2 * rr + 1] = v.y;
            } else {
                const uint8_t* p = reinterpret_cast<const uint8_t*>(chunk_in) + (size_t)g * 8;
                sy[2 * rr] = sy[2 * rr + 1] = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (g * 8u + (uint32_t)i < nvalid) sy[2 * rr + (i >> 2)] |= (uint32_t)p[i] << (8 * (i & 3));
            }
        }
    }
}

// LDS byte address of the encoder table entry of symbol j (0..7) of a round held in (lo, hi); the
// table sits at LDS address `tab`, a multiple of 1024
__device__ __forceinline__ uint32_t enc_addr(const uint32_t lo, const uint32_t hi, const int j,
                                             const uint32_t tab) {
    const uint32_t d = j < 4 ? lo : hi;
    const int sh = 8 * (j & 3) - 2;
    const uint32_t x = sh < 0 ? d << 2 : d >> sh;
    return (x & 0x3FCu) | tab;
}

struct EncPageLds {
    uint32_t enc[256];  // code | len << 16 by symbol; first, so that its address is a multiple of 1024
    uint32_t img[kImgWords + 8];
    uint32_t tlen[256];    // bits of string (q, l) at q * 64 + l
    uint32_t rstart[132];  // first bit of region r; [128] = B
    uint8_t r2s[256];
    uint8_t lbr[256];
    uint32_t red[8];
};

// First pass of one lane: one table lookup per symbol, the codes of each pair of symbols joined
// into one word (at most 32 bits) and kept in registers with their lengths (four to a register), so
// that the second pass -- which can only start once every lane's bit count is known -- needs no
// table.  Returns the bits of the lane's string.
//
// (sub-dword operand selection: the length is the entry's upper half, the code its lower half)
__device__ __forceinline__ uint32_t join_pair(const uint32_t ea, const uint32_t eb) {
    uint32_t c;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n\t"
        "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=&v"(c)
        : "v"(eb), "v"(ea));
    return c;
}
template <int BYTE>
__device__ __forceinline__ void pair_len_into(uint32_t& packed, const uint32_t ea, const uint32_t eb) {
    if constexpr (BYTE == 0)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(packed) : "v"(ea), "v"(eb));
    else if constexpr (BYTE == 1)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(packed) : "v"(ea), "v"(eb));
    else if constexpr (BYTE == 2)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(packed) : "v"(ea), "v"(eb));
    else
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(packed) : "v"(ea), "v"(eb));
}

template <bool FULL>
__device__ __forceinline__ uint32_t join_string(const uint32_t (&sy)[16], const uint32_t q,
                                                const int lane, const uint32_t nvalid,
                                                const uint32_t tab, uint32_t (&comb)[32],
                                                uint32_t (&clp)[8]) {
    uint32_t t = 0;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        uint32_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = lds_u32(enc_addr(sy[2 * rr], sy[2 * rr + 1], j, tab));
        if (!FULL) {
            const uint32_t i0 = ((q * 8u + (uint32_t)rr) * 64u + (uint32_t)lane) * 8u;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (i0 + (uint32_t)j >= nvalid) e[j] = 0;
        }
        clp[rr] = 0;
        comb[4 * rr + 0] = join_pair(e[0], e[1]);
        pair_len_into<0>(clp[rr], e[0], e[1]);
        comb[4 * rr + 1] = join_pair(e[2], e[3]);
        pair_len_into<1>(clp[rr], e[2], e[3]);
        comb[4 * rr + 2] = join_pair(e[4], e[5]);
        pair_len_into<2>(clp[rr], e[4], e[5]);
        comb[4 * rr + 3] = join_pair(e[6], e[7]);
        pair_len_into<3>(clp[rr], e[6], e[7]);
        t = __builtin_amdgcn_sad_u8(clp[rr], 0u, t);  // + the four lengths
    }
    return t;
}

// Second pass of one lane: its 32 joined pairs appended at bit position `pos` of its own
// direction's bit space (forward: the strings area as stored; backward: the same area mirrored, bit
// m of it being stored bit 32 Wn - 1 - m).  Whole words go out as they fill; the first and the last
// one share their word with a neighbouring string, so every word is OR-ed into its (zeroed) place.
template <bool BACK, bool IN_LDS>
__device__ __forceinline__ void emit_string(const uint32_t (&comb)[32], const uint32_t (&clp)[8],
                                            const uint32_t pos, const uint32_t wn,
                                            uint32_t* __restrict__ img_str,
                                            uint32_t* __restrict__ glob_str) {
    uint32_t hi = 0, lo = 0;  // the bits not yet written, right-aligned in hi:lo
    uint32_t ab = pos & 31u;  // their number, with the bits in front of the string's start
    int idx = BACK ? (int)wn - 1 - (int)(pos >> 5) : (int)(pos >> 5);
    auto put = [&](const uint32_t w) {
        const uint32_t v = BACK ? __builtin_bitreverse32(w) : w;
        if (IN_LDS)
            __hip_atomic_fetch_or(&img_str[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
            __hip_atomic_fetch_or(&glob_str[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        idx += BACK ? -1 : 1;
    };
#pragma unroll
    for (int p = 0; p < 32; ++p) {
        const uint32_t cl = (clp[p >> 2] >> (8 * (p & 3))) & 0xFFu;  // 0 .. 32
        const u64 acc = ((((u64)hi << 32) | (u64)lo) << cl) | (u64)comb[p];
        hi = (uint32_t)(acc >> 32);
        lo = (uint32_t)acc;
        ab += cl;
        if (ab >= 32u) {
            put(__builtin_amdgcn_alignbit(hi, lo, ab));  // bits ab - 1 .. ab - 32 of hi:lo
            ab &= 31u;
        }
    }
    if (ab != 0u) put(lo << (32u - ab));
}

// SHARED: a page without a record; `enc_tab` is then the stream codebook's table and cb is null
template <int LAYOUT, bool FULL, bool SHARED>
__device__ __forceinline__ void encode_page(EncPageLds& L, const void* chunk_in, uint8_t* side_chunk,
                                            const uint32_t* __restrict__ enc_tab,
                                            const hf_chunk_cb* __restrict__ cb, const uint32_t nvalid,
                                            uint32_t* __restrict__ payload, const uint32_t off0,
                                            const uint32_t words, uint32_t* err) {
    const int lane = hf_lane();
    const uint32_t tid = threadIdx.x;
    const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t a0 = off0 & 3u;
    const bool in_lds = a0 + words <= (uint32_t)kImgWords;  // else only the header is built in LDS
    const uint32_t tab = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(L.enc));

    uint32_t sy[16];
    load_string<LAYOUT, FULL>(chunk_in, side_chunk, q, lane, nvalid, sy);
    constexpr uint32_t hdr0 = SHARED ? 1u : kHdr;  // words in front of the listed symbols
    L.enc[tid] = enc_tab[tid];
    uint32_t mode = 0, entropy = 0;
    if (!SHARED) {
        if (tid < 64) reinterpret_cast<uint32_t*>(L.r2s)[tid] = reinterpret_cast<const uint32_t*>(cb->rank2sym)[tid];
        else if (tid < 128) reinterpret_cast<uint32_t*>(L.lbr)[tid - 64] = reinterpret_cast<const uint32_t*>(cb->len_by_rank)[tid - 64];
        if (tid == 127) {
            mode = cb->mode;
            entropy = cb->entropy_q16;
        }
    }
    {
        // the image is OR-ed into: zero what the page covers (the whole header part otherwise)
        const uint32_t upto = in_lds ? a0 + words : a0 + hdr0 + (SHARED ? 0u : 64u) + kGap;
        for (uint32_t k = tid; k * 4u < upto + 4u; k += 256u)
            reinterpret_cast<uint4*>(L.img)[k] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    // first pass: the bits of this lane's string; the record's explicit ranks
    uint32_t comb[32], clp[8];
    {
        const uint32_t t = join_string<FULL>(sy, q, lane, nvalid, tab, comb, clp);
        L.tlen[tid] = t;
        if (!SHARED) {
            const bool desc = tid > 0 && L.r2s[tid - 1] >= L.r2s[tid];
            const u64 m = __ballot(desc);
            if (lane == 0) L.red[q] = m ? q * 64u + 63u - (uint32_t)__builtin_clzll(m) : 0u;
        }
    }
    __syncthreads();

    uint32_t head = 0;
    if (!SHARED) {
        head = L.red[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) head = L.red[w] > head ? L.red[w] : head;
        head = (uint32_t)__builtin_amdgcn_readfirstlane((int)head);
    }
    const uint32_t hw = (head + 3u) / 4u;
    uint32_t* const gap = L.img + a0 + hdr0 + hw;
    if (q < 2) {
        // region lengths, their prefix sum, the length table
        const uint32_t R = L.tlen[tid] + L.tlen[tid + 128];
        uint32_t incl = R;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (q == 1) incl += hf_wave_sum_u32(L.tlen[lane] + L.tlen[lane + 128]);  // wavefront 0's regions
        L.rstart[tid] = incl - R;
        const uint32_t bit = 12u * tid, sh = bit & 31u;
        __hip_atomic_fetch_or(&gap[bit >> 5], R << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (sh > 20u)
            __hip_atomic_fetch_or(&gap[(bit >> 5) + 1], R >> (32u - sh), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        if (tid == 127) {
            L.rstart[128] = incl;
            L.img[a0] = SHARED ? (incl << 10) | kSharedFlag : head | ((mode & 1u) << 9) | (incl << 10);
            if (!SHARED) L.img[a0 + 1] = entropy;
        }
    }
    if (!SHARED) {
        // number of ranks per code length, 16-bit fields two to a word
        const uint32_t len = L.lbr[tid];
#pragma unroll
        for (uint32_t d = 1; d <= 16; ++d) {
            const uint32_t cnt = (uint32_t)__popcll(__ballot(len == d));
            if (lane == 0 && cnt)
                __hip_atomic_fetch_add(&L.img[a0 + 2 + (d - 1) / 2], cnt << (16u * ((d - 1) & 1u)),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (tid < head) reinterpret_cast<uint8_t*>(L.img + a0 + kHdr)[tid] = L.r2s[tid];
    }
    __syncthreads();

    const uint32_t B = L.rstart[128];
    // counts that were not this page's: the page would not have the size its place was cut for
    const bool fits = (SHARED ? page_words_shared(B) : page_words(head, B)) == words;
    if (!fits) {
        if (tid == 0) atomicOr(err, HF_KERR_CORRUPT);
        return;  // (the same in every thread)
    }
    const uint32_t wn = (B + 31u) / 32u;
    const uint32_t hdr_words = hdr0 + hw + kGap;
    uint32_t* const img_str = L.img + a0 + hdr_words;
    uint32_t* const glob_str = payload + off0 + hdr_words;
    if (!in_lds) {
        for (uint32_t k = tid; k < wn; k += 256u) glob_str[k] = 0u;
        __threadfence();
        __syncthreads();
    }
    {
        const uint32_t r = (q & 1u) * 64u + (uint32_t)lane;
        const uint32_t pos = q < 2 ? L.rstart[r] : wn * 32u - L.rstart[r + 1];
        if (in_lds) {
            if (q < 2) emit_string<false, true>(comb, clp, pos, wn, img_str, glob_str);
            else emit_string<true, true>(comb, clp, pos, wn, img_str, glob_str);
        } else {
            if (q < 2) emit_string<false, false>(comb, clp, pos, wn, img_str, glob_str);
            else emit_string<true, false>(comb, clp, pos, wn, img_str, glob_str);
        }
    }
    __syncthreads();

    // copy out: whole 16-byte pieces where the page covers them, single words at its two ends
    const uint32_t ncopy = in_lds ? words : hdr_words;
    uint32_t* const dst = payload + (off0 - a0);  // 16-byte aligned
    for (uint32_t g = tid; g * 4u < a0 + ncopy; g += 256u) {
        const uint32_t k0 = g * 4u;
        if (k0 >= a0 && k0 + 4u <= a0 + ncopy) {
            const uint4 v = reinterpret_cast<const uint4*>(L.img)[g];
            *reinterpret_cast<uint4*>(dst + k0) = v;
        } else {
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i)
                if (k0 + i >= a0 && k0 + i < a0 + ncopy) dst[k0 + i] = L.img[k0 + i];
        }
    }
}

// cbp: the per-page records (hf_chunk_cb[nchunks]), or the stream's codebook (hf_cb_dev) when SHARED
template <int LAYOUT, bool SHARED>
__global__ __launch_bounds__(256) void hf_k_encode_pages(
    const void* __restrict__ data, const uint64_t n, const uint32_t nchunks,
    const void* __restrict__ cbp, uint8_t* __restrict__ side, uint32_t* __restrict__ payload,
    const uint32_t* __restrict__ page_off, const u64* __restrict__ total_words, const u64 limit,
    uint32_t* err) {
    __shared__ __attribute__((aligned(1024))) EncPageLds L;
    // (the pages do not fit the payload buffer, which the scan has reported: nothing is written)
    if (*total_words > limit) return;
    // Workgroup i of a launch runs on XCD i mod 8, and each XCD has an L2 of its own: every XCD
    // takes one contiguous eighth of the pages (the grid is 8 x ceil(nchunks / 8) workgroups), so
    // that the lines two neighbouring pages share are completed in one L2 and an XCD writes one
    // contiguous part of the payload.  (Measured on config 2: 5.13 -> 4.80 ms; with pages dealt out
    // round robin the decoder is the faster one, and it keeps that order.)
    const uint32_t per_xcd = (nchunks + 7u) / 8u;
    const uint32_t c = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (c >= nchunks) return;
    const uint32_t off0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)page_off[c]);
    const uint32_t off1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)page_off[c + 1]);
    if (off1 < off0 || (u64)off1 > limit || off1 - off0 > HF_PAGE_MAX_WORDS ||
        (lds_address(L.enc) & 1023u) != 0) {  // (never with offsets of the scan)
        if (threadIdx.x == 0) atomicOr(err, HF_KERR_CORRUPT);
        return;
    }
    const size_t elem_bytes = LAYOUT == HF_LAYOUT_BF16_EXP ? 2 : 1;
    const uint64_t first = (uint64_t)c * HF_CHUNK_SYMS;
    const uint64_t left = n - first;
    const uint32_t nvalid = left < HF_CHUNK_SYMS ? (uint32_t)left : HF_CHUNK_SYMS;
    const void* chunk_in = reinterpret_cast<const uint8_t*>(data) + first * elem_bytes;
    uint8_t* side_chunk = side ? side + first : nullptr;
    const hf_chunk_cb* const cb = SHARED ? nullptr : reinterpret_cast<const hf_chunk_cb*>(cbp) + c;
    const uint32_t* const enc_tab = SHARED ? reinterpret_cast<const hf_cb_dev*>(cbp)->enc : cb->enc;
    if (nvalid == HF_CHUNK_SYMS)
        encode_page<LAYOUT, true, SHARED>(L, chunk_in, side_chunk, enc_tab, cb, nvalid, payload, off0,
                                          off1 - off0, err);
    else
        encode_page<LAYOUT, false, SHARED>(L, chunk_in, side_chunk, enc_tab, cb, nvalid, payload, off0,
                                           off1 - off0, err);
}

// ---- encoder of a stream with one codebook, in one read of the data -----------------------------
//
// Without per-page counts a page's size is only known once its symbols have been looked up, so its
// place is found while it is encoded: every page publishes its size in a descriptor as soon as the
// first pass has added up its bits, packs its strings into the LDS image, and then sums the
// descriptors in front of its own back to the nearest one that already holds an inclusive prefix
// (decoupled look-back).  The image is built without knowing the alignment of its place, so the
// copy-out reads it word by word and still writes 16-byte pieces.  The stream is the same, bit for
// bit, as the one placed from per-page counts.
//
// A page waits only for pages with lower numbers, and a GPU starts the workgroups of a
// one-dimensional grid in the order of their numbers, so the page with the lowest number among
// those not yet finished is always running and never waits: the walk cannot deadlock.  It is
// bounded all the same (one second on the 100 MHz clock, and at once when another page has given
// up): the call then reports HF_ERR_TIMEOUT and the stream is not complete.
constexpr u64 kPgAggregate = 1ull << 62;
constexpr u64 kPgPrefix = 2ull << 62;
constexpr u64 kPgValueMask = (1ull << 62) - 1ull;
constexpr u64 kPgSpinTicks = 100000000ull;
constexpr u64 kPgGaveUp = ~0ull;
#ifndef HF_PAGE_LOOK_K
#define HF_PAGE_LOOK_K 4
#endif
constexpr int kPgLookK = HF_PAGE_LOOK_K;  // descriptors per lane and pass: a pass covers 256 pages

// words in front of page c (c >= 1); kPgGaveUp after a timeout.  One wavefront, all lanes.
__device__ u64 page_lookback(const u64* desc, const uint32_t c, const int lane, uint32_t* err) {
    u64 excl = 0;
    long long idx = (long long)c - 1;
    uint32_t spins = 0;
    u64 t0 = 0;
    while (idx >= 0) {
        u64 d[kPgLookK];
#pragma unroll
        for (int k = 0; k < kPgLookK; ++k) d[k] = 0;
        for (;;) {
            // (only descriptors not yet seen valid are read again)
#pragma unroll
            for (int k = 0; k < kPgLookK; ++k) {
                const long long j = idx - 64 * k - lane;
                if ((d[k] >> 62) == 0)  // in front of page 0: prefix 0
                    d[k] = j >= 0 ? __hip_atomic_load(&desc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : kPgPrefix;
            }
            // the nearest prefix (step kf, lane lf); every descriptor nearer than it must hold a size
            int kf = kPgLookK;
            bool holes = false;
            u64 v = 0;
#pragma unroll
            for (int k = 0; k < kPgLookK; ++k) {
                const u64 inv = __ballot((d[k] >> 62) == 0);
                const u64 pre = __ballot((d[k] >> 62) == 2);
                if (kf == kPgLookK) {  // (wave-uniform)
                    u64 upto = ~0ull;
                    if (pre) {
                        kf = k;
                        const int lf = (int)__builtin_ctzll(pre);
                        upto = lf >= 63 ? ~0ull : ((2ull << lf) - 1ull);
                    }
                    holes = holes || (inv & upto) != 0;
                    v += ((upto >> lane) & 1ull) ? (d[k] & kPgValueMask) : 0ull;
                }
            }
            if (!holes) {
                excl += hf_wave_sum_u64(v);
                if (kf < kPgLookK) return excl;
                break;
            }
            if ((++spins & 15u) == 0) {
                const u64 now = __builtin_amdgcn_s_memrealtime();
                if (t0 == 0) t0 = now;
                const uint32_t seen = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (now - t0 > kPgSpinTicks || (seen & HF_KERR_TIMEOUT) != 0) {
                    if (lane == 0) atomicOr(err, HF_KERR_TIMEOUT);
                    return kPgGaveUp;
                }
            }
            __builtin_amdgcn_s_sleep(2);
        }
        idx -= 64 * kPgLookK;
    }
    return excl;
}

// The place of page c (`words` long): wavefront 0 walks back, everybody learns the result -- the
// offset in the low half, 1 in the high half if the page may be written.  A page that does not fit,
// or whose walk gave up, writes nothing (its descriptor is completed all the same, so that the
// pages behind it end too).  All 256 threads call.
__device__ __forceinline__ u64 page_place(EncPageLds& L, u64* desc, const uint32_t c,
                                          const uint32_t nchunks, const u64 limit, const uint32_t words,
                                          uint32_t* __restrict__ page_off, uint32_t* err) {
    const int lane = hf_lane();
    if (threadIdx.x < 64u) {
        // (a walk that gave up anywhere leaves sums that mean nothing: nobody writes after it)
        const uint32_t seen = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 e = c == 0 ? 0ull : page_lookback(desc, c, lane, err);
        if (lane == 0) {
            const bool gave_up = e == kPgGaveUp || (seen & HF_KERR_TIMEOUT) != 0;
            const u64 end = gave_up ? 0ull : e + (u64)words;
            if (c != 0)
                __hip_atomic_store(&desc[c], kPgPrefix | end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool fits = !gave_up && end <= limit;
            if (!gave_up && !fits) atomicOr(err, HF_KERR_OVERFLOW);
            L.red[0] = (uint32_t)e;
            L.red[1] = fits ? 1u : 0u;
            page_off[c] = fits ? (uint32_t)e : 0u;
            if (c + 1u == nchunks) {
                const uint32_t t = fits ? (uint32_t)end : 0xFFFFFFFFu;
                page_off[nchunks] = t;
                err[1] = t;  // (the call's total, beside the error word)
            }
        }
    }
    __syncthreads();
    return (u64)L.red[0] | ((u64)L.red[1] << 32);
}

// copy out: image word k goes to payload word off0 + k; 16-byte pieces of the payload where the
// page covers them whole, single words at its two ends (the image was built without knowing the
// alignment of its place, so it is read word by word)
__device__ __forceinline__ void page_copy_out(const EncPageLds& L, uint32_t* __restrict__ payload,
                                              const uint32_t off0, const uint32_t ncopy) {
    const uint32_t tid = threadIdx.x;
    const uint32_t a0 = off0 & 3u;
    uint32_t* const dst = payload + (off0 - a0);  // 16-byte aligned
    for (uint32_t g = tid; g * 4u < a0 + ncopy; g += 256u) {
        const uint32_t k0 = g * 4u;
        if (k0 >= a0 && k0 + 4u <= a0 + ncopy) {
            const uint32_t* const src = L.img + (k0 - a0);
            *reinterpret_cast<uint4*>(dst + k0) = make_uint4(src[0], src[1], src[2], src[3]);
        } else {
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i)
                if (k0 + i >= a0 && k0 + i < a0 + ncopy) dst[k0 + i] = L.img[k0 + i - a0];
        }
    }
}

// A page whose strings are too long for the LDS image (Huffman forced on data it does not
// compress): they go straight to their place off0, known by now.  Nothing was kept across the
// walk -- every lane fetches its 64 symbols once more, one at a time, and packs them with a loop
// that needs few registers (this path is rare, and its registers would otherwise set the occupancy
// of the whole kernel).  L holds the encoder table, the header, the region lengths and rstart[];
// the side bytes were written by the first pass.
template <int LAYOUT>
__device__ __forceinline__ void encode_page_big(EncPageLds& L, const void* chunk_in,
                                                const uint32_t nvalid, uint32_t* __restrict__ payload,
                                                const uint32_t off0) {
    const int lane = hf_lane();
    const uint32_t tid = threadIdx.x;
    const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    constexpr uint32_t hdr_words = 1u + kGap;
    const uint32_t B = L.rstart[128];
    const uint32_t wn = (B + 31u) / 32u;
    uint32_t* const glob_str = payload + off0 + hdr_words;
    for (uint32_t k = tid; k < wn; k += 256u) glob_str[k] = 0u;
    __threadfence();
    __syncthreads();
    const bool back = q >= 2;
    const uint32_t r = (q & 1u) * 64u + (uint32_t)lane;
    const uint32_t pos = back ? wn * 32u - L.rstart[r + 1] : L.rstart[r];
    u64 acc = 0;              // the string so far, its newest bits lowest
    uint32_t ab = pos & 31u;  // bits not yet written, with those in front of the string's start
    int idx = back ? (int)wn - 1 - (int)(pos >> 5) : (int)(pos >> 5);
    auto put = [&](const uint32_t w) {
        __hip_atomic_fetch_or(&glob_str[idx], back ? __builtin_bitreverse32(w) : w, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT);
        idx += back ? -1 : 1;
    };
#pragma unroll 1
    for (uint32_t rr = 0; rr < 8; ++rr) {
        const uint32_t i0 = ((q * 8u + rr) * 64u + (uint32_t)lane) * 8u;
#pragma unroll 1
        for (uint32_t j = 0; j < 8; ++j) {
            if (i0 + j >= nvalid) break;
            const uint32_t sym =
                LAYOUT == HF_LAYOUT_BF16_EXP
                    ? ((uint32_t)reinterpret_cast<const uint16_t*>(chunk_in)[i0 + j] >> 7) & 0xFFu
                    : (uint32_t)reinterpret_cast<const uint8_t*>(chunk_in)[i0 + j];
            const uint32_t e = L.enc[sym];
            const uint32_t cl = e >> 16;
            acc = (acc << cl) | (u64)(e & 0xFFFFu);
            ab += cl;
            if (ab >= 32u) {
                ab -= 32u;
                put((uint32_t)(acc >> ab));
            }
        }
    }
    if (ab != 0u) put((uint32_t)(acc << (32u - ab)));
    page_copy_out(L, payload, off0, hdr_words);
}

template <int LAYOUT, bool FULL>
__device__ __forceinline__ void encode_page_placed_late(
    EncPageLds& L, const void* chunk_in, uint8_t* side_chunk, const uint32_t* __restrict__ enc_tab,
    const uint32_t nvalid, uint32_t* __restrict__ payload, u64* desc, const uint32_t c,
    const uint32_t nchunks, const u64 limit, uint32_t* __restrict__ page_off, uint32_t* err) {
    const int lane = hf_lane();
    const uint32_t tid = threadIdx.x;
    const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t tab = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(L.enc));
    constexpr uint32_t hdr_words = 1u + kGap;
    constexpr uint32_t kHdrPieces = (hdr_words + 3u) / 4u;  // 16-byte pieces cleared up front

    uint32_t sy[16];
    load_string<LAYOUT, FULL>(chunk_in, side_chunk, q, lane, nvalid, sy);
    L.enc[tid] = enc_tab[tid];
    // the header word and the region lengths are OR-ed into; the strings area is cleared below,
    // once its size is known
    if (tid < kHdrPieces) reinterpret_cast<uint4*>(L.img)[tid] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    uint32_t comb[32], clp[8];
    L.tlen[tid] = join_string<FULL>(sy, q, lane, nvalid, tab, comb, clp);
    __syncthreads();

    if (q < 2) {
        // region lengths, their prefix sum, the length table (as in encode_page)
        uint32_t* const gap = L.img + 1;
        const uint32_t R = L.tlen[tid] + L.tlen[tid + 128];
        uint32_t incl = R;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (q == 1) incl += hf_wave_sum_u32(L.tlen[lane] + L.tlen[lane + 128]);
        L.rstart[tid] = incl - R;
        const uint32_t bit = 12u * tid, sh = bit & 31u;
        __hip_atomic_fetch_or(&gap[bit >> 5], R << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (sh > 20u)
            __hip_atomic_fetch_or(&gap[(bit >> 5) + 1], R >> (32u - sh), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        if (tid == 127) {
            L.rstart[128] = incl;
            L.img[0] = (incl << 10) | kSharedFlag;
            // this page's size, for the pages behind it (page 0 is its own prefix)
            const u64 w = (u64)page_words_shared(incl);
            __hip_atomic_store(&desc[c], (c == 0 ? kPgPrefix : kPgAggregate) | w, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        // wavefronts 2 and 3 add up the page's bits themselves and clear the strings area
        const uint32_t B2 = hf_wave_sum_u32(L.tlen[lane] + L.tlen[lane + 64] + L.tlen[lane + 128] +
                                            L.tlen[lane + 192]);
        const uint32_t w2 = page_words_shared(B2);
        if (w2 <= (uint32_t)kImgWords)
            for (uint32_t k = kHdrPieces + (tid - 128u); k * 4u < w2; k += 128u)
                reinterpret_cast<uint4*>(L.img)[k] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    const uint32_t B = L.rstart[128];
    const uint32_t words = page_words_shared(B);
    const uint32_t wn = (B + 31u) / 32u;
    const bool in_lds = words <= (uint32_t)kImgWords;

    if (in_lds) {
        const uint32_t r = (q & 1u) * 64u + (uint32_t)lane;
        const uint32_t pos = q < 2 ? L.rstart[r] : wn * 32u - L.rstart[r + 1];
        uint32_t* const img_str = L.img + hdr_words;
        if (q < 2) emit_string<false, true>(comb, clp, pos, wn, img_str, img_str);
        else emit_string<true, true>(comb, clp, pos, wn, img_str, img_str);
    }
    __syncthreads();
    // (one walk for both kinds of page, at a point where the joined codes are no longer needed)
    const u64 pl = page_place(L, desc, c, nchunks, limit, words, page_off, err);
    if ((pl >> 32) == 0) return;  // (the same in every thread)
    if (in_lds) page_copy_out(L, payload, (uint32_t)pl, words);
    else encode_page_big<LAYOUT>(L, chunk_in, nvalid, payload, (uint32_t)pl);
}

// desc: one cleared 8-byte descriptor per page; err[1] receives the stream's total words
template <int LAYOUT>
__global__ __launch_bounds__(256) void hf_k_encode_pages_once(
    const void* __restrict__ data, const uint64_t n, const uint32_t nchunks,
    const hf_cb_dev* __restrict__ cb, uint8_t* __restrict__ side, uint32_t* __restrict__ payload,
    uint32_t* __restrict__ page_off, u64* desc, const u64 limit, uint32_t* err) {
    __shared__ __attribute__((aligned(1024))) EncPageLds L;
    const uint32_t c = blockIdx.x;
    const size_t elem_bytes = LAYOUT == HF_LAYOUT_BF16_EXP ? 2 : 1;
    const uint64_t first = (uint64_t)c * HF_CHUNK_SYMS;
    const uint64_t left = n - first;
    const uint32_t nvalid = left < HF_CHUNK_SYMS ? (uint32_t)left : HF_CHUNK_SYMS;
    const void* chunk_in = reinterpret_cast<const uint8_t*>(data) + first * elem_bytes;
    uint8_t* side_chunk = side ? side + first : nullptr;
    if (nvalid == HF_CHUNK_SYMS)
        encode_page_placed_late<LAYOUT, true>(L, chunk_in, side_chunk, cb->enc, nvalid, payload, desc, c,
                                              nchunks, limit, page_off, err);
    else
        encode_page_placed_late<LAYOUT, false>(L, chunk_in, side_chunk, cb->enc, nvalid, payload, desc,
                                               c, nchunks, limit, page_off, err);
}

// ---- decoder ---------------------------------------------------------------------------------

// inclusive prefix sum over the 64 lanes of a wavefront, data-parallel-primitive moves only (no
// LDS crossbar round trips): Hillis-Steele inside each row of 16 lanes, then the row totals
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31
    return v;
}
// the same within rows of 16 lanes only
__device__ __forceinline__ uint32_t row_scan_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    return v;
}
__device__ __forceinline__ uint32_t lane_value(const uint32_t v, const uint32_t l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}

// The decode tables of one codebook as they lie in LDS -- and, for a stream with one codebook, in
// global memory, ready to be copied in 16-byte pieces (hf_k_page_tab_image makes that copy once
// per call).
template <int BITS>
struct DecTabLds {
    uint16_t lut[1 << BITS];  // first: the lookup ORs the table's address into the index
    uint16_t tail[512];
    uint8_t r2s[256];
    uint32_t cls[20];  // [d]: ranks with codes of at most d bits << 17 | their share of the code space
    uint32_t tail_n, tail_start;  // the direct table's span of the code space (tail_n = 0: none)
    uint32_t valid;               // (image only) the codebook is canonical and its tables are complete
    uint32_t pad;
};
static_assert(sizeof(DecTabLds<11>) % 16 == 0, "copied in 16-byte pieces");

template <int BITS>
struct DecPageLds {
    DecTabLds<BITS> tab;
    uint32_t img[kImgWords + kImgSlack];
    uint32_t present[8];
    // first bit of region r ([128] = the sum of the region lengths); once every thread has taken
    // its two values (before the last barrier in front of the symbol loop) the backward strings'
    // bit counts are handed to the forward ones through the same words
    uint32_t rstart[132];
    uint32_t red[4];
};

// Tables of one page, shared by its four wavefronts: a first level of BITS bits (entry: symbol
// << 8 | length, 0 = the code is longer) and, for the longer codes, a direct table over their part
// of the 16-bit code space when that part is small (rare symbols: the usual case), the bounds of
// each length's block otherwise.
template <int BITS, bool PAIRCHECK>
struct PageTables {
    static constexpr int kBits = BITS;
    static constexpr bool kPairCheck = PAIRCHECK;
    uint32_t lut_addr, tail_addr, cls_addr, r2s_addr;  // LDS byte addresses (wave-uniform)
    uint32_t tail_n, tail_start;

    __device__ __forceinline__ uint32_t lookup(const uint32_t top) const {
        return lds_u16(((top >> (32 - kBits)) << 1) | lut_addr);
    }
    // entry (symbol << 8 | length) of the code at the top of `top`, 0 if none matches
    __device__ __forceinline__ uint32_t long_code(const uint32_t top) const {
        const uint32_t v16 = top >> 16;
        if (tail_n != 0) {  // wave-uniform
            const uint32_t idx = v16 - tail_start;
            return idx < tail_n ? lds_u16(tail_addr + idx * 2u) : 0u;
        }
        uint32_t e = 0;
        for (uint32_t d = 16; d > (uint32_t)kBits; --d) {  // the lowest matching block wins
            const uint32_t lo = lds_u32(cls_addr + (d - 1u) * 4u), hi = lds_u32(cls_addr + d * 4u);
            if (v16 < (hi & 0x1FFFFu)) {
                const uint32_t rank = (lo >> 17) + ((v16 - (lo & 0x1FFFFu)) >> (16u - d));
                e = (lds_u8(r2s_addr + (rank & 255u)) << 8) | d;
            }
        }
        return v16 >= (lds_u32(cls_addr + (uint32_t)kBits * 4u) & 0x1FFFFu) ? e : 0u;
    }
};

// One lane's walk through its own string.  The lane keeps two consecutive words of the string (wc,
// wx) and the number s of bits of wx that do NOT yet belong to the next 32: those 32 bits are one
// funnel shift of the pair, ({wc, wx} >> s), with s in 0..31 (s = 0: wc is used up and the bits are
// wx itself, which is why the count runs 1..32 bits of wc consumed and never 0).  A pair of symbols
// takes at most 32 bits, so after it at most one new word moves in -- wc = wx, wx = the word that was
// requested a pair earlier, whose LDS latency thus lies beside the two table lookups and not in
// front of them.  No 64-bit shifts, no window to refill.  Inside a pair the second lookup shifts
// the 32 bits alone (at least 16 of them remain).  Backward strings walk down through the words
// and take each word bit-reversed.
//
// The 64 symbols of string (q, lane) of a full page whose image is in LDS.  bitpos: forward, the
// string's first bit; backward, the bit behind its first one (the region's end).  Returns the bits
// consumed.
template <int LAYOUT, bool BACK, class TAB>
__device__ __forceinline__ uint32_t decode_string(const uint32_t img_str_addr, const uint32_t bitpos,
                                                  const uint32_t q, const int lane, const TAB& T,
                                                  const uint8_t* __restrict__ side_chunk,
                                                  void* __restrict__ out_chunk, uint32_t* err) {
    uint32_t wc, wx, wp;
    int s;
    if (BACK) {
        const uint32_t a = img_str_addr + (bitpos >> 5) * 4u;
        s = (int)(bitpos & 31u);
        wc = __builtin_bitreverse32(lds_u32(a));
        wx = __builtin_bitreverse32(lds_u32(a - 4u));
        wp = a - 8u;
    } else {
        const int before = (int)bitpos - 1;  // (-1: the word in front of the strings, never used)
        const uint32_t a = img_str_addr + (uint32_t)((before >> 5) * 4);
        s = (int)(~(uint32_t)before & 31u);
        wc = lds_u32(a);
        wx = lds_u32(a + 4u);
        wp = a + 8u;
    }
    const uint32_t wp0 = wp;
    const int s0 = s;
    uint32_t wn = lds_u32(wp);  // the next word, requested ahead
    const hf_gmem_u8* const su = hf_uniform_ptr(side_chunk);
    hf_gmem_u8* const ou = hf_uniform_ptr(out_chunk);
    const uint32_t lane8 = (uint32_t)lane * 8u, lane16 = (uint32_t)lane * 16u;
    hf_u32x2 sd[8];
    if (LAYOUT == HF_LAYOUT_BF16_EXP) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
            sd[rr] = __builtin_nontemporal_load(
                reinterpret_cast<const hf_gmem_u32x2*>(su + (q * 8u + (uint32_t)rr) * 512u + lane8));
    }
    auto long_one = [&](const uint32_t top) -> uint32_t {
        uint32_t e = T.long_code(top);
        if (e == 0) {  // no code matches: flag it and keep moving
            atomicOr(err, HF_KERR_CORRUPT);
            e = 1;
        }
        return e;
    };
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const uint32_t r = q * 8u + (uint32_t)rr;
        uint32_t e[8];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            const uint32_t hi = __builtin_amdgcn_alignbit(wc, wx, (uint32_t)s);
            // (a first-level entry of 0 is a longer code: a lane that met one has not moved, its
            // second lookup returns the same entry, and one test can cover both)
            uint32_t e0 = T.lookup(hi), e1;
            if constexpr (TAB::kPairCheck) {
                e1 = T.lookup(hi << (e0 & 31u));
                if (__builtin_expect(__ballot(e1 == 0u) != 0ull, 0)) {
                    if (e1 == 0u) {
                        if (e0 == 0u) {
                            e0 = long_one(hi);
                            e1 = T.lookup(hi << (e0 & 31u));
                        }
                        if (e1 == 0u) e1 = long_one(hi << (e0 & 31u));
                    }
                }
            } else {
                if (__builtin_expect(e0 == 0u, 0)) e0 = long_one(hi);
                const uint32_t t1 = hi << (e0 & 31u);
                e1 = T.lookup(t1);
                if (__builtin_expect(e1 == 0u, 0)) e1 = long_one(t1);
            }
            e[2 * jp] = e0;
            e[2 * jp + 1] = e1;
            s -= (int)((e0 & 0xFFu) + (e1 & 0xFFu));  // 2 .. 32 bits
            if (s < 0) {
                wc = wx;
                wx = BACK ? __builtin_bitreverse32(wn) : wn;
                s += 32;
                wp += BACK ? (uint32_t)-4 : 4u;
                wn = lds_u32(wp);
            }
        }
        if (LAYOUT == HF_LAYOUT_BF16_EXP) {
            // two bf16 per dword (see decode_chunk_full in hf_decode.hip)
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t t = __builtin_amdgcn_perm(0u, i < 2 ? sd[rr].x : sd[rr].y,
                                                         (i & 1) ? 0x0C030C02u : 0x0C010C00u);
                const uint32_t expo = __builtin_amdgcn_perm(e[2 * i + 1], e[2 * i], 0x0C050C01u);
                uint32_t sm;
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(sm) : "v"(t & 0x00800080u), "s"(255u), "v"(t));
                asm("v_lshl_or_b32 %0, %1, 7, %2" : "=v"(o[i]) : "v"(expo), "v"(sm));
            }
            __builtin_nontemporal_store(hf_u32x4{o[0], o[1], o[2], o[3]},
                                        reinterpret_cast<hf_gmem_u32x4*>(ou + r * 1024u + lane16));
        } else {
            const uint32_t a = __builtin_amdgcn_perm(e[1], e[0], 0x0C0C0501u);
            const uint32_t b = __builtin_amdgcn_perm(e[3], e[2], 0x0C0C0501u);
            const uint32_t c = __builtin_amdgcn_perm(e[5], e[4], 0x0C0C0501u);
            const uint32_t d = __builtin_amdgcn_perm(e[7], e[6], 0x0C0C0501u);
            __builtin_nontemporal_store(hf_u32x2{a | (b << 16), c | (d << 16)},
                                        reinterpret_cast<hf_gmem_u32x2*>(ou + r * 512u + lane8));
        }
    }
    // 32 bits per word moved in, and what the count of unused bits of wx lost
    const uint32_t moved = (BACK ? wp0 - wp : wp - wp0) / 4u;
    return moved * 32u + (uint32_t)(s0 - s);
}

// The same for pages that are not full or whose strings stayed in global memory: symbol by symbol,
// any direction, words fetched through `word(i)` (index into the strings area; zero outside it).
template <int LAYOUT, class TAB, class FETCH>
__device__ __forceinline__ uint32_t decode_string_slow(const bool back, const uint32_t bitpos, const bool empty,
                                       const uint32_t q, const int lane, const uint32_t nvalid,
                                       const TAB& T, FETCH word, const uint8_t* __restrict__ side_chunk,
                                       void* __restrict__ out_chunk, uint32_t* err) {
    if (empty) return 0u;  // a string without symbols (only past the end of a partial page)
    int wi = (int)(bitpos >> 5);
    const uint32_t sh = back ? 31u - (bitpos & 31u) : (bitpos & 31u);
    auto next = [&]() {
        const uint32_t w = word(wi);
        wi += back ? -1 : 1;
        return back ? __builtin_bitreverse32(w) : w;
    };
    u64 win = (u64)next() << 32;
    win |= (u64)next();
    win <<= sh;
    int room = (int)sh - 32;
    uint32_t taken = 2;
    for (uint32_t rr = 0; rr < 8; ++rr) {
        const uint32_t g = (q * 8u + rr) * 64u + (uint32_t)lane;
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t i = g * 8u + j;
            if (i >= nvalid) continue;
            if ((j & 1u) == 0 && room > 0) {
                win |= (u64)next() << (room & 63);
                room -= 32;
                ++taken;
            }
            const uint32_t top = (uint32_t)(win >> 32);
            uint32_t e = T.lookup(top);
            if (e == 0u) e = T.long_code(top);
            if (e == 0u) {
                atomicOr(err, HF_KERR_CORRUPT);
                e = 1;
            }
            win <<= (e & 63u);
            room += (int)(e & 0xFFu);
            const uint32_t s = e >> 8;
            if (LAYOUT == HF_LAYOUT_BF16_EXP) {
                const uint32_t b = side_chunk[i];
                reinterpret_cast<uint16_t*>(out_chunk)[i] =
                    (uint16_t)(((b & 0x80u) << 8) | ((s & 0xFFu) << 7) | (b & 0x7Fu));
            } else {
                reinterpret_cast<uint8_t*>(out_chunk)[i] = (uint8_t)s;
            }
        }
    }
    return taken * 32u - (uint32_t)(32 - room) - sh;
}

// ranks with at most d bits (<< 17) and the code space they fill, for d = 0 .. 16, from the number
// of ranks per length (lane d - 1 holds that of length d); lanes 0 .. 16 of wavefront 0 write
// cls[].  Returns the inclusive scan of the counts (lane d - 1: ranks with at most d bits); total
// and kraft are the scans' last values.
__device__ __forceinline__ uint32_t class_bounds(const uint32_t cnt, const int lane, const uint32_t q,
                                                 uint32_t* cls, uint32_t& total, uint32_t& kraft) {
    const uint32_t cumv = row_scan_incl(cnt);
    const uint32_t csv = row_scan_incl(cnt << (15 - (lane & 15)));
    total = lane_value(cumv, 15);
    kraft = lane_value(csv, 15);
    if (q == 0 && lane <= 16) {
        // cls[d] for d = 0 .. 16 (lane d takes the values of lane d - 1)
        const uint32_t pc = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)((cumv << 17) | csv), 0x111,
                                                                  0xF, 0xF, true);
        cls[lane] = lane == 16 ? (total << 17) | kraft : pc;
    }
    return cumv;
}

// Thread `tid` fills the span of rank tid (code length my_len, symbol r2s[tid]) in the first level
// or in the direct table of the longer codes; cls[] and r2s[] are complete, the tables zeroed.
// Returns the rank's left-aligned 16-bit code.
template <int BITS>
__device__ __forceinline__ uint32_t fill_tables(DecTabLds<BITS>& D, const uint32_t tid,
                                                const uint32_t my_len, const uint32_t kraft,
                                                uint32_t& tail_n, uint32_t& tail_start) {
    const uint32_t pc = D.cls[my_len - 1];
    const uint32_t my_start = (pc & 0x1FFFFu) + ((tid - (pc >> 17)) << (16u - my_len));
    const uint32_t val = ((uint32_t)D.r2s[tid] << 8) | my_len;
    const uint32_t v2 = val | (val << 16);
    // (a code's span starts at a multiple of its size: the widest aligned store that fits)
    auto fill = [&](uint16_t* to, const uint32_t at, const uint32_t span) {
        if (span >= 8) {
            for (uint32_t i = 0; i < span; i += 8)
                *reinterpret_cast<uint4*>(to + at + i) = make_uint4(v2, v2, v2, v2);
        } else if (span == 4) {
            *reinterpret_cast<uint2*>(to + at) = make_uint2(v2, v2);
        } else if (span == 2) {
            *reinterpret_cast<uint32_t*>(to + at) = v2;
        } else {
            to[at] = (uint16_t)val;
        }
    };
    tail_start = (uint32_t)__builtin_amdgcn_readfirstlane((int)(D.cls[BITS] & 0x1FFFFu));
    const uint32_t tail_span = kraft - tail_start;
    tail_n = tail_span <= 512u ? tail_span : 0u;
    if (my_len <= (uint32_t)BITS)
        fill(D.lut, my_start >> (16 - BITS), 1u << (BITS - my_len));
    else if (tail_n != 0)
        fill(D.tail, my_start - tail_start, 1u << (16u - my_len));
    return my_start;
}

// The table image of a stream-wide codebook: ranks from the left-aligned codes (canonical codes
// ascend with the rank), ranks per length, then the fill every page decoder does.  valid = 0 when
// the codes are not the canonical ones of their lengths (an uploaded table of another shape): the
// page decoder then reports every page.
template <int BITS>
__global__ __launch_bounds__(256) void hf_k_page_tab_image(const hf_cb_dev* __restrict__ cb,
                                                           DecTabLds<BITS>* __restrict__ out) {
    __shared__ __attribute__((aligned(2 << BITS))) DecTabLds<BITS> D;
    __shared__ uint32_t key[256];
    __shared__ uint32_t cnt[16];
    __shared__ uint8_t lbr[256];
    __shared__ uint32_t bad;
    const int lane = hf_lane();
    const uint32_t tid = threadIdx.x;
    const uint32_t q = tid >> 6;
#pragma unroll
    for (int j = 0; j < (2 << BITS) / 16 / 256; ++j)
        reinterpret_cast<uint4*>(D.lut)[tid + 256 * j] = make_uint4(0, 0, 0, 0);
    if (tid < 64) reinterpret_cast<uint4*>(D.tail)[tid] = make_uint4(0, 0, 0, 0);
    if (tid < 16) cnt[tid] = 0;
    if (tid == 0) bad = 0;
    const uint32_t e = cb->enc[tid];
    const uint32_t len = e >> 16, code = e & 0xFFFFu;
    const bool used = len >= 1 && len <= HF_MAX_CODE_LEN;
    const uint32_t mine = used ? code << (16u - len) : 0x10000u + tid;  // unused symbols rank last
    key[tid] = mine;
    D.r2s[tid] = 0;
    lbr[tid] = 0;
    __syncthreads();
    uint32_t rank = 0;
    for (uint32_t j = 0; j < 256; ++j) rank += key[j] < mine ? 1u : 0u;
    if (used) {
        // (two symbols with one code would take one rank: caught by the start check below)
        D.r2s[rank] = (uint8_t)tid;
        lbr[rank] = (uint8_t)len;
        atomicAdd(&cnt[len - 1], 1u);
    }
    __syncthreads();
    uint32_t total, kraft;
    class_bounds(lane < 16 ? cnt[lane & 15] : 0u, lane, q, D.cls, total, kraft);
    __syncthreads();
    uint32_t tail_n = 0, tail_start = 0;
    const uint32_t my_len = lbr[tid];
    if (kraft <= 65536u && tid < total) {
        // the canonical code of rank tid must be the symbol's own code: only then is the span the
        // fill writes known to lie inside the tables
        const uint32_t pc = D.cls[my_len ? my_len - 1 : 0];
        const uint32_t start = (pc & 0x1FFFFu) + ((tid - (pc >> 17)) << (16u - my_len));
        if (my_len == 0 || tid < (pc >> 17) || start != key[D.r2s[tid]])
            atomicOr(&bad, 1u);
        else
            fill_tables<BITS>(D, tid, my_len, kraft, tail_n, tail_start);
    }
    __syncthreads();
    if (tid == 0) {
        tail_start = D.cls[BITS] & 0x1FFFFu;
        const uint32_t span = kraft - tail_start;
        D.tail_n = kraft <= 65536u && span <= 512u ? span : 0u;
        D.tail_start = tail_start;
        D.valid = (kraft <= 65536u && total >= 1u && bad == 0u) ? 1u : 0u;
        D.pad = 0;
    }
    __syncthreads();
    constexpr uint32_t kPieces = sizeof(DecTabLds<BITS>) / 16;
    for (uint32_t g = tid; g < kPieces; g += 256u)
        reinterpret_cast<uint4*>(out)[g] = reinterpret_cast<const uint4*>(&D)[g];
}

// One workgroup per page.  SHARED: the pages carry no record and `tabimg` holds the tables of the
// stream's codebook.
template <int LAYOUT, int BITS, bool PAIRCHECK, bool SHARED>
__global__ __launch_bounds__(256) void hf_k_decode_pages(
    const uint32_t* __restrict__ payload, const uint32_t payload_words,
    const uint32_t* __restrict__ page_off, const uint8_t* __restrict__ side, const uint64_t n,
    const uint32_t nchunks, const uint32_t* __restrict__ page_ids,
    const uint32_t* __restrict__ out_slots, const uint32_t n_slots, void* __restrict__ out,
    const DecTabLds<BITS>* __restrict__ tabimg, uint32_t* err) {
    typedef PageTables<BITS, PAIRCHECK> Tab;
    __shared__ __attribute__((aligned(2 << BITS))) DecPageLds<BITS> L;
    constexpr uint32_t hdr0 = SHARED ? 1u : kHdr;
    const int lane = hf_lane();
    const uint32_t tid = threadIdx.x;
    const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t k = blockIdx.x;
    auto bad = [&]() {
        if (tid == 0) atomicOr(err, HF_KERR_CORRUPT);
    };
    // work item k: page page_ids[k] (or k) to page out_slots[k] (or k, or its own place) of `out`
    const uint32_t c = page_ids ? (uint32_t)__builtin_amdgcn_readfirstlane((int)page_ids[k]) : k;
    const uint32_t slot = out_slots ? (uint32_t)__builtin_amdgcn_readfirstlane((int)out_slots[k]) : k;
    if (c >= nchunks || (out_slots && slot >= n_slots)) return bad();
    const uint32_t off0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)page_off[c]);
    const uint32_t off1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)page_off[c + 1]);
    const uint64_t first = (uint64_t)c * HF_CHUNK_SYMS;
    const uint64_t left = n - first;
    const uint32_t nvalid = left < HF_CHUNK_SYMS ? (uint32_t)left : HF_CHUNK_SYMS;
    if (off1 < off0 || off1 > payload_words || off1 - off0 < hdr0 + kGap ||
        off1 - off0 > HF_PAGE_MAX_WORDS || (page_ids && nvalid != HF_CHUNK_SYMS))
        return bad();
    const uint32_t words = off1 - off0;
    const uint32_t a0 = off0 & 3u;
    const uint32_t base = off0 - a0;
    const bool in_lds = a0 + words <= (uint32_t)kImgWords;
    // the page (its header part when it is too large for the image) on its way to LDS, 16 bytes
    // per thread and step, all steps requested before the first is used; the buffer's last, partial
    // piece (a buffer whose size is not a multiple of 16 bytes) is fetched word by word
    constexpr int kSteps = (kImgWords + 1023) / 1024;
    const uint32_t nload = in_lds ? a0 + words : a0 + hdr0 + (SHARED ? 0u : 64u) + kGap;
    const uint32_t whole = payload_words & ~3u;
    uint4 v[kSteps];
#pragma unroll
    for (int j = 0; j < kSteps; ++j) {
        const uint32_t at = (tid + 256u * (uint32_t)j) * 4u;
        v[j] = make_uint4(0, 0, 0, 0);
        if (at < nload) {
            if (__builtin_expect(base + at < whole, 1)) {
                const hf_u32x4 t =
                    __builtin_nontemporal_load(reinterpret_cast<const hf_u32x4*>(payload + base + at));
                v[j] = make_uint4(t.x, t.y, t.z, t.w);
            } else {
                const uint32_t i = base + at;
                if (i < payload_words) v[j].x = payload[i];
                if (i + 1 < payload_words) v[j].y = payload[i + 1];
                if (i + 2 < payload_words) v[j].z = payload[i + 2];
            }
        }
    }
    if (SHARED) {
        // the stream's tables, ready-made
        constexpr uint32_t kPieces = sizeof(DecTabLds<BITS>) / 16;
#pragma unroll
        for (uint32_t j = 0; j < (kPieces + 255u) / 256u; ++j) {
            const uint32_t g = tid + 256u * j;
            if (g < kPieces) reinterpret_cast<uint4*>(&L.tab)[g] = reinterpret_cast<const uint4*>(tabimg)[g];
        }
    } else {
        // cleared tables: first-level entries no short code covers say "longer", tail entries no
        // code covers match nothing
#pragma unroll
        for (int j = 0; j < (2 << BITS) / 16 / 256; ++j)
            reinterpret_cast<uint4*>(L.tab.lut)[tid + 256 * j] = make_uint4(0, 0, 0, 0);
        if (tid < 64) reinterpret_cast<uint4*>(L.tab.tail)[tid] = make_uint4(0, 0, 0, 0);
        if (tid < 8) L.present[tid] = 0u;
    }
    reinterpret_cast<uint4*>(L.img)[tid] = v[0];
    __syncthreads();

    // ---- the header (and the record) ----
    const uint32_t hdr = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.img[a0]);
    const uint32_t head = SHARED ? 0u : hdr & 0x1FFu;
    const uint32_t B = SHARED ? (hdr >> 10) & 0x7FFFFu : hdr >> 10;
    if (SHARED) {
        if ((hdr >> 29) != 1u || (hdr & 0x3FFu) != 0u || page_words_shared(B) != words || L.tab.valid != 1u)
            return bad();
    } else {
        if ((hdr >> 29) != 0u || head > 255u || page_words(head, B) != words) return bad();
    }
    const uint32_t hw = (head + 3u) / 4u;
    uint32_t my_len = 1, kraft = 0;
    if (!SHARED) {
        // Lengths never decrease with the rank and the codes of one length tile the left-aligned
        // 16-bit code space one after the other.  Lane d - 1 takes length d: ranks with at most d
        // bits (cum) and the code space they fill (cs), by two scans over 16 lanes; rank tid then
        // has the smallest length whose cum exceeds it -- one compare against each of 15 scalars.
        const uint32_t cnt = lane < 16 ? reinterpret_cast<const uint16_t*>(L.img + a0 + 2)[lane] : 0u;
        uint32_t total;
        const uint32_t cumv = class_bounds(cnt, lane, q, L.tab.cls, total, kraft);
        if (total != 256u || kraft > 65536u) return bad();
#pragma unroll
        for (uint32_t d = 1; d <= 15; ++d) my_len += tid >= lane_value(cumv, d - 1) ? 1u : 0u;
        if (tid < head) {
            const uint32_t s = reinterpret_cast<const uint8_t*>(L.img + a0 + kHdr)[tid];
            L.tab.r2s[tid] = (uint8_t)s;
            __hip_atomic_fetch_or(&L.present[s >> 5], 1u << (s & 31u), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    // region lengths and their prefix sum (wavefronts 0 and 1: one region per thread; wavefront 1
    // adds up wavefront 0's lengths itself)
    uint32_t R = 0;
    if (q < 2) {
        const uint32_t* gap = L.img + a0 + hdr0 + hw;
        auto region = [&](const uint32_t r) {
            const uint32_t bit = 12u * r;
            return __builtin_amdgcn_alignbit(gap[(bit >> 5) + 1], gap[bit >> 5], bit & 31u) & 0xFFFu;
        };
        R = region(tid);
        uint32_t incl = wave_scan_incl(R);
        if (q == 1) incl += lane_value(wave_scan_incl(region((uint32_t)lane)), 63);
        L.rstart[tid] = incl - R;
        if (tid == 127) L.rstart[128] = incl;
    }
    if (!SHARED) __syncthreads();  // (present[] complete)

    if (!SHARED) {
        // The ranks behind the listed ones: the symbols not listed, ascending.  Symbol s = tid is
        // not listed <=> its bit of `present` is clear, and then its rank is head + s - (listed
        // symbols below s).  Lane i < 8 counts word i; a scan gives the counts below each word.
        const uint32_t pcv = row_scan_incl(lane < 8 ? (uint32_t)__popc(L.present[lane & 7]) : 0u);
        if (lane_value(pcv, 7) != head) return bad();  // a symbol listed twice
        const uint32_t w = tid >> 5;                   // 2 q or 2 q + 1
        const uint32_t before = q == 0 ? 0u : lane_value(pcv, 2u * q - 1u);
        const uint32_t mid = lane_value(pcv, 2u * q) - before;  // listed symbols in word 2 q
        const uint32_t mine = L.present[w];
        const uint32_t below = before + (lane >= 32 ? mid : 0u) +
                               (uint32_t)__popc(mine & ((1u << (tid & 31u)) - 1u));
        if (!((mine >> (tid & 31u)) & 1u)) L.tab.r2s[head + tid - below] = (uint8_t)tid;
    }
    // the rest of the image
#pragma unroll
    for (int j = 1; j < kSteps; ++j) {
        const uint32_t g = tid + 256u * (uint32_t)j;
        if (g * 4u < nload) reinterpret_cast<uint4*>(L.img)[g] = v[j];
    }
    __syncthreads();

    // ---- tables ----
    Tab T;
    if (SHARED) {
        T.tail_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.tab.tail_n);
        T.tail_start = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.tab.tail_start);
    } else {
        fill_tables<BITS>(L.tab, tid, my_len, kraft, T.tail_n, T.tail_start);
    }
    T.lut_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(L.tab.lut));
    T.tail_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(L.tab.tail));
    T.cls_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(L.tab.cls));
    T.r2s_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(L.tab.r2s));
    const uint32_t r = (q & 1u) * 64u + (uint32_t)lane;
    const uint32_t s0 = L.rstart[r], s1 = L.rstart[r + 1];
    const bool sums = L.rstart[128] == B;
    __syncthreads();
    if (!sums) return bad();  // the regions do not add up to the page's bits

    // ---- the four sub-decoders ----
    const size_t elem_bytes = LAYOUT == HF_LAYOUT_BF16_EXP ? 2 : 1;
    const uint8_t* side_chunk = side ? side + first : nullptr;
    const uint64_t out_first = page_ids ? (uint64_t)slot * HF_CHUNK_SYMS : first;
    void* out_chunk = reinterpret_cast<uint8_t*>(out) + out_first * elem_bytes;
    const uint32_t hdr_words = hdr0 + hw + kGap;
    uint32_t used;
    // (a full page's strings have 64 bits at least; a region shorter than two of them is damage,
    // and the whole wavefront then goes the careful way)
    const bool fast = in_lds && nvalid == HF_CHUNK_SYMS && __ballot(s1 - s0 < 128u) == 0ull;
    if (fast) {
        const uint32_t str_addr = lds_address(L.img + a0 + hdr_words);
        if (q < 2)
            used = decode_string<LAYOUT, false>(str_addr, s0, q, lane, T, side_chunk, out_chunk, err);
        else
            used = decode_string<LAYOUT, true>(str_addr, s1, q, lane, T, side_chunk, out_chunk, err);
    } else {
        const uint32_t wn = (B + 31u) / 32u;
        const uint32_t* gstr = payload + off0 + hdr_words;
        const uint32_t* lstr = L.img + a0 + hdr_words;
        auto word = [&](const int i) -> uint32_t {
            if (i < 0 || (uint32_t)i >= wn) return 0u;
            return in_lds ? lstr[i] : gstr[i];
        };
        const bool back = q >= 2;
        // the string's first symbol index tells whether it has any symbol at all
        const bool empty = (q * 8u * 64u + (uint32_t)lane) * 8u >= nvalid;
        if (!empty && s1 == s0) atomicOr(err, HF_KERR_CORRUPT);  // symbols, but no bits for them
        used = decode_string_slow<LAYOUT>(back, back ? s1 - 1u : s0, empty || s1 == s0, q, lane, nvalid, T,
                                          word, side_chunk, out_chunk, err);
    }
    if (q >= 2) L.rstart[tid - 128] = used;
    __syncthreads();
    // the two strings of a region meet exactly
    if (q < 2 && used + L.rstart[tid] != R) atomicOr(err, HF_KERR_CORRUPT);
}

}  // namespace

// sizes -> page offsets (d_page_off, nchunks + 1 values, the last one the total) -> pages.
// cbs: the per-page records; or, shared_cb given, null (pages without records, one codebook)
void hf_launch_encode_pages(const void* d_data, uint64_t n, int layout, const hf_chunk_cb* cbs,
                            const hf_cb_dev* shared_cb, const uint32_t* d_freq_chunks, uint8_t* d_side,
                            uint32_t* d_payload, uint64_t cap_words, uint32_t* d_page_off,
                            unsigned long long* d_scan_aux, uint32_t* d_err, hipStream_t s) {
    const uint64_t nchunks = (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS;
    if (nchunks == 0) return;
    const uint64_t m = nchunks + 1;
    if (shared_cb)
        hipLaunchKernelGGL(hf_k_page_sizes_shared, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, s,
                           d_freq_chunks, shared_cb, (uint32_t)nchunks, d_page_off);
    else
        hipLaunchKernelGGL(hf_k_page_sizes, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, s, d_freq_chunks,
                           cbs, (uint32_t)nchunks, d_page_off);
    const u64 limit = cap_words < 0xFFFFFFFFull ? cap_words : 0xFFFFFFFFull;
    hf_launch_scan_u32(d_page_off, m, d_scan_aux, limit, d_err, s);
    const u64* d_total = d_scan_aux + (m + 1023) / 1024;
    const void* cbp = shared_cb ? static_cast<const void*>(shared_cb) : static_cast<const void*>(cbs);
    const bool wide = layout == HF_LAYOUT_BF16_EXP;
    const unsigned grid = (unsigned)((nchunks + 7) / 8 * 8);  // (see the kernel: an eighth per XCD)
#define HF_LAUNCH_ENC(LAY, SH)                                                                      \
    hipLaunchKernelGGL((hf_k_encode_pages<LAY, SH>), dim3(grid), dim3(256), 0, s, d_data,             \
                       n, (uint32_t)nchunks, cbp, wide ? d_side : (uint8_t*)nullptr, d_payload,        \
                       d_page_off, d_total, limit, d_err)
    if (wide && shared_cb) HF_LAUNCH_ENC(HF_LAYOUT_BF16_EXP, true);
    else if (wide) HF_LAUNCH_ENC(HF_LAYOUT_BF16_EXP, false);
    else if (shared_cb) HF_LAUNCH_ENC(HF_LAYOUT_U8, true);
    else HF_LAUNCH_ENC(HF_LAYOUT_U8, false);
#undef HF_LAUNCH_ENC
}

// Pages without records placed while they are encoded (no per-page counts): d_desc holds one
// CLEARED 8-byte descriptor per page; the total lands in d_err[1] and in d_page_off[nchunks].
void hf_launch_encode_pages_once(const void* d_data, uint64_t n, int layout, const hf_cb_dev* shared_cb,
                                 uint8_t* d_side, uint32_t* d_payload, uint64_t cap_words,
                                 uint32_t* d_page_off, unsigned long long* d_desc, uint32_t* d_err,
                                 hipStream_t s) {
    const uint64_t nchunks = (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS;
    if (nchunks == 0) return;
    const u64 limit = cap_words < 0xFFFFFFFFull ? cap_words : 0xFFFFFFFFull;
    if (layout == HF_LAYOUT_BF16_EXP)
        hipLaunchKernelGGL(hf_k_encode_pages_once<HF_LAYOUT_BF16_EXP>, dim3((unsigned)nchunks), dim3(256),
                           0, s, d_data, n, (uint32_t)nchunks, shared_cb, d_side, d_payload, d_page_off,
                           d_desc, limit, d_err);
    else
        hipLaunchKernelGGL(hf_k_encode_pages_once<HF_LAYOUT_U8>, dim3((unsigned)nchunks), dim3(256), 0, s,
                           d_data, n, (uint32_t)nchunks, shared_cb, (uint8_t*)nullptr, d_payload,
                           d_page_off, d_desc, limit, d_err);
}

size_t hf_page_tab_image_bytes(void) { return sizeof(DecTabLds<11>); }

// shared_cb given: pages without records; d_tab_image (hf_page_tab_image_bytes) receives the tables
// of that codebook first
void hf_launch_decode_pages(const uint32_t* d_payload, uint64_t payload_words,
                            const uint32_t* d_page_off, const uint8_t* d_side, uint64_t n,
                            int layout, const hf_cb_dev* shared_cb, void* d_tab_image,
                            const uint32_t* d_page_ids, const uint32_t* d_out_slots,
                            uint64_t n_slots, uint64_t n_ids, void* d_out, uint32_t* d_err,
                            hipStream_t s) {
    const uint64_t nchunks = (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS;
    const uint64_t n_work = d_page_ids ? n_ids : nchunks;
    if (n_work == 0) return;
    // (page offsets are 32-bit word indices: nothing behind 2^32 words is ever addressed)
    const uint32_t pw32 = payload_words > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)payload_words;
    DecTabLds<11>* const img = static_cast<DecTabLds<11>*>(d_tab_image);
    if (shared_cb)
        hipLaunchKernelGGL(hf_k_page_tab_image<11>, dim3(1), dim3(256), 0, s, shared_cb, img);
    const bool wide = layout == HF_LAYOUT_BF16_EXP;
#define HF_LAUNCH_DEC(LAY, SH)                                                                       \
    hipLaunchKernelGGL((hf_k_decode_pages<LAY, 11, false, SH>), dim3((unsigned)n_work), dim3(256), 0, s, \
                       d_payload, pw32, d_page_off, wide ? d_side : (const uint8_t*)nullptr, n,          \
                       (uint32_t)nchunks, d_page_ids, d_out_slots, (uint32_t)n_slots, d_out,             \
                       (const DecTabLds<11>*)img, d_err)
    if (wide && shared_cb) HF_LAUNCH_DEC(HF_LAYOUT_BF16_EXP, true);
    else if (wide) HF_LAUNCH_DEC(HF_LAYOUT_BF16_EXP, false);
    else if (shared_cb) HF_LAUNCH_DEC(HF_LAYOUT_U8, true);
    else HF_LAUNCH_DEC(HF_LAYOUT_U8, false);
#undef HF_LAUNCH_DEC
}
