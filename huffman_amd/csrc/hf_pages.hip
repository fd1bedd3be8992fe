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

// Second pass of one lane: its 64 codes appended to the bit position `pos` of its own direction's
// bit space (forward: the strings area as stored; backward: the same area mirrored, bit m of it
// being stored bit 32 Wn - 1 - m).  Whole words go out as they fill; the first and the last one
// share their word with a neighbouring string, so every word is OR-ed into its (zeroed) place.
template <bool FULL, bool BACK, bool IN_LDS>
__device__ __forceinline__ void emit_string(const uint32_t (&sy)[16], const uint32_t q, const int lane,
                                            const uint32_t nvalid, const uint32_t tab,
                                            const uint32_t pos, const uint32_t wn,
                                            uint32_t* __restrict__ img_str,
                                            uint32_t* __restrict__ glob_str) {
    u64 acc = 0;
    uint32_t ab = pos & 31u;
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
    for (int rr = 0; rr < 8; ++rr) {
        uint32_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = lds_u32(enc_addr(sy[2 * rr], sy[2 * rr + 1], j, tab));
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            uint32_t ea = e[2 * jp], eb = e[2 * jp + 1];
            if (!FULL) {
                const uint32_t i0 = ((q * 8u + (uint32_t)rr) * 64u + (uint32_t)lane) * 8u + 2u * jp;
                if (i0 >= nvalid) ea = 0;
                if (i0 + 1 >= nvalid) eb = 0;
            }
            const uint32_t la = ea >> 16, lb = eb >> 16, cl = la + lb;
            acc = (acc << cl) | (u64)(((ea & 0xFFFFu) << lb) | (eb & 0xFFFFu));
            ab += cl;
            if (ab >= 32u) {
                put((uint32_t)(acc >> (ab - 32u)));
                ab -= 32u;
            }
        }
    }
    if (ab != 0u) put((uint32_t)(acc << (32u - ab)));
}

template <int LAYOUT, bool FULL>
__device__ __forceinline__ void encode_page(EncPageLds& L, const void* chunk_in, uint8_t* side_chunk,
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
    L.enc[tid] = cb->enc[tid];
    if (tid < 64) reinterpret_cast<uint32_t*>(L.r2s)[tid] = reinterpret_cast<const uint32_t*>(cb->rank2sym)[tid];
    else if (tid < 128) reinterpret_cast<uint32_t*>(L.lbr)[tid - 64] = reinterpret_cast<const uint32_t*>(cb->len_by_rank)[tid - 64];
    uint32_t mode = 0, entropy = 0;
    if (tid == 127) {
        mode = cb->mode;
        entropy = cb->entropy_q16;
    }
    {
        // the image is OR-ed into: zero what the page covers (the whole header part otherwise)
        const uint32_t upto = in_lds ? a0 + words : a0 + kHdr + 64u + kGap;
        for (uint32_t k = tid; k * 4u < upto + 4u; k += 256u)
            reinterpret_cast<uint4*>(L.img)[k] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    // first pass: the bits of this lane's string; the record's explicit ranks
    {
        uint32_t t = 0;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t len = lds_u16(enc_addr(sy[2 * rr], sy[2 * rr + 1], j, tab) + 2u);
                if (!FULL && ((q * 8u + (uint32_t)rr) * 64u + (uint32_t)lane) * 8u + (uint32_t)j >= nvalid)
                    len = 0;
                t += len;
            }
        L.tlen[tid] = t;
        const bool desc = tid > 0 && L.r2s[tid - 1] >= L.r2s[tid];
        const u64 m = __ballot(desc);
        if (lane == 0) L.red[q] = m ? q * 64u + 63u - (uint32_t)__builtin_clzll(m) : 0u;
    }
    __syncthreads();

    uint32_t head = L.red[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) head = L.red[w] > head ? L.red[w] : head;
    head = (uint32_t)__builtin_amdgcn_readfirstlane((int)head);
    const uint32_t hw = (head + 3u) / 4u;
    uint32_t* const gap = L.img + a0 + kHdr + hw;
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
            L.img[a0] = head | ((mode & 1u) << 9) | (incl << 10);
            L.img[a0 + 1] = entropy;
        }
    }
    {
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
    const bool fits = page_words(head, B) == words;
    if (!fits) {
        if (tid == 0) atomicOr(err, HF_KERR_CORRUPT);
        return;  // (the same in every thread)
    }
    const uint32_t wn = (B + 31u) / 32u;
    const uint32_t hdr_words = kHdr + hw + kGap;
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
            if (q < 2) emit_string<FULL, false, true>(sy, q, lane, nvalid, tab, pos, wn, img_str, glob_str);
            else emit_string<FULL, true, true>(sy, q, lane, nvalid, tab, pos, wn, img_str, glob_str);
        } else {
            if (q < 2) emit_string<FULL, false, false>(sy, q, lane, nvalid, tab, pos, wn, img_str, glob_str);
            else emit_string<FULL, true, false>(sy, q, lane, nvalid, tab, pos, wn, img_str, glob_str);
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

template <int LAYOUT>
__global__ __launch_bounds__(256) void hf_k_encode_pages(
    const void* __restrict__ data, const uint64_t n, const uint32_t nchunks,
    const hf_chunk_cb* __restrict__ cbs, uint8_t* __restrict__ side, uint32_t* __restrict__ payload,
    const uint32_t* __restrict__ page_off, const u64* __restrict__ total_words, const u64 limit,
    uint32_t* err) {
    __shared__ __attribute__((aligned(1024))) EncPageLds L;
    // (the pages do not fit the payload buffer, which the scan has reported: nothing is written)
    if (*total_words > limit) return;
    const uint32_t c = blockIdx.x;
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
    if (nvalid == HF_CHUNK_SYMS)
        encode_page<LAYOUT, true>(L, chunk_in, side_chunk, cbs + c, nvalid, payload, off0, off1 - off0, err);
    else
        encode_page<LAYOUT, false>(L, chunk_in, side_chunk, cbs + c, nvalid, payload, off0, off1 - off0, err);
}

// ---- decoder ---------------------------------------------------------------------------------

template <int BITS>
struct DecPageLds {
    uint16_t lut[1 << BITS];  // first: ChunkTablesT::lookup ORs the table's address into the index
    uint32_t img[kImgWords + kImgSlack];
    uint16_t tail[512];
    uint8_t r2s[256];
    uint32_t present[8];
    uint32_t rstart[132];
    uint32_t used[256];
    uint32_t red[8];
};

// 16 bytes of the payload at word index idx (a multiple of 4); words at or past the buffer's end
// read as zero (a buffer whose size is not a multiple of 16 bytes ends in a partial piece)
__device__ __forceinline__ uint4 load_piece(const uint32_t* __restrict__ payload, const u64 payload_words,
                                            const u64 idx) {
    if (idx + 4 <= payload_words) {
        const hf_u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const hf_u32x4*>(payload + idx));
        return make_uint4(t.x, t.y, t.z, t.w);
    }
    uint4 v = make_uint4(0, 0, 0, 0);
    if (idx < payload_words) v.x = payload[idx];
    if (idx + 1 < payload_words) v.y = payload[idx + 1];
    if (idx + 2 < payload_words) v.z = payload[idx + 2];
    return v;
}

// refill of a lane that walks its own words: take the next word when fewer than 32 bits are buffered
template <bool BACK>
__device__ __forceinline__ void refill_own(u64& win, int& room, uint32_t& wp) {
    const uint32_t w0 = lds_u32(wp);
    const uint32_t w = BACK ? __builtin_bitreverse32(w0) : w0;
    const bool need = room > 0;
    win |= (u64)(need ? w : 0u) << (room & 63);
    room -= need ? 32 : 0;
    wp += need ? (BACK ? (uint32_t)-4 : 4u) : 0u;
}

// the 64 symbols of string (q, lane) of a full page whose image is in LDS; returns the bits consumed
template <int LAYOUT, bool BACK, class TAB>
__device__ __forceinline__ uint32_t decode_string(const uint32_t img_str_addr, const uint32_t bitpos,
                                                  const uint32_t q, const int lane, const TAB& T,
                                                  const uint8_t* __restrict__ side_chunk,
                                                  void* __restrict__ out_chunk, uint32_t* err) {
    // forward: bitpos = the string's first bit; backward: its first bit too, the region's last
    const uint32_t wi = img_str_addr + (bitpos >> 5) * 4u;
    const uint32_t sh = BACK ? 31u - (bitpos & 31u) : (bitpos & 31u);
    u64 win;
    uint32_t wp;
    if (BACK) {
        win = ((u64)__builtin_bitreverse32(lds_u32(wi)) << 32) | (u64)__builtin_bitreverse32(lds_u32(wi - 4u));
        wp = wi - 8u;
    } else {
        win = ((u64)lds_u32(wi) << 32) | (u64)lds_u32(wi + 4u);
        wp = wi + 8u;
    }
    win <<= sh;
    int room = (int)sh - 32;
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
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const uint32_t r = q * 8u + (uint32_t)rr;
        uint32_t e[8];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            refill_own<BACK>(win, room, wp);
            if constexpr (TAB::kPairCheck)
                decode_pair_single(win, room, T, err, e[2 * jp], e[2 * jp + 1]);
            else {
                e[2 * jp] = decode_one(win, room, T, err);
                e[2 * jp + 1] = decode_one(win, room, T, err);
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
    // words taken x 32, less what is still buffered, less the bits skipped in the first word
    const uint32_t taken = BACK ? (wi + 4u - wp) / 4u - 1u : (wp - wi) / 4u;
    return taken * 32u - (uint32_t)(32 - room) - sh;
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
            const uint32_t s = decode_one(win, room, T, err) >> 8;
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

template <int LAYOUT, int BITS>
__global__ __launch_bounds__(256) void hf_k_decode_pages(
    const uint32_t* __restrict__ payload, const uint64_t payload_words,
    const uint32_t* __restrict__ page_off, const uint8_t* __restrict__ side, const uint64_t n,
    const uint32_t nchunks, const uint32_t* __restrict__ page_ids,
    const uint32_t* __restrict__ out_slots, const uint32_t n_slots, void* __restrict__ out,
    uint32_t* err) {
    typedef ChunkTablesT<BITS, 0, (BITS >= 12)> Tab;
    constexpr int kLong = Tab::kLong;
    __shared__ __attribute__((aligned(2 << BITS))) DecPageLds<BITS> L;
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
    if (off1 < off0 || (uint64_t)off1 > payload_words || off1 - off0 < kHdr + kGap ||
        off1 - off0 > HF_PAGE_MAX_WORDS || (page_ids && nvalid != HF_CHUNK_SYMS))
        return bad();
    const uint32_t words = off1 - off0;
    const uint32_t a0 = off0 & 3u;
    const u64 base = (u64)(off0 - a0);
    const bool in_lds = a0 + words <= (uint32_t)kImgWords;
    // the page (its header part when it is too large for the image) on its way to LDS, 16 bytes
    // per thread and step, all steps requested before the first is used
    constexpr int kSteps = (kImgWords + 1023) / 1024;
    const uint32_t nload = in_lds ? a0 + words : a0 + kHdr + 64u + kGap;
    uint4 v[kSteps];
#pragma unroll
    for (int j = 0; j < kSteps; ++j) {
        const uint32_t g = tid + 256u * (uint32_t)j;
        v[j] = g * 4u < nload ? load_piece(payload, payload_words, base + g * 4u) : make_uint4(0, 0, 0, 0);
    }
    // cleared tables: first-level entries no short code covers are escapes, tail entries no code
    // covers match nothing
#pragma unroll
    for (int j = 0; j < (2 << BITS) / 16 / 256; ++j)
        reinterpret_cast<uint4*>(L.lut)[tid + 256 * j] = make_uint4(0, 0, 0, 0);
    if (tid < 64) reinterpret_cast<uint4*>(L.tail)[tid] = make_uint4(0, 0, 0, 0);
    if (tid < 8) L.present[tid] = 0u;
    reinterpret_cast<uint4*>(L.img)[tid] = v[0];
    __syncthreads();

    // ---- the record ----
    const uint32_t hdr = L.img[a0];
    const uint32_t head = hdr & 0x1FFu, B = hdr >> 10;
    if ((hdr >> 29) != 0u || head > 255u || page_words(head, B) != words) return bad();
    const uint32_t hw = (head + 3u) / 4u;
    // lengths: rank tid has the smallest length L with cum(L) > tid; ranks of one length tile the
    // left-aligned 16-bit code space from cs16(L) on, 2^(16-L) values each
    uint32_t my_len = 0, my_start = 0, total = 0, kraft = 0;
    uint32_t cum_short = 0;            // ranks with codes of at most BITS bits
    uint32_t lstart[kLong], lend[kLong], loff[kLong];
#pragma unroll
    for (uint32_t d = 1; d <= 16; ++d) {
        const uint32_t cnt = (L.img[a0 + 2 + (d - 1) / 2] >> (16u * ((d - 1) & 1u))) & 0xFFFFu;
        if (my_len == 0 && tid < total + cnt) {
            my_len = d;
            my_start = kraft + ((tid - total) << (16 - d));
        }
        if (d > (uint32_t)BITS) {
            lstart[d - BITS - 1] = kraft;
            loff[d - BITS - 1] = total;
            lend[d - BITS - 1] = kraft + (cnt << (16 - d));
        }
        total += cnt;
        kraft += cnt << (16 - d);
        if (d == (uint32_t)BITS) cum_short = total;
    }
    (void)cum_short;
    if (total != 256u || kraft > 65536u) return bad();
    if (tid < head) {
        const uint32_t s = reinterpret_cast<const uint8_t*>(L.img + a0 + kHdr)[tid];
        L.r2s[tid] = (uint8_t)s;
        __hip_atomic_fetch_or(&L.present[s >> 5], 1u << (s & 31u), __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // region lengths and their prefix sum (wavefronts 0 and 1: one region per thread)
    uint32_t R = 0;
    if (q < 2) {
        const uint32_t* gap = L.img + a0 + kHdr + hw;
        const uint32_t bit = 12u * tid;
        const u64 two = ((u64)gap[(bit >> 5) + 1] << 32) | (u64)gap[bit >> 5];
        R = (uint32_t)(two >> (bit & 31u)) & 0xFFFu;
        uint32_t incl = R;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (q == 0 && lane == 63) L.red[0] = incl;
        L.used[tid] = incl;  // (parked: the inclusive sum within the wavefront)
    }
    __syncthreads();

    uint32_t npresent = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) npresent += (uint32_t)__popc(L.present[w]);
    if (npresent != head) return bad();  // a symbol listed twice
    {
        // the ranks behind the listed ones: the symbols not listed, ascending
        const uint32_t s = tid, w = s >> 5;
        uint32_t below = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) below += (uint32_t)i < w ? (uint32_t)__popc(L.present[i]) : 0u;
        const uint32_t mine = L.present[w];
        below += (uint32_t)__popc(mine & ((1u << (s & 31u)) - 1u));
        if (!((mine >> (s & 31u)) & 1u)) L.r2s[head + s - below] = (uint8_t)s;
    }
    if (q < 2) {
        const uint32_t incl = L.used[tid] + (q == 1 ? L.red[0] : 0u);
        L.rstart[tid] = incl - R;
        if (tid == 127) L.rstart[128] = incl;
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
    {
        const uint32_t val = ((uint32_t)L.r2s[tid] << 8) | my_len;
        const uint32_t v2 = val | (val << 16);
        auto fill = [&](uint16_t* base, const uint32_t at, const uint32_t span) {
            if (span >= 8) {
                for (uint32_t i = 0; i < span; i += 8)
                    *reinterpret_cast<uint4*>(base + at + i) = make_uint4(v2, v2, v2, v2);
            } else if (span == 4) {
                *reinterpret_cast<uint2*>(base + at) = make_uint2(v2, v2);
            } else if (span == 2) {
                *reinterpret_cast<uint32_t*>(base + at) = v2;
            } else {
                base[at] = (uint16_t)val;
            }
        };
        // the long codes' part of the code space; a direct table when it is small (rare symbols:
        // the usual case), the blocks' bounds otherwise
        const uint32_t tail_span = lend[kLong - 1] - lstart[0];
        const uint32_t tail_n = tail_span <= 512u ? tail_span : 0u;
        if (my_len <= (uint32_t)BITS)
            fill(L.lut, my_start >> (16 - BITS), 1u << (BITS - my_len));
        else if (tail_n != 0)
            fill(L.tail, my_start - lstart[0], 1u << (16 - my_len));
        T.lut_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(L.lut));
        T.rank2sym = L.r2s;
        T.tail = L.tail;
        T.tail_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)tail_n);
#pragma unroll
        for (int i = 0; i < kLong; ++i) {
            T.start[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lstart[i]);
            T.end[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lend[i]);
            T.off[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)loff[i]);
        }
    }
    __syncthreads();
    if (L.rstart[128] != B) return bad();  // the regions do not add up to the page's bits

    // ---- the four sub-decoders ----
    const size_t elem_bytes = LAYOUT == HF_LAYOUT_BF16_EXP ? 2 : 1;
    const uint8_t* side_chunk = side ? side + first : nullptr;
    const uint64_t out_first = page_ids ? (uint64_t)slot * HF_CHUNK_SYMS : first;
    void* out_chunk = reinterpret_cast<uint8_t*>(out) + out_first * elem_bytes;
    const uint32_t hdr_words = kHdr + hw + kGap;
    const uint32_t r = (q & 1u) * 64u + (uint32_t)lane;
    const uint32_t s0 = L.rstart[r], s1 = L.rstart[r + 1];
    uint32_t used;
    // (a full page's strings have 64 bits at least; a region shorter than two of them is damage,
    // and the whole wavefront then goes the careful way)
    const bool fast = in_lds && nvalid == HF_CHUNK_SYMS && __ballot(s1 - s0 < 128u) == 0ull;
    if (fast) {
        const uint32_t str_addr = lds_address(L.img + a0 + hdr_words);
        if (q < 2)
            used = decode_string<LAYOUT, false>(str_addr, s0, q, lane, T, side_chunk, out_chunk, err);
        else
            used = decode_string<LAYOUT, true>(str_addr, s1 - 1u, q, lane, T, side_chunk, out_chunk, err);
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
    L.used[tid] = used;
    __syncthreads();
    // the two strings of a region meet exactly
    if (q < 2 && L.used[tid] + L.used[tid + 128] != R) atomicOr(err, HF_KERR_CORRUPT);
}

}  // namespace

// sizes -> page offsets (d_page_off, nchunks + 1 values, the last one the total) -> pages
void hf_launch_encode_pages(const void* d_data, uint64_t n, int layout, const hf_chunk_cb* cbs,
                            const uint32_t* d_freq_chunks, uint8_t* d_side, uint32_t* d_payload,
                            uint64_t cap_words, uint32_t* d_page_off,
                            unsigned long long* d_scan_aux, uint32_t* d_err, hipStream_t s) {
    const uint64_t nchunks = (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS;
    if (nchunks == 0) return;
    const uint64_t m = nchunks + 1;
    hipLaunchKernelGGL(hf_k_page_sizes, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, s, d_freq_chunks,
                       cbs, (uint32_t)nchunks, d_page_off);
    const u64 limit = cap_words < 0xFFFFFFFFull ? cap_words : 0xFFFFFFFFull;
    hf_launch_scan_u32(d_page_off, m, d_scan_aux, limit, d_err, s);
    const u64* d_total = d_scan_aux + (m + 1023) / 1024;
    if (layout == HF_LAYOUT_BF16_EXP)
        hipLaunchKernelGGL(hf_k_encode_pages<HF_LAYOUT_BF16_EXP>, dim3((unsigned)nchunks), dim3(256), 0,
                           s, d_data, n, (uint32_t)nchunks, cbs, d_side, d_payload, d_page_off, d_total,
                           limit, d_err);
    else
        hipLaunchKernelGGL(hf_k_encode_pages<HF_LAYOUT_U8>, dim3((unsigned)nchunks), dim3(256), 0, s,
                           d_data, n, (uint32_t)nchunks, cbs, (uint8_t*)nullptr, d_payload, d_page_off,
                           d_total, limit, d_err);
}

void hf_launch_decode_pages(const uint32_t* d_payload, uint64_t payload_words,
                            const uint32_t* d_page_off, const uint8_t* d_side, uint64_t n,
                            int layout, const uint32_t* d_page_ids, const uint32_t* d_out_slots,
                            uint64_t n_slots, uint64_t n_ids, void* d_out, uint32_t* d_err,
                            hipStream_t s) {
    const uint64_t nchunks = (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS;
    const uint64_t n_work = d_page_ids ? n_ids : nchunks;
    if (n_work == 0) return;
    if (layout == HF_LAYOUT_BF16_EXP)
        hipLaunchKernelGGL((hf_k_decode_pages<HF_LAYOUT_BF16_EXP, 11>), dim3((unsigned)n_work),
                           dim3(256), 0, s, d_payload, payload_words, d_page_off, d_side, n,
                           (uint32_t)nchunks, d_page_ids, d_out_slots, (uint32_t)n_slots, d_out, d_err);
    else
        hipLaunchKernelGGL((hf_k_decode_pages<HF_LAYOUT_U8, 11>), dim3((unsigned)n_work), dim3(256), 0,
                           s, d_payload, payload_words, d_page_off, (const uint8_t*)nullptr, n,
                           (uint32_t)nchunks, d_page_ids, d_out_slots, (uint32_t)n_slots, d_out, d_err);
}
