// Packed-stream encoder: the lookup mux of EntropyAwareHuffmanSystem
// (reference: top_encoder.scala:124-162 emits one {code, length} beat per symbol and stops
// there) followed by the bit packer the reference does not have.
//
// Stream layout.  A chunk is 16384 symbols and belongs to one wavefront.  In round r lane l
// owns the 8 consecutive symbols of group g = r*64 + l, so every global access is a
// contiguous 16 B (bf16) or 8 B (bytes) per lane.  Each lane concatenates its codes MSB first
// into a private bit string cut into 32-bit words; the chunk payload is those words of all 64
// lanes in exactly the order a lock-step decoder asks for them: before every pair of symbols
// each lane holding fewer than 32 buffered bits takes the next word, lanes in ascending
// order.  The encoder replays that schedule from the code lengths alone, so each word's slot
// is known before its bits are complete (never more than two slots are pending per lane).
//
// A chunk's words are staged in LDS, its size is published in a descriptor, and a decoupled
// look-back over the earlier descriptors yields its place in the payload; the staged words
// then leave with coalesced stores.  Chunks are handed out through an atomic ticket so a
// chunk only ever waits for chunks whose wavefronts are already running.
//
// HBM traffic per bf16 element: 2 B read, 1 B side stream written, ~L/8 B payload written.
#include "hf_common.h"

namespace {

typedef unsigned long long u64;

constexpr int kWaves = 4;            // wavefronts per workgroup
constexpr int kStgWords = 4224;      // LDS staging per wavefront (16.5 KiB: 8 bits/symbol + slack)
constexpr int kMaxChunkWords = 129 * 64;                  // 128 code words + 1 look-ahead, per lane
constexpr int kSpillWords = kMaxChunkWords - kStgWords;   // global spill per wavefront slot

constexpr u64 kDescAggregate = 1ull << 62;
constexpr u64 kDescPrefix = 2ull << 62;
constexpr u64 kDescValueMask = (1ull << 62) - 1ull;
constexpr u64 kSpinTicks = 100000000ull;  // 1 s

// Exclusive prefix (in words) of chunk c over the chunk descriptors.  Lane j inspects the
// descriptor of chunk idx - j; the nearest descriptor holding an inclusive prefix ends the
// walk.  A descriptor is one 8-byte {status, value} word written by a single agent-scope
// store, so a reader never needs more than the word itself.
__device__ u64 lookback(const u64* desc, const uint32_t c, const int lane, uint32_t* err) {
    u64 excl = 0;
    long long idx = (long long)c - 1;
    uint32_t spins = 0;
    u64 t0 = 0;
    while (idx >= 0) {
        const long long j = idx - lane;
        u64 d;
        for (;;) {
            d = j >= 0 ? __hip_atomic_load(&desc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       : kDescPrefix;  // before chunk 0: prefix 0
            const u64 m_inv = __ballot((d >> 62) == 0);
            const u64 m_pre = __ballot((d >> 62) == 2);
            const int first_pre = m_pre ? (int)__builtin_ctzll(m_pre) : 64;
            const u64 upto = first_pre >= 63 ? ~0ull : ((2ull << first_pre) - 1ull);
            if ((m_inv & upto) == 0) {
                const u64 v = (lane <= first_pre) ? (d & kDescValueMask) : 0ull;
                excl += hf_wave_sum_u64(v);
                if (first_pre < 64) return excl;
                break;
            }
            // bounded wait: give up after kSpinTicks of the 100 MHz wall clock
            if ((++spins & 63u) == 0) {
                const u64 now = __builtin_amdgcn_s_memrealtime();
                if (t0 == 0) t0 = now;
                if (now - t0 > kSpinTicks) {
                    if (lane == 0) atomicOr(err, HF_KERR_TIMEOUT);
                    return excl;
                }
            }
            __builtin_amdgcn_s_sleep(2);
        }
        idx -= 64;
    }
    return excl;
}

template <int LAYOUT, bool FULL>
__device__ __forceinline__ void load_group(const void* chunk_in, const uint32_t g,
                                           const uint32_t nvalid, uint32_t (&raw)[4]) {
    if (LAYOUT == HF_LAYOUT_BF16_EXP) {
        if (FULL || (g * 8 + 8 <= nvalid)) {
            const uint4 v = reinterpret_cast<const uint4*>(chunk_in)[g];
            raw[0] = v.x;
            raw[1] = v.y;
            raw[2] = v.z;
            raw[3] = v.w;
        } else {
            const uint16_t* p = reinterpret_cast<const uint16_t*>(chunk_in) + (size_t)g * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t lo = (g * 8 + 2 * i < nvalid) ? p[2 * i] : 0u;
                const uint32_t hi = (g * 8 + 2 * i + 1 < nvalid) ? p[2 * i + 1] : 0u;
                raw[i] = lo | (hi << 16);
            }
        }
    } else {
        if (FULL || (g * 8 + 8 <= nvalid)) {
            const uint2 v = reinterpret_cast<const uint2*>(chunk_in)[g];
            raw[0] = v.x;
            raw[1] = v.y;
        } else {
            const uint8_t* p = reinterpret_cast<const uint8_t*>(chunk_in) + (size_t)g * 8;
            raw[0] = raw[1] = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t b = (g * 8 + i < nvalid) ? p[i] : 0u;
                raw[i >> 2] |= b << (8 * (i & 3));
            }
        }
        raw[2] = raw[3] = 0;
    }
}

// One chunk.  Returns the number of payload words (wave-uniform).
template <int LAYOUT, bool FULL>
__device__ uint32_t encode_chunk(const void* chunk_in, const uint32_t nvalid,
                                 const uint32_t* __restrict__ enc, uint32_t* __restrict__ stg,
                                 uint32_t* __restrict__ spill, uint8_t* side_chunk,
                                 const int lane) {
    u64 acc = 0;          // code bits not yet emitted, LSB aligned
    uint32_t ab = 0;      // valid bits in acc (< 32 between pairs)
    int nb = 0;           // bits the decoder holds for this lane
    uint32_t p0 = 0, p1 = 0, fc = 0;  // slots of the words asked for but not yet complete
    uint32_t pos = 0;     // next free slot of the chunk (wave-uniform)

    const uint32_t rounds = FULL ? HF_ROUNDS : (nvalid + 511u) / 512u;
    uint32_t cur[4], nxt[4];
    load_group<LAYOUT, FULL>(chunk_in, (uint32_t)lane, nvalid, cur);
    if (rounds > 1)
        load_group<LAYOUT, FULL>(chunk_in, 64u + lane, nvalid, nxt);
    else
        nxt[0] = nxt[1] = nxt[2] = nxt[3] = 0;

    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t g = r * 64u + lane;
        uint32_t nn[4] = {0, 0, 0, 0};
        if (r + 2 < rounds) load_group<LAYOUT, FULL>(chunk_in, g + 128u, nvalid, nn);

        uint32_t s[8];
        if (LAYOUT == HF_LAYOUT_BF16_EXP) {
            uint32_t sd[2] = {0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t d = cur[i];
                s[2 * i] = (d >> 7) & 0xFFu;
                s[2 * i + 1] = (d >> 23) & 0xFFu;
                const uint32_t b0 = ((d >> 8) & 0x80u) | (d & 0x7Fu);
                const uint32_t b1 = ((d >> 24) & 0x80u) | ((d >> 16) & 0x7Fu);
                sd[i >> 1] |= (b0 | (b1 << 8)) << (16 * (i & 1));
            }
            if (FULL || (g * 8 + 8 <= nvalid)) {
                reinterpret_cast<uint2*>(side_chunk)[g] = make_uint2(sd[0], sd[1]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (g * 8 + i < nvalid)
                        side_chunk[(size_t)g * 8 + i] = (uint8_t)(sd[i >> 2] >> (8 * (i & 3)));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = (cur[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        }

        uint32_t e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = enc[s[i]];

#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            uint32_t ea = e[2 * jp], eb = e[2 * jp + 1];
            bool active = true;
            if (!FULL) {
                const uint32_t i0 = g * 8 + 2 * jp;
                active = i0 < nvalid;
                if (!active) ea = 0;
                if (i0 + 1 >= nvalid) eb = 0;
            }
            const uint32_t la = ea >> 16, lb = eb >> 16;
            const uint32_t cl = la + lb;
            const uint32_t comb = ((ea & 0xFFFFu) << lb) | (eb & 0xFFFFu);

            // the decoder's refill schedule
            const bool need = active && nb < 32;
            const u64 m = __ballot(need);
            const uint32_t p = pos + hf_mbcnt(m);
            p0 = (need && fc == 0) ? p : p0;
            p1 = (need && fc != 0) ? p : p1;
            fc += need ? 1u : 0u;
            nb += need ? 32 : 0;
            pos += (uint32_t)__popcll(m);
            nb -= (int)cl;

            // pack
            acc = (acc << cl) | (u64)comb;
            ab += cl;
            const bool done = ab >= 32;
            if (done) {
                const uint32_t w = (uint32_t)(acc >> (ab - 32));
                if (p0 < (uint32_t)kStgWords)
                    stg[p0] = w;
                else
                    spill[p0 - kStgWords] = w;
            }
            p0 = done ? p1 : p0;
            fc -= done ? 1u : 0u;
            ab -= done ? 32u : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cur[i] = nxt[i];
            nxt[i] = nn[i];
        }
    }
    // words asked for but not filled: the zero-padded last word, then an all-zero word
    if (fc > 0) {
        const uint32_t w = (uint32_t)(acc << (32 - ab));
        if (p0 < (uint32_t)kStgWords)
            stg[p0] = w;
        else
            spill[p0 - kStgWords] = w;
        p0 = p1;
        fc -= 1;
    }
    if (fc > 0) {
        if (p0 < (uint32_t)kStgWords)
            stg[p0] = 0;
        else
            spill[p0 - kStgWords] = 0;
    }
    return pos;
}

template <int LAYOUT>
__global__ __launch_bounds__(kWaves * 64) void hf_k_encode(
    const void* __restrict__ data, const uint64_t n, const uint32_t nchunks,
    const hf_cb_dev* __restrict__ cb, uint8_t* __restrict__ side, uint32_t* __restrict__ payload,
    const uint64_t cap_words, uint32_t* __restrict__ chunk_off, u64* desc, uint32_t* ticket,
    uint32_t* spill_base, uint32_t* err) {
    __shared__ uint32_t enc[256];
    __shared__ uint32_t stg_all[kWaves][kStgWords];
    enc[threadIdx.x] = cb->enc[threadIdx.x];
    __syncthreads();

    const int lane = hf_lane();
    const int wave = threadIdx.x >> 6;
    uint32_t* stg = stg_all[wave];
    uint32_t* spill = spill_base + ((size_t)blockIdx.x * kWaves + wave) * kSpillWords;
    const size_t elem_bytes = LAYOUT == HF_LAYOUT_BF16_EXP ? 2 : 1;

    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(ticket, 1u);
        const uint32_t c = __builtin_amdgcn_readfirstlane(t);
        if (c >= nchunks) break;

        const uint64_t first = (uint64_t)c * HF_CHUNK_SYMS;
        const uint64_t left = n - first;
        const uint32_t nvalid = left < HF_CHUNK_SYMS ? (uint32_t)left : HF_CHUNK_SYMS;
        const void* chunk_in = reinterpret_cast<const uint8_t*>(data) + first * elem_bytes;
        uint8_t* side_chunk = side ? side + first : nullptr;

        uint32_t total;
        if (nvalid == HF_CHUNK_SYMS)
            total = encode_chunk<LAYOUT, true>(chunk_in, nvalid, enc, stg, spill, side_chunk, lane);
        else
            total = encode_chunk<LAYOUT, false>(chunk_in, nvalid, enc, stg, spill, side_chunk, lane);
        total = __builtin_amdgcn_readfirstlane(total);

        if (lane == 0)
            __hip_atomic_store(&desc[c], kDescAggregate | (u64)total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        // the prefix is the same in every lane: keep it in scalar registers so the branches
        // below are plain scalar branches
        const u64 base_v = lookback(desc, c, lane, err);
        const u64 base = ((u64)__builtin_amdgcn_readfirstlane((uint32_t)(base_v >> 32)) << 32) |
                         (u64)__builtin_amdgcn_readfirstlane((uint32_t)base_v);
        if (lane == 0) {
            __hip_atomic_store(&desc[c], kDescPrefix | (base + total), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            chunk_off[c] = (uint32_t)base;
            if (c == nchunks - 1) chunk_off[nchunks] = (uint32_t)(base + total);
        }
        const bool fits = base + total <= cap_words && base + total <= 0xFFFFFFFFull;
        if (!fits && lane == 0) atomicOr(err, HF_KERR_OVERFLOW);
        // LDS writes of all lanes are complete before any lane reads them back
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t ncopy = fits ? total : 0u;  // nothing is written past the buffer
        const uint32_t in_lds = ncopy < (uint32_t)kStgWords ? ncopy : (uint32_t)kStgWords;
        uint32_t* dst = payload + base;
        for (uint32_t i = lane; i < in_lds; i += 64) dst[i] = stg[i];
        if (ncopy > (uint32_t)kStgWords) {
            // the spill words were written by other lanes of this wavefront: drain the
            // stores, then read them past the vector L1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (uint32_t i = kStgWords + lane; i < ncopy; i += 64)
                dst[i] = __hip_atomic_load(&spill[i - kStgWords], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // the staging area is reused by this wavefront's next chunk
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

size_t hf_encode_spill_bytes(int num_cu) {
    return (size_t)num_cu * 2 * kWaves * kSpillWords * sizeof(uint32_t);
}

void hf_launch_encode(const void* d_data, uint64_t n, int layout, const hf_cb_dev* cb,
                      uint8_t* d_side, uint32_t* d_payload, uint64_t cap_words,
                      uint32_t* d_chunk_off, unsigned long long* d_desc, uint32_t* d_ticket,
                      uint32_t* d_spill, uint32_t* d_err, int num_cu, hipStream_t s) {
    const uint64_t nchunks = (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS;
    if (nchunks == 0) return;
    uint64_t grid = (nchunks + kWaves - 1) / kWaves;
    const uint64_t cap = (uint64_t)num_cu * 2;  // two 67 KiB workgroups fit one CU's LDS
    if (grid > cap) grid = cap;
    if (layout == HF_LAYOUT_BF16_EXP)
        hipLaunchKernelGGL(hf_k_encode<HF_LAYOUT_BF16_EXP>, dim3((unsigned)grid),
                           dim3(kWaves * 64), 0, s, d_data, n, (uint32_t)nchunks, cb, d_side,
                           d_payload, cap_words, d_chunk_off, d_desc, d_ticket, d_spill, d_err);
    else
        hipLaunchKernelGGL(hf_k_encode<HF_LAYOUT_U8>, dim3((unsigned)grid), dim3(kWaves * 64), 0, s,
                           d_data, n, (uint32_t)nchunks, cb, (uint8_t*)nullptr, d_payload,
                           cap_words, d_chunk_off, d_desc, d_ticket, d_spill, d_err);
}
