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
// A chunk's words are staged in LDS.  The kWaves chunks a workgroup encodes side by side form a
// tile; the tile's size is published in a descriptor, a decoupled look-back over the earlier
// descriptors yields its place in the payload, and the staged words then leave with coalesced
// stores.  Tiles are handed out through an atomic ticket so a tile only ever waits for tiles
// whose workgroups are already running.
//
// HBM traffic per bf16 element: 2 B read, 1 B side stream written, ~L/8 B payload written.
#include <type_traits>

#include "hf_common.h"

namespace {

typedef unsigned long long u64;

constexpr int kWaves = 4;            // wavefronts per workgroup
constexpr int kStgWords = 4224;      // LDS staging per wavefront (16.5 KiB: 8 bits/symbol + slack)
constexpr int kMaxChunkWords = 129 * 64;                  // 128 code words + 1 look-ahead, per lane
constexpr int kSpillWords = kMaxChunkWords - kStgWords;   // global spill per wavefront slot
constexpr int kDepth = 8;            // input groups (rounds) a wavefront keeps in flight

constexpr u64 kDescAggregate = 1ull << 62;
constexpr u64 kDescPrefix = 2ull << 62;
constexpr u64 kDescValueMask = (1ull << 62) - 1ull;
constexpr u64 kSpinTicks = 100000000ull;  // 1 s

// Exclusive prefix (in words) of tile t over the tile descriptors.  Each pass inspects the 128
// tiles before the current position, two per lane (lane j: idx - j and idx - 64 - j); the
// nearest descriptor holding an inclusive prefix ends the walk.  A descriptor is one 8-byte
// {status, value} word written by a single agent-scope store, so a reader never needs more
// than the word itself.
__device__ u64 lookback(const u64* desc, const uint32_t t, const int lane, uint32_t* err) {
    u64 excl = 0;
    long long idx = (long long)t - 1;
    uint32_t spins = 0;
    u64 t0 = 0;
    while (idx >= 0) {
        const long long j0 = idx - lane, j1 = idx - 64 - lane;
        for (;;) {
            // before tile 0: prefix 0
            const u64 d0 = j0 >= 0 ? __hip_atomic_load(&desc[j0], __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT)
                                   : kDescPrefix;
            const u64 d1 = j1 >= 0 ? __hip_atomic_load(&desc[j1], __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT)
                                   : kDescPrefix;
            const u64 inv0 = __ballot((d0 >> 62) == 0), pre0 = __ballot((d0 >> 62) == 2);
            const u64 inv1 = __ballot((d1 >> 62) == 0), pre1 = __ballot((d1 >> 62) == 2);
            // position of the nearest prefix among the 128 (128 = none)
            const int first = pre0 ? (int)__builtin_ctzll(pre0)
                                   : (pre1 ? 64 + (int)__builtin_ctzll(pre1) : 128);
            const int f0 = first < 63 ? first : 63;           // last near lane that counts
            const int f1 = first < 64 ? -1 : (first - 64 < 63 ? first - 64 : 63);
            const u64 upto0 = f0 >= 63 ? ~0ull : ((2ull << f0) - 1ull);
            const u64 upto1 = f1 < 0 ? 0ull : (f1 >= 63 ? ~0ull : ((2ull << f1) - 1ull));
            if (((inv0 & upto0) | (inv1 & upto1)) == 0) {
                u64 v = lane <= f0 ? (d0 & kDescValueMask) : 0ull;
                v += lane <= f1 ? (d1 & kDescValueMask) : 0ull;
                excl += hf_wave_sum_u64(v);
                if (first < 128) return excl;
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
            __builtin_amdgcn_s_sleep(1);
        }
        idx -= 128;
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

// Per-lane packing state of one chunk.
struct LaneState {
    u64 acc = 0;          // code bits not yet emitted, LSB aligned
    uint32_t ab = 0;      // valid bits in acc (< 32 between pairs)
    int nb = 0;           // bits the decoder holds for this lane; nb + ab = 32 * (pending slots)
    uint32_t p0 = 0, p1 = 0;  // slots of the words asked for but not yet complete (at most 2)
    uint32_t pos = 0;     // next free slot of the chunk (wave-uniform)
};

// Lanes with `done` place word `w` in slot `slot`.  The LDS path is a plain predicated
// ds_write; the global spill area is only touched under a wave-uniform branch, so the common
// path carries no pointer selection.
//
// CHECK = false: the caller has established (from the wave-uniform slot counter) that every
// slot in play lies inside the LDS staging area, and the test disappears.
template <bool CHECK = true>
__device__ __forceinline__ void put_word(const bool done, const uint32_t slot, const uint32_t w,
                                         uint32_t* __restrict__ stg, uint32_t* __restrict__ spill) {
    if (!CHECK) {
        if (done) stg[slot] = w;
        return;
    }
    const bool in_lds = slot < (uint32_t)kStgWords;
    if (done && in_lds) stg[slot] = w;
    if (__builtin_expect(__any(done && !in_lds), 0)) {
        if (done && !in_lds) spill[slot - kStgWords] = w;
    }
}

// One round of one lane: 8 symbols out of `raw`, side bytes stored, codes looked up, the
// decoder's refill schedule replayed and completed words placed.
template <int LAYOUT, bool FULL, bool CHECK>
__device__ __forceinline__ void encode_round(const uint32_t (&raw)[4], const uint32_t g,
                                             const uint32_t nvalid, LaneState& st,
                                             const uint32_t* __restrict__ enc,
                                             uint32_t* __restrict__ stg,
                                             uint32_t* __restrict__ spill, uint8_t* side_chunk,
                                             hf_gmem_u8* side_row) {
    uint32_t s[8];
    if (LAYOUT == HF_LAYOUT_BF16_EXP) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t d = raw[i];
            s[2 * i] = (d >> 7) & 0xFFu;
            s[2 * i + 1] = (d >> 23) & 0xFFu;
        }
        // side bytes of four elements at once: v_perm_b32 gathers their low bytes (mantissa
        // under the exponent's last bit) and their high bytes (sign over exponent bits), one
        // bit-select keeps the mantissas of the first and the signs of the second
        uint32_t sd[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t lo = __builtin_amdgcn_perm(raw[2 * h + 1], raw[2 * h], 0x06040200u);
            const uint32_t hi = __builtin_amdgcn_perm(raw[2 * h + 1], raw[2 * h], 0x07050301u);
            // (lo & mask) | (hi & ~mask), spelled out: the compiler splits it in three
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(sd[h]) : "s"(0x7F7F7F7Fu), "v"(lo), "v"(hi));
        }
        if (FULL) {
            *reinterpret_cast<hf_gmem_u32x2*>(side_row) = hf_u32x2{sd[0], sd[1]};
        } else if (g * 8 + 8 <= nvalid) {
            reinterpret_cast<uint2*>(side_chunk)[g] = make_uint2(sd[0], sd[1]);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (g * 8 + i < nvalid)
                    side_chunk[(size_t)g * 8 + i] = (uint8_t)(sd[i >> 2] >> (8 * (i & 3)));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = (raw[i >> 2] >> (8 * (i & 3))) & 0xFFu;
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

        // the decoder's refill schedule: a lane asks for a word when it holds < 32 bits
        const bool need = active && st.nb < 32;
        const u64 m = __ballot(need);
        const uint32_t p = st.pos + hf_mbcnt(m);
        // no slot pending <=> nb + ab == 0 <=> nb == 0 (ab < 32 and the sum is a multiple of 32)
        const bool empty = st.nb == 0;
        st.p1 = need ? p : st.p1;                      // the newest slot
        if (FULL) {
            // an empty lane always needs a word, and 0 <= nb < 64 makes (nb | 32) the count
            // after the refill whether or not one happened
            st.p0 = empty ? p : st.p0;
            st.nb = (st.nb | 32) - (int)cl;
        } else {
            st.p0 = (need && empty) ? p : st.p0;       // the oldest slot
            st.nb += (need ? 32 : 0) - (int)cl;
        }
        st.pos += (uint32_t)__popcll(m);

        // pack: append the pair, emit the oldest pending word once 32 bits are complete
        st.acc = (st.acc << cl) | (u64)comb;
        st.ab += cl;
        const bool done = st.ab >= 32;
        put_word<CHECK>(done, st.p0, (uint32_t)(st.acc >> (st.ab & 31u)), stg, spill);
        st.p0 = done ? st.p1 : st.p0;
        st.ab &= 31u;
    }
}

// One chunk.  Returns the number of payload words (wave-uniform).  Always inlined: as a called
// function its table and staging pointers would be generic ones, and every LDS access a flat
// access whose wait also drains the input loads in flight.
// Round r of a full chunk: the 8 symbols of group r * 64 + lane.  `gin` is the chunk's base in
// scalar registers; round r adds a scalar offset, the lane's share is a constant vector offset.
template <int LAYOUT>
__device__ __forceinline__ void load_round(const hf_gmem_u8* gin, const uint32_t r, const int lane,
                                           uint32_t (&b)[4]) {
    if (LAYOUT == HF_LAYOUT_BF16_EXP) {
        const hf_u32x4 v =
            *reinterpret_cast<const hf_gmem_u32x4*>(gin + r * 1024u + (uint32_t)lane * 16u);
        b[0] = v.x, b[1] = v.y, b[2] = v.z, b[3] = v.w;
    } else {
        const hf_u32x2 v =
            *reinterpret_cast<const hf_gmem_u32x2*>(gin + r * 512u + (uint32_t)lane * 8u);
        b[0] = v.x, b[1] = v.y, b[2] = b[3] = 0;
    }
}

// Issue the loads of the first kDepth rounds of a full chunk.
template <int LAYOUT>
__device__ __forceinline__ void preload_chunk(const void* chunk_in, const int lane,
                                              uint32_t (&buf)[kDepth][4]) {
    const hf_gmem_u8* const gin = hf_uniform_ptr(chunk_in);
#pragma unroll
    for (int d = 0; d < kDepth; ++d) load_round<LAYOUT>(gin, (uint32_t)d, lane, buf[d]);
}

// `buf` holds (or has loads in flight for) the first kDepth rounds when FULL.
template <int LAYOUT, bool FULL>
__device__ __forceinline__ uint32_t encode_chunk(const void* chunk_in, const uint32_t nvalid,
                                 const uint32_t* __restrict__ enc, uint32_t* __restrict__ stg,
                                 uint32_t* __restrict__ spill, uint8_t* side_chunk,
                                 const int lane, uint32_t (&buf)[kDepth][4]) {
    LaneState st;
    if (FULL) {
        // Input groups are fetched kDepth rounds ahead into kDepth named register sets; a set
        // is refilled right after its symbols are extracted, so no register is copied and the
        // compiler's waits stay counted.  With two wavefronts per SIMD this depth is what keeps
        // enough bytes in flight to cover HBM latency.
        const hf_gmem_u8* const gin = hf_uniform_ptr(chunk_in);
        hf_gmem_u8* const gside = hf_uniform_ptr(side_chunk);
        const uint32_t lane8 = (uint32_t)lane * 8u;
        for (uint32_t r0 = 0; r0 < HF_ROUNDS; r0 += kDepth) {
#pragma unroll
            for (int h = 0; h < kDepth; h += 4) {
                // four rounds hand out at most 4 x 256 slots and every pending slot is below
                // the counter: if they all fit the staging area no per-word test is needed
                const bool fits = st.pos + 1024u <= (uint32_t)kStgWords;  // wave-uniform
                auto four = [&](auto check) {
#pragma unroll
                    for (int d = h; d < h + 4; ++d) {
                        const uint32_t r = r0 + d;
                        uint32_t raw[4] = {buf[d][0], buf[d][1], buf[d][2], buf[d][3]};
                        // refill this set for round r + kDepth (the last rounds re-read their
                        // own group)
                        load_round<LAYOUT>(gin, r + kDepth < HF_ROUNDS ? r + kDepth : r, lane,
                                           buf[d]);
                        encode_round<LAYOUT, true, decltype(check)::value>(
                            raw, r * 64u + lane, nvalid, st, enc, stg, spill, side_chunk,
                            gside + r * 512u + lane8);
                    }
                };
                if (fits)
                    four(std::false_type{});
                else
                    four(std::true_type{});
            }
        }
    } else {
        const uint32_t rounds = (nvalid + 511u) / 512u;
        for (uint32_t r = 0; r < rounds; ++r) {
            const uint32_t g = r * 64u + lane;
            uint32_t raw[4];
            load_group<LAYOUT, false>(chunk_in, g, nvalid, raw);
            encode_round<LAYOUT, false, true>(raw, g, nvalid, st, enc, stg, spill, side_chunk,
                                              nullptr);
        }
    }
    // words asked for but not filled: the zero-padded last word, then an all-zero word
    const uint32_t pending = (uint32_t)(st.nb + (int)st.ab) >> 5;  // 0, 1 or 2 slots
    put_word(pending > 0, st.p0, (uint32_t)(st.acc << (32 - st.ab)), stg, spill);
    put_word(pending > 1, st.p1, 0u, stg, spill);
    return st.pos;
}

template <bool PER_CHUNK>
struct EncLds {
    uint32_t enc[PER_CHUNK ? kWaves * 256 : 256];
    uint32_t stg[kWaves][kStgWords];
    u64 base;
    uint32_t tot[kWaves];
    uint32_t ticket;
};

// PER_CHUNK = false: one codebook (hf_cb_dev) for the whole stream, its encoder table shared
// by the workgroup.  PER_CHUNK = true: chunk c is coded with record c of an hf_chunk_cb array
// (per-block codebooks); each wavefront keeps its chunk's table in its own 1 KiB of LDS.
template <int LAYOUT, bool PER_CHUNK>
__global__ __launch_bounds__(kWaves * 64) void hf_k_encode(
    const void* __restrict__ data, const uint64_t n, const uint32_t nchunks,
    const void* __restrict__ cbv, uint8_t* __restrict__ side, uint32_t* __restrict__ payload,
    const uint64_t cap_words, uint32_t* __restrict__ chunk_off, u64* desc, uint32_t* ticket,
    uint32_t* spill_base, uint32_t* err) {
    // one object, so that the encoder table sits at LDS address 0 and a lookup's address is the
    // symbol times four with nothing added
    __shared__ EncLds<PER_CHUNK> L;
    uint32_t(&stg_all)[kWaves][kStgWords] = L.stg;
    uint32_t(&tile_tot)[kWaves] = L.tot;
    uint32_t& tile_ticket = L.ticket;
    u64& tile_base = L.base;
    if (!PER_CHUNK)
        L.enc[threadIdx.x] = reinterpret_cast<const hf_cb_dev*>(cbv)->enc[threadIdx.x];
    uint32_t* enc = PER_CHUNK ? L.enc + (threadIdx.x >> 6) * 256 : L.enc;

    const int lane = hf_lane();
    const int wave = threadIdx.x >> 6;
    uint32_t* stg = stg_all[wave];
    uint32_t* spill = spill_base + ((size_t)blockIdx.x * kWaves + wave) * kSpillWords;
    const size_t elem_bytes = LAYOUT == HF_LAYOUT_BF16_EXP ? 2 : 1;

    // A tile is kWaves consecutive chunks, one per wavefront of the workgroup.  Tiles are handed
    // out through an atomic ticket, so a tile only ever waits for tiles already running.  The
    // chunk sizes of a tile are summed through LDS and ONE wavefront runs the look-back for
    // the tile: with thousands of chunks in flight that keeps the chain of dependent
    // descriptor reads short.
    for (;;) {
        if (threadIdx.x == 0) tile_ticket = atomicAdd(ticket, 1u);
        __syncthreads();  // (1) ticket visible; also closes the previous tile's LDS reads
        const uint32_t t = tile_ticket;
        if ((uint64_t)t * kWaves >= nchunks) break;
        const uint32_t c = t * kWaves + wave;

        uint32_t total = 0;
        if (c < nchunks) {
            const uint64_t first = (uint64_t)c * HF_CHUNK_SYMS;
            const uint64_t left = n - first;
            const uint32_t nvalid = left < HF_CHUNK_SYMS ? (uint32_t)left : HF_CHUNK_SYMS;
            const void* chunk_in = reinterpret_cast<const uint8_t*>(data) + first * elem_bytes;
            uint8_t* side_chunk = side ? side + first : nullptr;
            if (PER_CHUNK) {
                const hf_chunk_cb* ccb = reinterpret_cast<const hf_chunk_cb*>(cbv) + c;
#pragma unroll
                for (int k = 0; k < 4; ++k) enc[lane + 64 * k] = ccb->enc[lane + 64 * k];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            uint32_t buf[kDepth][4];
            if (nvalid == HF_CHUNK_SYMS) {
                preload_chunk<LAYOUT>(chunk_in, lane, buf);
                total = encode_chunk<LAYOUT, true>(chunk_in, nvalid, enc, stg, spill, side_chunk,
                                                   lane, buf);
            } else {
                total = encode_chunk<LAYOUT, false>(chunk_in, nvalid, enc, stg, spill, side_chunk,
                                                    lane, buf);
            }
            total = __builtin_amdgcn_readfirstlane(total);
        }
        if (lane == 0) tile_tot[wave] = total;
        __syncthreads();  // (2) all chunk sizes of the tile known; staged words visible
        uint32_t my_off = 0, tile_total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const uint32_t x = tile_tot[w];
            my_off += w < wave ? x : 0u;
            tile_total += x;
        }
        if (wave == 0) {
            if (lane == 0)
                __hip_atomic_store(&desc[t], kDescAggregate | (u64)tile_total, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            const u64 b = lookback(desc, t, lane, err);
            if (lane == 0) {
                __hip_atomic_store(&desc[t], kDescPrefix | (b + tile_total), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                tile_base = b;
            }
        }
        __syncthreads();  // (3) tile base known
        // the prefix is the same in every lane: keep it in scalar registers so the branches
        // below are plain scalar branches
        const u64 tb = tile_base;
        // (the builtin returns int: the halves pass through uint32_t, or the low one would be
        // sign-extended over the high one once the payload passes 2^31 words)
        const uint32_t tb_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tb >> 32));
        const uint32_t tb_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tb);
        const u64 base = (((u64)tb_hi << 32) | (u64)tb_lo) + my_off;
        if (c < nchunks) {
            if (lane == 0) {
                chunk_off[c] = (uint32_t)base;
                if (c == nchunks - 1) chunk_off[nchunks] = (uint32_t)(base + total);
            }
            const bool fits = base + total <= cap_words && base + total <= 0xFFFFFFFFull;
            if (!fits && lane == 0) atomicOr(err, HF_KERR_OVERFLOW);
            const uint32_t ncopy = fits ? total : 0u;  // nothing is written past the buffer
            const uint32_t in_lds = ncopy < (uint32_t)kStgWords ? ncopy : (uint32_t)kStgWords;
            uint32_t* dst = payload + base;
            // four LDS reads in flight per lane, then four coalesced 256-byte stores
            uint32_t i = lane;
            for (; i + 192 < in_lds; i += 256) {
                const uint32_t a0 = stg[i], a1 = stg[i + 64], a2 = stg[i + 128], a3 = stg[i + 192];
                dst[i] = a0;
                dst[i + 64] = a1;
                dst[i + 128] = a2;
                dst[i + 192] = a3;
            }
            for (; i < in_lds; i += 64) dst[i] = stg[i];
            if (ncopy > (uint32_t)kStgWords) {
                // the spill words were written by other lanes of this wavefront: drain the
                // stores, then read them past the vector L1
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (uint32_t k = kStgWords + lane; k < ncopy; k += 64)
                    dst[k] = __hip_atomic_load(&spill[k - kStgWords], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
}

}  // namespace

size_t hf_encode_spill_bytes(int num_cu) {
    return (size_t)num_cu * 2 * kWaves * kSpillWords * sizeof(uint32_t);
}

template <bool PER_CHUNK>
static void launch_encode_impl(const void* d_data, uint64_t n, int layout, const void* cb,
                               uint8_t* d_side, uint32_t* d_payload, uint64_t cap_words,
                               uint32_t* d_chunk_off, unsigned long long* d_desc,
                               uint32_t* d_ticket, uint32_t* d_spill, uint32_t* d_err, int num_cu,
                               hipStream_t s) {
    const uint64_t nchunks = (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS;
    if (nchunks == 0) return;
    uint64_t grid = (nchunks + kWaves - 1) / kWaves;
    const uint64_t cap = (uint64_t)num_cu * 2;  // two ~70 KiB workgroups fit one CU's LDS
    if (grid > cap) grid = cap;
    if (layout == HF_LAYOUT_BF16_EXP)
        hipLaunchKernelGGL((hf_k_encode<HF_LAYOUT_BF16_EXP, PER_CHUNK>), dim3((unsigned)grid),
                           dim3(kWaves * 64), 0, s, d_data, n, (uint32_t)nchunks, cb, d_side,
                           d_payload, cap_words, d_chunk_off, d_desc, d_ticket, d_spill, d_err);
    else
        hipLaunchKernelGGL((hf_k_encode<HF_LAYOUT_U8, PER_CHUNK>), dim3((unsigned)grid),
                           dim3(kWaves * 64), 0, s, d_data, n, (uint32_t)nchunks, cb,
                           (uint8_t*)nullptr, d_payload, cap_words, d_chunk_off, d_desc, d_ticket,
                           d_spill, d_err);
}

void hf_launch_encode(const void* d_data, uint64_t n, int layout, const hf_cb_dev* cb,
                      uint8_t* d_side, uint32_t* d_payload, uint64_t cap_words,
                      uint32_t* d_chunk_off, unsigned long long* d_desc, uint32_t* d_ticket,
                      uint32_t* d_spill, uint32_t* d_err, int num_cu, hipStream_t s) {
    launch_encode_impl<false>(d_data, n, layout, cb, d_side, d_payload, cap_words, d_chunk_off,
                              d_desc, d_ticket, d_spill, d_err, num_cu, s);
}

void hf_launch_encode_chunks(const void* d_data, uint64_t n, int layout, const hf_chunk_cb* cbs,
                             uint8_t* d_side, uint32_t* d_payload, uint64_t cap_words,
                             uint32_t* d_chunk_off, unsigned long long* d_desc,
                             uint32_t* d_ticket, uint32_t* d_spill, uint32_t* d_err, int num_cu,
                             hipStream_t s) {
    launch_encode_impl<true>(d_data, n, layout, cbs, d_side, d_payload, cap_words, d_chunk_off,
                             d_desc, d_ticket, d_spill, d_err, num_cu, s);
}
