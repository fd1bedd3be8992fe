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
// descriptors (run by a fifth, "scout" wavefront of the workgroup) yields its place in the
// payload, and the staged words then leave with coalesced stores.  Tiles are handed out through
// an atomic ticket so a tile only ever waits for tiles whose workgroups are already running.
// With a codebook that promises small chunks a wavefront stages two chunks, and the placement
// of one tile overlaps the encoding of the next (see hf_k_encode).
//
// HBM traffic per bf16 element: 2 B read, 1 B side stream written, ~L/8 B payload written.
#include <atomic>
#include <type_traits>

#include "hf_common.h"

namespace {

typedef unsigned long long u64;

constexpr int kWaves = 4;            // wavefronts per workgroup
constexpr int kStgWords = 4224;      // LDS staging per wavefront (16.5 KiB: 8 bits/symbol + slack)
constexpr int kMaxChunkWords = 129 * 64;                  // 128 code words + 1 look-ahead, per lane
constexpr int kHalfWords = kStgWords / 2;                 // staging of one chunk when two are kept
constexpr int kSpillWords = kMaxChunkWords - kHalfWords;  // global spill per staged chunk
constexpr int kPipeMargin = 192;     // expected chunk size must stay this far below kHalfWords
constexpr int kDepth = 8;            // input groups (rounds) a wavefront keeps in flight

constexpr u64 kDescAggregate = 1ull << 62;
constexpr u64 kDescPrefix = 2ull << 62;
constexpr u64 kDescValueMask = (1ull << 62) - 1ull;
constexpr u64 kSpinTicks = 100000000ull;  // 1 s

constexpr int kLookK = 2;  // descriptors per lane and pass: a pass covers 128 tiles

// Exclusive prefix (in words) of tile t over the tile descriptors.  Each pass inspects the
// 64 * kLookK tiles before the current position, kLookK per lane (lane j, step k: tile
// idx - 64 k - j), all loads in flight together; the nearest descriptor holding an inclusive
// prefix ends the walk.
// A descriptor is one 8-byte {status, value} word written by a single agent-scope store, so a
// reader never needs more than the word itself.
__device__ u64 lookback(const u64* desc, const uint32_t t, const int lane, uint32_t* err) {
    u64 excl = 0;
    long long idx = (long long)t - 1;
    uint32_t spins = 0;
    u64 t0 = 0;
    while (idx >= 0) {
        u64 d[kLookK];
#pragma unroll
        for (int k = 0; k < kLookK; ++k) d[k] = 0;
        for (;;) {
            // only descriptors not seen valid yet are (re)read: every workgroup of the grid polls
            // the same few cache lines, and that traffic is what limits how fast they answer
#pragma unroll
            for (int k = 0; k < kLookK; ++k) {
                const long long j = idx - 64 * k - lane;
                if ((d[k] >> 62) == 0)  // before tile 0: prefix 0
                    d[k] = j >= 0 ? __hip_atomic_load(&desc[j], __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT)
                                  : kDescPrefix;
            }
            // nearest prefix: step kf, lane lf (kLookK = none); all descriptors nearer than it
            // must hold at least an aggregate
            int kf = kLookK, lf = 0;
            bool holes = false;
            u64 v = 0;
#pragma unroll
            for (int k = 0; k < kLookK; ++k) {
                const u64 inv = __ballot((d[k] >> 62) == 0);
                const u64 pre = __ballot((d[k] >> 62) == 2);
                if (kf == kLookK) {  // still in front of the nearest prefix (wave-uniform)
                    u64 upto = ~0ull;  // lanes that count in this step
                    if (pre) {
                        kf = k;
                        lf = (int)__builtin_ctzll(pre);
                        upto = lf >= 63 ? ~0ull : ((2ull << lf) - 1ull);
                    }
                    holes = holes || (inv & upto) != 0;
                    v += ((upto >> lane) & 1ull) ? (d[k] & kDescValueMask) : 0ull;
                }
            }
            if (!holes) {
                excl += hf_wave_sum_u64(v);
                if (kf < kLookK) return excl;
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
            __builtin_amdgcn_s_sleep(4);
        }
        idx -= 64 * kLookK;
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

// The encoder table by LDS byte address: `tab` is the table's address (a multiple of 1024, wave-
// uniform), `x` a value whose bits 2..9 are the symbol.  Address = (x & 0x3FC) | tab is a single
// v_and_or_b32; through a pointer the compiler adds the table's base to the masked index even
// when that base is zero.
typedef __attribute__((address_space(3))) const uint32_t hf_lds_u32;
__device__ __forceinline__ uint32_t enc_lookup(const uint32_t tab, const uint32_t x) {
    uint32_t a;
    if (__builtin_constant_p(tab) && tab == 0u) {
        a = x & 0x3FCu;
    } else {
        // spelled out: the compiler makes an AND and an OR of it (the mask is a literal and the
        // table address a scalar, one scalar operand too many for a single instruction)
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a) : "v"(x), "s"(0x3FCu), "v"(tab));
    }
    return *reinterpret_cast<hf_lds_u32*>((uintptr_t)a);
}
__device__ __forceinline__ uint32_t lds_address(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
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
                                         uint32_t* __restrict__ stg, uint32_t* __restrict__ spill,
                                         const uint32_t limit) {
    if (!CHECK) {
        if (done) stg[slot] = w;
        return;
    }
    const bool in_lds = slot < limit;
    if (done && in_lds) stg[slot] = w;
    if (__builtin_expect(__any(done && !in_lds), 0)) {
        if (done && !in_lds) spill[slot - limit] = w;
    }
}

// One round of one lane: 8 symbols out of `raw`, side bytes stored, codes looked up, the
// decoder's refill schedule replayed and completed words placed.
template <int LAYOUT, bool FULL, bool CHECK>
__device__ __forceinline__ void encode_round(const uint32_t (&raw)[4], const uint32_t g,
                                             const uint32_t nvalid, LaneState& st,
                                             const uint32_t enc,
                                             uint32_t* __restrict__ stg,
                                             uint32_t* __restrict__ spill, const uint32_t limit,
                                             uint8_t* side_chunk, hf_gmem_u8* side_row) {
    uint32_t s4[8];  // the symbols, times four, in bits 2..9 (higher bits: don't care)
    if (LAYOUT == HF_LAYOUT_BF16_EXP) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t d = raw[i];
            s4[2 * i] = d >> 5;
            s4[2 * i + 1] = d >> 21;
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
        for (int i = 0; i < 8; ++i) {
            const int sh = 8 * (i & 3) - 2;
            s4[i] = sh < 0 ? raw[i >> 2] << 2 : raw[i >> 2] >> sh;
        }
    }

    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = enc_lookup(enc, s4[i]);

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
        put_word<CHECK>(done, st.p0, (uint32_t)(st.acc >> (st.ab & 31u)), stg, spill, limit);
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
                                 const uint32_t enc, uint32_t* __restrict__ stg,
                                 uint32_t* __restrict__ spill, const uint32_t limit,
                                 uint8_t* side_chunk, const int lane,
                                 uint32_t (&buf)[kDepth][4]) {
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
                const bool fits = st.pos + 1024u <= limit;  // wave-uniform
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
                            raw, r * 64u + lane, nvalid, st, enc, stg, spill, limit, side_chunk,
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
            encode_round<LAYOUT, false, true>(raw, g, nvalid, st, enc, stg, spill, limit,
                                              side_chunk, nullptr);
        }
    }
    // words asked for but not filled: the zero-padded last word, then an all-zero word
    const uint32_t pending = (uint32_t)(st.nb + (int)st.ab) >> 5;  // 0, 1 or 2 slots
    put_word(pending > 0, st.p0, (uint32_t)(st.acc << (32 - st.ab)), stg, spill, limit);
    put_word(pending > 1, st.p1, 0u, stg, spill, limit);
    return st.pos;
}

template <bool PER_CHUNK>
struct EncLds {
    uint32_t enc[PER_CHUNK ? kWaves * 256 : 256];
    uint32_t stg[kWaves][kStgWords];
    u64 base[2];            // exclusive prefix of the tile staged in half h (from the scout)
    uint32_t tot[2][kWaves];
    uint32_t base_seq[2];   // (iteration + 1) of the tile base[h] belongs to
    uint32_t ticket;
    uint32_t done_cnt;      // encoder arrivals, kWaves per iteration, never reset
};

// Spin on an LDS word until it holds `want` (another wavefront of the workgroup stores it).
__device__ __forceinline__ void wait_lds_eq(uint32_t* p, const uint32_t want) {
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != want)
        __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ void wait_lds_ge(uint32_t* p, const uint32_t want) {
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want)
        __builtin_amdgcn_s_sleep(2);
}

// PER_CHUNK = false: one codebook (hf_cb_dev) for the whole stream, its encoder table shared
// by the workgroup.  PER_CHUNK = true: chunk c is coded with record c of an hf_chunk_cb array
// (per-block codebooks); each wavefront keeps its chunk's table in its own 1 KiB of LDS.
//
// A workgroup is kWaves encoder wavefronts plus one scout.  A tile is kWaves consecutive
// chunks, one per encoder; tiles are handed out through an atomic ticket, so a tile only ever
// waits for tiles already running.  The encoder that finishes a tile last publishes the tile's
// size in its descriptor.  The scout finds the tile's place in the payload (decoupled look-back
// over the descriptors of the earlier tiles), publishes the inclusive prefix and hands the
// place to the encoders through LDS; they then copy the staged words out with coalesced stores.
//
// When the codebook promises chunks of less than half the staging area (`pipelined`), a
// wavefront stages two chunks: while tile i is being encoded into one half, the scout places
// tile i-1, and the encoders copy it out of the other half afterwards.  Waiting for slower
// workgroups ahead in ticket order and the latency of the descriptor reads then overlap a whole
// tile of encoding instead of stalling the workgroup.  A chunk that outgrows its half spills to
// global memory as it would outgrow the whole area.  Otherwise (`lag` = 0) a tile is placed and
// copied out right after it is encoded.
template <int LAYOUT, bool PER_CHUNK>
__global__ __launch_bounds__((kWaves + 1) * 64) __attribute__((amdgpu_waves_per_eu(1, 4))) void
hf_k_encode(const void* __restrict__ data, const uint64_t n, const uint32_t nchunks,
            const void* __restrict__ cbv, uint8_t* __restrict__ side,
            uint32_t* __restrict__ payload, const uint64_t cap_words,
            uint32_t* __restrict__ chunk_off, u64* desc, uint32_t* ticket, uint32_t* spill_base,
            uint32_t* err) {
    // One object, so that the encoder table sits at LDS address 0 and a lookup's address is the
    // symbol times four with nothing added.  It lives in dynamic LDS on purpose: from a static
    // size the compiler derives "at most 3 waves per SIMD" and pads the register allocation to
    // 136 to enforce it, but two 5-wave workgroups per CU put four waves on one SIMD, which
    // only fit at 128 registers or fewer.
    extern __shared__ __attribute__((aligned(16))) unsigned char hf_enc_smem[];
    EncLds<PER_CHUNK>& L = *reinterpret_cast<EncLds<PER_CHUNK>*>(hf_enc_smem);
    const int lane = hf_lane();
    // (read through readfirstlane so that the compiler knows the value is wave-uniform and the
    // role split below is a scalar branch)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool scout = wave == kWaves;
    const int ew = scout ? 0 : wave;  // encoder index (the scout's is never used)
    uint32_t hint = 0;
    if (!PER_CHUNK) {
        const hf_cb_dev* cb = reinterpret_cast<const hf_cb_dev*>(cbv);
        if (threadIdx.x < 256) L.enc[threadIdx.x] = cb->enc[threadIdx.x];
        hint = __builtin_amdgcn_readfirstlane(cb->chunk_words_hint);
    }
    if (threadIdx.x == 0) {
        L.base_seq[0] = L.base_seq[1] = 0;
        L.done_cnt = 0;
    }
    const bool pipelined = hint != 0 && hint + kPipeMargin <= (uint32_t)kHalfWords;
    const uint32_t lag = pipelined ? 1u : 0u;
    const uint32_t limit = pipelined ? (uint32_t)kHalfWords : (uint32_t)kStgWords;
    uint32_t* enc = PER_CHUNK ? L.enc + ew * 256 : L.enc;
    // the table's LDS address, wave-uniform; enc_lookup needs it to be a multiple of 1024
    const uint32_t enc_tab = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(enc));
    if ((enc_tab & 1023u) != 0 || (!PER_CHUNK && enc_tab != 0)) {
        // (the same in every wavefront: the whole grid leaves)
        if (threadIdx.x == 0) atomicOr(err, HF_KERR_CORRUPT);
        return;
    }
    uint32_t* spill_w = spill_base + ((size_t)blockIdx.x * kWaves + ew) * 2 * kSpillWords;
    const size_t elem_bytes = LAYOUT == HF_LAYOUT_BF16_EXP ? 2 : 1;

    // state of the tile that is waiting for its place (iteration i - lag)
    bool pending = false;
    uint32_t p_t = 0, p_total = 0, p_iter = 0;
    for (uint32_t i = 0;; ++i) {
        // (no ticket: the stream is one tile and this its only workgroup -- tile 0, then none.
        // Tile 0 looks back at nothing, so what the descriptors hold does not matter either.)
        if (threadIdx.x == 0) L.ticket = ticket ? atomicAdd(ticket, 1u) : i;
        __syncthreads();  // ticket visible; the copy-out of the previous iteration is complete
        const uint32_t t = L.ticket;
        const bool live = (uint64_t)t * kWaves < nchunks;
        if (!live && !pending) break;
        const uint32_t h = i & lag;  // the half tile i is staged in

        // ---- encoders: tile i ----
        uint32_t total = 0;
        if (!scout && live) {
            const uint32_t c = t * kWaves + ew;
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
                uint32_t* stg = L.stg[ew] + h * kHalfWords;
                uint32_t* spill = spill_w + h * kSpillWords;
                uint32_t buf[kDepth][4];
                if (nvalid == HF_CHUNK_SYMS) {
                    preload_chunk<LAYOUT>(chunk_in, lane, buf);
                    total = encode_chunk<LAYOUT, true>(chunk_in, nvalid, PER_CHUNK ? enc_tab : 0u,
                                                       stg, spill, limit, side_chunk, lane, buf);
                } else {
                    total = encode_chunk<LAYOUT, false>(chunk_in, nvalid, PER_CHUNK ? enc_tab : 0u,
                                                        stg, spill, limit, side_chunk, lane, buf);
                }
                total = __builtin_amdgcn_readfirstlane(total);
            }
            if (lane == 0) {
                L.tot[h][ew] = total;
                // the last encoder to arrive sees every size (the LDS executes one wavefront's
                // store and atomic in order) and publishes the tile's aggregate
                const uint32_t arrived = __hip_atomic_fetch_add(
                    &L.done_cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (arrived == i * kWaves + kWaves - 1) {
                    uint32_t sum = 0;
#pragma unroll
                    for (int w = 0; w < kWaves; ++w) sum += L.tot[h][w];
                    __hip_atomic_store(&desc[t], kDescAggregate | (u64)sum, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }

        // ---- the tile to place in this iteration: tile i (lag 0) or tile i-1 (lag 1) ----
        const bool place = lag ? pending : live;
        const uint32_t q_t = lag ? p_t : t;
        const uint32_t q_iter = lag ? p_iter : i;
        const uint32_t q_total = lag ? p_total : total;
        const uint32_t qh = q_iter & lag;
        if (place) {
            if (scout) {
                // all sizes of the tile are in LDS once its encoders have all arrived
                wait_lds_ge(&L.done_cnt, (q_iter + 1) * kWaves);
                const u64 b = lookback(desc, q_t, lane, err);
                if (lane == 0) {
                    uint32_t sum = 0;
#pragma unroll
                    for (int w = 0; w < kWaves; ++w) sum += L.tot[qh][w];
                    __hip_atomic_store(&desc[q_t], kDescPrefix | (b + sum), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    L.base[qh] = b;
                    __hip_atomic_store(&L.base_seq[qh], q_iter + 1, __ATOMIC_RELEASE,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                wait_lds_ge(&L.done_cnt, (q_iter + 1) * kWaves);  // every tot[qh][*] is written
                wait_lds_eq(&L.base_seq[qh], q_iter + 1);
                uint32_t my_off = 0;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) my_off += w < ew ? L.tot[qh][w] : 0u;
                // the prefix is the same in every lane: keep it in scalar registers so the
                // branches below are plain scalar branches (the builtin returns int: the halves
                // pass through uint32_t, or the low one would be sign-extended over the high one
                // once the payload passes 2^31 words)
                const u64 tb = L.base[qh];
                const uint32_t tb_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tb >> 32));
                const uint32_t tb_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tb);
                const u64 base = (((u64)tb_hi << 32) | (u64)tb_lo) + my_off;
                const uint32_t c = q_t * kWaves + ew;
                if (c < nchunks) {
                    if (lane == 0) {
                        chunk_off[c] = (uint32_t)base;
                        if (c == nchunks - 1) {
                            chunk_off[nchunks] = (uint32_t)(base + q_total);
                            err[1] = (uint32_t)(base + q_total);  // the total, next to the error word
                        }
                    }
                    const bool fits = base + q_total <= cap_words && base + q_total <= 0xFFFFFFFFull;
                    if (!fits && lane == 0) atomicOr(err, HF_KERR_OVERFLOW);
                    const uint32_t ncopy = fits ? q_total : 0u;  // nothing goes past the buffer
                    const uint32_t in_lds = ncopy < limit ? ncopy : limit;
                    const uint32_t* stg = L.stg[ew] + qh * kHalfWords;
                    const uint32_t* spill = spill_w + qh * kSpillWords;
                    uint32_t* dst = payload + base;
                    // four LDS reads in flight per lane, then four coalesced 256-byte stores
                    uint32_t k = lane;
                    for (; k + 192 < in_lds; k += 256) {
                        const uint32_t a0 = stg[k], a1 = stg[k + 64], a2 = stg[k + 128],
                                       a3 = stg[k + 192];
                        dst[k] = a0;
                        dst[k + 64] = a1;
                        dst[k + 128] = a2;
                        dst[k + 192] = a3;
                    }
                    for (; k < in_lds; k += 64) dst[k] = stg[k];
                    if (ncopy > limit) {
                        // the spill words were written by other lanes of this wavefront: drain
                        // the stores, then read them past the vector L1
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        for (uint32_t m = limit + lane; m < ncopy; m += 64)
                            dst[m] = __hip_atomic_load(&spill[m - limit], __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
            }
        }
        pending = lag && live;
        p_t = t;
        p_total = total;
        p_iter = i;
        if (!live) break;
    }
}

// ---- slotted placement (per-block codebooks) ---------------------------------------------------
// Every chunk's place is known before the encoder starts: chunk c owns a slot of
// HF_SLOT_WORDS(B_c) = ceil(B_c / 32) + 126 words, B_c = sum over the chunk's symbols of their code
// lengths, which the per-chunk histogram and the chunk's codebook give without reading the data
// (a lane asks for a word only while it holds fewer than 32 bits, so it ends with at most 62
// unread ones and takes at most floor((T + 62) / 32) words for the T bits of its codes; 64 lanes
// take at most floor(B_c / 32) + 124: no chunk needs more).  Slot offsets are a prefix sum of the
// slot sizes.  With the place known, words go straight to global memory as they complete: no
// staging in LDS, no tile descriptors, no look-back, no scout, and a wavefront needs 1 KiB of LDS
// for its table instead of 17 KiB -- occupancy is bounded by registers, not LDS, and the
// per-wavefront latency of the symbol loop is covered by the other wavefronts.  What a slot has
// left over is zeroed.

// slot size of every chunk into slot[c]; slot[nchunks] = 0 (the scan turns it into the total).
// enc: the code table (code | len << 16 by symbol) of chunk 0; chunk c's lies enc_stride words
// further per chunk (0: one table for the whole stream).
__global__ __launch_bounds__(256) void hf_k_slot_sizes(const uint32_t* __restrict__ freq_chunks,
                                                       const uint32_t* __restrict__ enc,
                                                       const uint32_t enc_stride,
                                                       const uint32_t nchunks,
                                                       uint32_t* __restrict__ slot) {
    const int lane = hf_lane();
    const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (c > nchunks) return;
    if (c == nchunks) {
        if (lane == 0) slot[c] = 0;
        return;
    }
    const uint4 f = reinterpret_cast<const uint4*>(freq_chunks + (size_t)c * 256)[lane];
    const uint4 e = reinterpret_cast<const uint4*>(enc + (size_t)c * enc_stride)[lane];
    uint32_t bits = f.x * (e.x >> 16) + f.y * (e.y >> 16) + f.z * (e.z >> 16) + f.w * (e.w >> 16);
    bits = hf_wave_sum_u32(bits);
    // A chunk's own counts give at most 16384 * 16.  Counts from elsewhere may give anything (the
    // sum may even have wrapped): clamped, every slot stays below 8320 words, the 32-bit sums of
    // the scan cannot wrap, and the encoder finds the chunk too large for its slot.
    const uint32_t most = HF_CHUNK_SYMS * (uint32_t)HF_MAX_CODE_LEN;
    if (lane == 0) slot[c] = HF_SLOT_WORDS(bits < most ? bits : most);
}

// In-place exclusive prefix sum of m 32-bit values, three kernels: 1024 values per workgroup with
// the workgroup's total set aside, the totals scanned by one workgroup, the totals added back.
constexpr int kScanPerGroup = 1024;
__global__ __launch_bounds__(256) void hf_k_scan_local(uint32_t* __restrict__ a, const uint64_t m,
                                                       u64* __restrict__ group_sum) {
    __shared__ uint32_t wave_tot[4];
    const int lane = hf_lane(), wv = threadIdx.x >> 6;
    const uint64_t i0 = (uint64_t)blockIdx.x * kScanPerGroup + threadIdx.x * 4u;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i0 + k < m ? a[i0 + k] : 0u;
    const uint32_t mine = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    uint32_t base = incl - mine;
    for (int w = 0; w < wv; ++w) base += wave_tot[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < m) a[i0 + k] = base;
        base += v[k];
    }
    if (threadIdx.x == 255) group_sum[blockIdx.x] = (u64)base;  // (the group's total)
}
__global__ __launch_bounds__(256) void hf_k_scan_groups(u64* __restrict__ group_sum,
                                                        const uint32_t ngroups) {
    __shared__ u64 wave_tot[4];
    __shared__ u64 carry_s;
    const int lane = hf_lane(), wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t g0 = 0; g0 < ngroups; g0 += 256) {
        const uint32_t g = g0 + threadIdx.x;
        const u64 mine = g < ngroups ? group_sum[g] : 0ull;
        u64 incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u64 t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        u64 base = carry_s + incl - mine;
        for (int w = 0; w < wv; ++w) base += wave_tot[w];
        if (g < ngroups) group_sum[g] = base;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = base + mine;
        __syncthreads();
    }
}
// (limit: what the last value -- the total -- may be at most; above it the stream does not fit the
// payload buffer or its 32-bit offsets.  The total goes to group_sum[number of groups] in 64 bits:
// the encoder compares it with the limit itself and then leaves every chunk alone.)
__global__ __launch_bounds__(256) void hf_k_scan_add(uint32_t* __restrict__ a, const uint64_t m,
                                                     u64* __restrict__ group_sum,
                                                     const u64 limit, uint32_t* err) {
    const uint64_t i0 = (uint64_t)blockIdx.x * kScanPerGroup + threadIdx.x * 4u;
    const u64 add = group_sum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < m) {
            const u64 v = (u64)a[i0 + k] + add;
            a[i0 + k] = (uint32_t)v;
            if (i0 + k == m - 1) {
                group_sum[gridDim.x] = v;
                err[1] = (uint32_t)v;  // the stream's words, where the host reads them
                if (v > limit) atomicOr(err, HF_KERR_OVERFLOW);
            }
        }
    }
}

template <int LAYOUT>
__global__ __launch_bounds__(256) void hf_k_encode_slotted(
    const void* __restrict__ data, const uint64_t n, const uint32_t nchunks,
    const uint32_t* __restrict__ enc_src, const uint32_t enc_stride, uint8_t* __restrict__ side,
    uint32_t* __restrict__ payload, const uint32_t* __restrict__ slot_off,
    uint32_t* __restrict__ chunk_words, uint32_t* __restrict__ dump,
    const u64* __restrict__ total_words, const u64 limit, uint32_t* err) {
    __shared__ __attribute__((aligned(1024))) uint32_t enc_all[4 * 256];
    const int lane = hf_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (the slots do not fit, which the scan has reported: the offsets mean nothing, nothing is
    // written)
    if (*total_words > limit) return;
    uint32_t* enc = enc_all + wave * 256;
    const uint32_t enc_tab = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(enc));
    const size_t elem_bytes = LAYOUT == HF_LAYOUT_BF16_EXP ? 2 : 1;
    const uint32_t nwaves = gridDim.x * 4u;
    bool have_table = false;
    for (uint32_t c = blockIdx.x * 4u + (uint32_t)wave; c < nchunks; c += nwaves) {
        const uint64_t first = (uint64_t)c * HF_CHUNK_SYMS;
        const uint64_t left = n - first;
        const uint32_t nvalid = left < HF_CHUNK_SYMS ? (uint32_t)left : HF_CHUNK_SYMS;
        const void* chunk_in = reinterpret_cast<const uint8_t*>(data) + first * elem_bytes;
        uint8_t* side_chunk = side ? side + first : nullptr;
        const uint32_t off0 = __builtin_amdgcn_readfirstlane(slot_off[c]);
        const uint32_t off_next = __builtin_amdgcn_readfirstlane(slot_off[c + 1]);
        if (off_next < off0 || (u64)off_next > limit) {  // (never with offsets of the scan above)
            if (lane == 0) atomicOr(err, HF_KERR_CORRUPT);
            continue;
        }
        const uint32_t slot = off_next - off0;
        if (enc_stride != 0 || !have_table) {  // (a stream-wide table is fetched once)
            const uint32_t* src = enc_src + (size_t)c * enc_stride;
#pragma unroll
            for (int k = 0; k < 4; ++k) enc[lane + 64 * k] = src[lane + 64 * k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            have_table = true;
        }
        // The slot plays the staging area's part: words below its size land in it.  A word beyond
        // it -- the counts were not this chunk's -- goes to a dump area of one chunk's maximum
        // size that all wavefronts share (what it holds is never read), so nothing is written
        // outside the slot, and the chunk is reported.
        uint32_t* dst = payload + off0;
        uint32_t total;
        uint32_t buf[kDepth][4];
        if (nvalid == HF_CHUNK_SYMS) {
            preload_chunk<LAYOUT>(chunk_in, lane, buf);
            total = encode_chunk<LAYOUT, true>(chunk_in, nvalid, enc_tab, dst, dump, slot,
                                               side_chunk, lane, buf);
        } else {
            total = encode_chunk<LAYOUT, false>(chunk_in, nvalid, enc_tab, dst, dump, slot,
                                                side_chunk, lane, buf);
        }
        total = __builtin_amdgcn_readfirstlane(total);
        if (total > slot) {
            if (lane == 0) atomicOr(err, HF_KERR_CORRUPT);
            total = slot;
        }
        if (lane == 0) chunk_words[c] = total;
        for (uint32_t k = total + (uint32_t)lane; k < slot; k += 64u) dst[k] = 0u;
        // a per-chunk table is rewritten for the next chunk: all lookups of this one are done
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

// (what hf_k_encode_slotted may write behind `dump`: one chunk's largest size)
size_t hf_slot_dump_bytes() { return (size_t)kMaxChunkWords * sizeof(uint32_t); }

size_t hf_scan_aux_bytes(uint64_t m) {
    // (one total per group of the scan, and the grand total behind them)
    return (size_t)((m + kScanPerGroup - 1) / kScanPerGroup + 1) * sizeof(u64);
}

// In-place exclusive prefix sum of m 32-bit values on the stream (d_aux: hf_scan_aux_bytes(m)); the
// last value -- the total -- is left in 64 bits at d_aux[number of groups] and in d_err[1], and a
// total above `limit` sets the overflow bit.
void hf_launch_scan_u32(uint32_t* d_a, uint64_t m, unsigned long long* d_aux,
                        unsigned long long limit, uint32_t* d_err, hipStream_t s) {
    const unsigned ngroups = (unsigned)((m + kScanPerGroup - 1) / kScanPerGroup);
    hipLaunchKernelGGL(hf_k_scan_local, dim3(ngroups), dim3(256), 0, s, d_a, m, d_aux);
    hipLaunchKernelGGL(hf_k_scan_groups, dim3(1), dim3(256), 0, s, d_aux, ngroups);
    hipLaunchKernelGGL(hf_k_scan_add, dim3(ngroups), dim3(256), 0, s, d_a, m, d_aux, limit, d_err);
}

// slot sizes -> slot offsets (in d_slot_off, nchunks + 1 values) -> chunks encoded in their slots.
// d_enc / enc_stride: see hf_k_slot_sizes.
void hf_launch_encode_chunks_slotted(const void* d_data, uint64_t n, int layout,
                                     const uint32_t* d_enc, uint32_t enc_stride,
                                     const uint32_t* d_freq_chunks,
                                     uint8_t* d_side, uint32_t* d_payload, uint64_t cap_words,
                                     uint32_t* d_slot_off, uint32_t* d_chunk_words,
                                     unsigned long long* d_scan_aux, uint32_t* d_dump,
                                     uint32_t* d_err, int num_cu, hipStream_t s) {
    const uint64_t nchunks = (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS;
    if (nchunks == 0) return;
    const uint64_t m = nchunks + 1;
    const unsigned ngroups = (unsigned)((m + kScanPerGroup - 1) / kScanPerGroup);
    hipLaunchKernelGGL(hf_k_slot_sizes, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, s, d_freq_chunks,
                       d_enc, enc_stride, (uint32_t)nchunks, d_slot_off);
    const u64 limit = cap_words < 0xFFFFFFFFull ? cap_words : 0xFFFFFFFFull;
    hf_launch_scan_u32(d_slot_off, m, d_scan_aux, limit, d_err, s);
    const u64* d_total = d_scan_aux + ngroups;
    uint64_t grid = (nchunks + 3) / 4;
    // (measured flat from 4 to 8 workgroups per CU, 7 a hair ahead; 65 registers allow 7
    // wavefronts per SIMD on byte symbols, 108 allow 4 on 16-bit elements)
    const uint64_t cap = (uint64_t)num_cu * 7;
    if (grid > cap) grid = cap;
    if (layout == HF_LAYOUT_BF16_EXP)
        hipLaunchKernelGGL(hf_k_encode_slotted<HF_LAYOUT_BF16_EXP>, dim3((unsigned)grid), dim3(256),
                           0, s, d_data, n, (uint32_t)nchunks, d_enc, enc_stride, d_side, d_payload, d_slot_off,
                           d_chunk_words, d_dump, d_total, limit, d_err);
    else
        hipLaunchKernelGGL(hf_k_encode_slotted<HF_LAYOUT_U8>, dim3((unsigned)grid), dim3(256), 0, s,
                           d_data, n, (uint32_t)nchunks, d_enc, enc_stride, (uint8_t*)nullptr, d_payload,
                           d_slot_off, d_chunk_words, d_dump, d_total, limit, d_err);
}

// A stream of one tile needs neither the ticket nor cleared descriptors: the caller may skip
// clearing them.
bool hf_encode_single_tile(uint64_t nchunks) { return nchunks <= (uint64_t)kWaves; }

size_t hf_encode_spill_bytes(int num_cu) {
    return (size_t)num_cu * 2 * kWaves * 2 * kSpillWords * sizeof(uint32_t);
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
    if (hf_encode_single_tile(nchunks)) d_ticket = nullptr;  // (grid == 1)
    const size_t lds = sizeof(EncLds<PER_CHUNK>);
    // more than 64 KiB of dynamic LDS has to be allowed per kernel and per device: done on a
    // kernel's first launch on a device (one bit per device; repeating it is harmless, so two
    // threads racing here only do it twice)
    static std::atomic<unsigned long long> allowed[2];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    std::atomic<unsigned long long>& done = allowed[layout == HF_LAYOUT_BF16_EXP ? 1 : 0];
    if (!(done.load(std::memory_order_relaxed) & bit)) {
        const hipError_t e =
            layout == HF_LAYOUT_BF16_EXP
                ? hipFuncSetAttribute(
                      reinterpret_cast<const void*>(&hf_k_encode<HF_LAYOUT_BF16_EXP, PER_CHUNK>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                : hipFuncSetAttribute(
                      reinterpret_cast<const void*>(&hf_k_encode<HF_LAYOUT_U8, PER_CHUNK>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) done.fetch_or(bit, std::memory_order_relaxed);
    }
    if (layout == HF_LAYOUT_BF16_EXP)
        hipLaunchKernelGGL((hf_k_encode<HF_LAYOUT_BF16_EXP, PER_CHUNK>), dim3((unsigned)grid),
                           dim3((kWaves + 1) * 64), lds, s, d_data, n, (uint32_t)nchunks, cb, d_side,
                           d_payload, cap_words, d_chunk_off, d_desc, d_ticket, d_spill, d_err);
    else
        hipLaunchKernelGGL((hf_k_encode<HF_LAYOUT_U8, PER_CHUNK>), dim3((unsigned)grid),
                           dim3((kWaves + 1) * 64), lds, s, d_data, n, (uint32_t)nchunks, cb,
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
