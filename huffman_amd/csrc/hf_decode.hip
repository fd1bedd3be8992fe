// Packed-stream decoder.  The reference only names this block (ParrallelHuffmanDecoder,
// decoder/top_decoder.scala:6-10, an empty IO bundle; reports/figs/decoder.svg sketches a
// scheduler feeding parallel sub-decoders and a combiner), so the design is this library's.
//
// One wavefront per chunk of 16384 symbols, 64 lanes decoding 64 interleaved bit strings in
// lock step (layout: hf_encode.hip).  Before each pair of symbols the lanes holding fewer
// than 32 buffered bits take the next payload words, lanes in ascending order, so the
// wavefront consumes the payload strictly front to back.  The payload of a chunk is staged
// into a 4 KiB LDS window with one burst of 16-byte loads (re-staged when a chunk is larger
// than the window), so the symbol loop itself issues no payload loads; side bytes are
// fetched two rounds ahead and outputs leave as one 16-byte store per lane and round.
// Everything is touched once, so loads and stores are non-temporal.
//
// Lookup tables in LDS.  A stream whose codebook is short on average is decoded two symbols per
// lookup: a 4096-entry table indexed by the next 12 bits resolves both symbols of a pair when
// their codes fit into those bits together, which is what the pair-wise refill schedule of the
// format asks for (a wavefront takes the short path when all 64 lanes hit; the codebook kernel
// counts the entries that do and the decoder picks its path from that count).  Otherwise a
// 12-bit first-level table resolves one symbol per lookup.  Either way a code longer than the
// first level takes one more lookup in a 16-entry second-level table named by the first, and
// that case is tested once per pair: an escape entry has length zero, so a lane that meets one
// stands still and its second lookup returns the same entry.
//
// HBM traffic per bf16 element: ~L/8 B payload + 1 B side read, 2 B written.
#include <stddef.h>

#include "hf_common.h"
#include "hf_dec_tables.h"

namespace {

constexpr int kWaves = 8;
constexpr int kStageWords = 1024;   // payload window per wavefront, stream-wide tables (4 KiB)
constexpr int kStageWordsC = 512;   // payload window per wavefront, per-chunk tables (2 KiB)
constexpr int kStageSlack = 64;    // never staged: room for the reads of a clamped position
// per-block tables: an 11-bit first level per wavefront (see hf_dec_tables.h)
typedef ChunkTablesT<11, kStageWordsC> ChunkTables;


// LDS layout of the stream-wide decoder, fixed so that table addresses are compile-time
// constants: first level at 0 (the one-symbol table or the multi-symbol one, whichever the
// codebook calls for), second level behind it, then the wavefronts' payload windows.
constexpr uint32_t kLutAddr = 0;
// Second-level tables in LDS.  A codebook can have up to 256 of them (8 KiB) and the one-symbol
// path keeps them all; the multi-symbol path is only taken by codebooks with at most kSubPair
// (at least 99 % of its 4096 entries resolve two symbols, which leaves fewer than 44 escapes).
constexpr uint32_t kSubAll = 256;
constexpr uint32_t kSubPair = 64;

// a code longer than 12 bits: the first-level entry names a 16-entry table for bits 13..16
// (NSUB, a power of two: tables at SUB_ADDR; a table number is masked to it, so that a damaged
// table cannot send a read past them)
template <uint32_t SUB_ADDR, uint32_t NSUB>
__device__ __forceinline__ uint32_t second_level(const uint32_t top, const uint32_t esc,
                                                 uint32_t& len) {
    const uint32_t idx = ((esc & (NSUB - 1u)) << 4) | ((top >> (32 - HF_LUT_BITS - 4)) & 0xFu);
    const uint32_t e2 = lds_u16((idx << 1) + SUB_ADDR);
    len = e2 & 0xFFu;
    return e2 >> 8;
}

// Tables of a stream-wide codebook (any prefix-free code), shared by the workgroup, one symbol
// per lookup.  Entries: symbol << 8 | length, length 0 = escape with the second-level table's
// number in the symbol field.
template <uint32_t SUB_ADDR>
struct DecTables {
    static constexpr int kBits = HF_LUT_BITS;
    static constexpr int kStage = kStageWords;
    static constexpr bool kPairs = false;
    // the escape test once per pair: measured 2 % faster than once per symbol on a stream whose
    // codes all fit the first level, no different on one where 0.4 % of the symbols do not
    static constexpr bool kPairCheck = true;

    // first level, by the next 12 bits (2-byte entries)
    __device__ __forceinline__ uint32_t lookup(const uint32_t top) const {
        return lds_u16(((top >> (32 - kBits - 1)) & ((HF_LUT_SIZE - 1) * 2u)) + kLutAddr);
    }
    __device__ __forceinline__ uint32_t long_code(const uint32_t top, const uint32_t esc,
                                                  uint32_t& len) const {
        return second_level<SUB_ADDR, kSubAll>(top, esc, len);
    }
};

// The same codebook through the multi-symbol table (layout: hf_cb_dev::lut2).  `lookup` is its
// one-symbol view, used where a pair does not resolve in one go.
struct PairTables {
    static constexpr int kBits = HF_LUT_BITS;
    static constexpr int kStage = kStageWords;
    static constexpr bool kPairs = true;

    // by the next 12 bits (4-byte entries)
    __device__ __forceinline__ uint32_t lookup2(const uint32_t top) const {
        return lds_u32(((top >> (32 - kBits - 2)) & ((HF_LUT_SIZE - 1) * 4u)) + kLutAddr);
    }
    __device__ __forceinline__ uint32_t lookup(const uint32_t top) const {
        const uint32_t E = lookup2(top);
        return (E & 0xFF00u) | ((E >> 24) & 0x7Fu);
    }
    __device__ __forceinline__ uint32_t long_code(const uint32_t top, const uint32_t esc,
                                                  uint32_t& len) const {
        return second_level<HF_LUT_SIZE * 4, kSubPair>(top, esc, len);
    }
};



// A pair through the multi-symbol table.  Returns the symbols in bytes 1 and 2.  Every lane
// consumes what its entry resolves -- both codes, the first code, or nothing (escape); the lanes
// that are not through then finish the pair one symbol at a time, the others waiting.
__device__ __forceinline__ uint32_t decode_pair_multi(u64& win, int& room, const PairTables& T,
                                                      uint32_t* err) {
    const uint32_t top = (uint32_t)(win >> 32);
    const uint32_t E = T.lookup2(top);
    win <<= (E & 63u);
    room = add_byte0(room, E);
    uint32_t P = E;
    if (__builtin_expect(__ballot((int)E >= 0) != 0ull, 0)) {  // wave-uniform
        if ((int)E >= 0) {
            if ((E & 0xFFu) == 0) {  // the first code is longer than the table is wide
                uint32_t len;
                const uint32_t sym = T.long_code(top, (E >> 8) & 0xFFu, len);
                if (len == 0) {  // no code matches: flag it and keep moving
                    atomicOr(err, HF_KERR_CORRUPT);
                    len = 1;
                }
                win <<= len;
                room += (int)len;
                P = sym << 8;
            }
            const uint32_t e1 = decode_one(win, room, T, err);
            P = (P & 0xFF00u) | ((e1 & 0xFF00u) << 8);
        }
    }
    return P;
}

// Copy payload words [gs, gs + W) into the wavefront's LDS window; gs is a multiple of 4 so
// every access is a 16-byte one.  A well-formed chunk consumes no word at or past its end
// `off1`, so the loads of the window's part behind the chunk's last 16-byte group are redirected
// to that group (as are loads past the buffer, to its last whole 16 bytes): all loads are issued
// back to back without branches, and the redirected ones fall on one line and move no further
// bytes.  A buffer whose size is not a multiple of four words ends in a group of one to three
// words; when the chunk reaches into it (the stream's last chunk in a buffer of exactly the
// stream's size) those words are fetched one by one behind a wave-uniform branch, so nothing is
// ever read past the buffer.
template <int W>
__device__ __forceinline__ void stage_window_load(const uint32_t* __restrict__ payload,
                                                  const u64 payload_words, const u64 off1,
                                                  const u64 gs, const int lane,
                                                  uint4 (&v)[W / 256]) {
    const u64 whole = payload_words & ~3ull;  // words in whole 16-byte groups (>= 4: the caller
    const u64 last_buf = whole - 4;           // guarantees payload_words >= 4)
    const u64 last_chunk = (off1 ? off1 - 1 : 0) & ~3ull;
    const u64 last = last_chunk < last_buf ? last_chunk : last_buf;
#pragma unroll
    for (int k = 0; k < W / 256; ++k) {
        u64 idx = gs + (u64)(k * 256 + lane * 4);
        idx = idx < last ? idx : last;
        const hf_u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const hf_u32x4*>(payload + idx));
        v[k] = make_uint4(t.x, t.y, t.z, t.w);
    }
    if (__builtin_expect(payload_words != whole && last_chunk >= whole, 0)) {
#pragma unroll
        for (int k = 0; k < W / 256; ++k) {
            if (gs + (u64)(k * 256 + lane * 4) == whole) {
                v[k].x = payload[whole];
                if (whole + 1 < payload_words) v[k].y = payload[whole + 1];
                if (whole + 2 < payload_words) v[k].z = payload[whole + 2];
            }
        }
    }
}
template <int W>
__device__ __forceinline__ void stage_window_store(uint32_t* __restrict__ stg, const int lane,
                                                   const uint4 (&v)[W / 256]) {
#pragma unroll
    for (int k = 0; k < W / 256; ++k)
        *reinterpret_cast<uint4*>(stg + k * 256 + lane * 4) = v[k];
}
template <int W>
__device__ __forceinline__ void stage_window(const uint32_t* __restrict__ payload,
                                             const u64 payload_words, const u64 off1, const u64 gs,
                                             uint32_t* __restrict__ stg, const int lane) {
    uint4 v[W / 256];
    stage_window_load<W>(payload, payload_words, off1, gs, lane, v);
    stage_window_store<W>(stg, lane, v);
}

// refill step shared by both paths: lanes in `need` take consecutive words from the window
template <int W>
__device__ __forceinline__ void refill(const bool need, u64& win, int& room, uint32_t& lpos,
                                       const uint32_t* __restrict__ stg) {
    const u64 m = __ballot(need);
    if (m != 0) {  // wave-uniform
        // window index = lpos + (needing lanes below this one); v_mbcnt accumulates onto the
        // base.  A well-formed chunk never moves lpos past the staged words; the scalar clamp
        // plus the kStageSlack words behind the window keep a corrupt one inside the array.
        const uint32_t base = lpos < (uint32_t)W ? lpos : (uint32_t)W;
        const uint32_t idx = __builtin_amdgcn_mbcnt_hi(
            (uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base));
        const uint32_t w = stg[idx];
        if (need) {  // 1 <= room <= 32 here
            win |= (u64)w << (room & 63);
            room -= 32;
        }
        lpos += (uint32_t)__popcll(m);
    }
}

// A full chunk (16384 symbols).  Returns the global index of the first unread payload word.
template <int LAYOUT, class TAB>
__device__ u64 decode_chunk_full(const uint32_t* __restrict__ payload, const u64 payload_words,
                                 const u64 off0, const u64 off1,
                                 const uint8_t* __restrict__ side_chunk,
                                 void* __restrict__ out_chunk, const TAB& T,
                                 uint32_t* __restrict__ stg, uint32_t* err, const int lane,
                                 const bool staged = false) {
    u64 gs = off0 & ~3ull;
    // (`staged`, wave-uniform: the caller has put the first window in place already)
    if (!staged) stage_window<TAB::kStage>(payload, payload_words, off1, gs, stg, lane);
    uint32_t lpos = (uint32_t)(off0 - gs);  // window index of the next unread word
    u64 win = 0;                            // buffered bits, MSB aligned
    int room = 32;                          // 32 - buffered bit count

    // chunk bases as scalar registers: each round adds a scalar offset, the lane's part stays a
    // constant 32-bit vector offset, and no per-round vector address arithmetic is left
    const hf_gmem_u8* const su = hf_uniform_ptr(side_chunk);
    hf_gmem_u8* const ou = hf_uniform_ptr(out_chunk);
    const uint32_t lane8 = (uint32_t)lane * 8u, lane16 = (uint32_t)lane * 16u;
    // side bytes of even rounds live in sdA, of odd rounds in sdB; each is reloaded for two
    // rounds ahead right after its use, so no register is rotated and every wait is counted
    hf_u32x2 sdA = {0u, 0u}, sdB = {0u, 0u};
    if (LAYOUT == HF_LAYOUT_BF16_EXP) {
        sdA = __builtin_nontemporal_load(reinterpret_cast<const hf_gmem_u32x2*>(su + lane8));
        sdB = __builtin_nontemporal_load(reinterpret_cast<const hf_gmem_u32x2*>(su + 512 + lane8));
    }
    auto round = [&](const uint32_t r, hf_u32x2& sd) {
        // a round takes at most 4 x 64 words: slide the window when it could run out
        if (lpos + 256u > (uint32_t)TAB::kStage && gs + TAB::kStage < off1) {
            gs = (gs + lpos) & ~3ull;
            lpos = lpos & 3u;
            stage_window<TAB::kStage>(payload, payload_words, off1, gs, stg, lane);
        }
        // one-symbol tables: e[] holds the eight entries, symbol in byte 1; the multi-symbol
        // table: P[] holds the four pairs, symbols in bytes 1 and 2
        uint32_t e[8], P[4];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            refill<TAB::kStage>(room > 0, win, room, lpos, stg);
            if constexpr (TAB::kPairs)
                P[jp] = decode_pair_multi(win, room, T, err);
            else if constexpr (TAB::kPairCheck)
                decode_pair_single(win, room, T, err, e[2 * jp], e[2 * jp + 1]);
            else {
                e[2 * jp] = decode_one(win, room, T, err);
                e[2 * jp + 1] = decode_one(win, room, T, err);
            }
        }
        if (LAYOUT == HF_LAYOUT_BF16_EXP) {
            // two bf16 per dword with packed operations (v_perm_b32; selector byte 0x0C reads
            // as zero): the exponents and the side bytes go to bytes 0 and 2, then the sign
            // moves from bit 7 to bit 15 of each half and the exponent slots in at bit 7
            // sign move: with x = t & 0x00800080, t + 255 * x clears bit 7 and sets bit 15
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t t = __builtin_amdgcn_perm(0u, i < 2 ? sd.x : sd.y,
                                                         (i & 1) ? 0x0C030C02u : 0x0C010C00u);
                uint32_t expo;
                if constexpr (TAB::kPairs)
                    expo = __builtin_amdgcn_perm(0u, P[i], 0x0C020C01u);
                else
                    expo = __builtin_amdgcn_perm(e[2 * i + 1], e[2 * i], 0x0C050C01u);
                // the two three-operand forms are spelled out: the compiler splits them otherwise
                uint32_t sm;
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(sm) : "v"(t & 0x00800080u), "s"(255u), "v"(t));
                asm("v_lshl_or_b32 %0, %1, 7, %2" : "=v"(o[i]) : "v"(expo), "v"(sm));
            }
            // side bytes for two rounds ahead; the last two rounds reload round 31's
            const uint32_t pf = r + 2u < HF_ROUNDS ? r + 2u : HF_ROUNDS - 1u;
            sd = __builtin_nontemporal_load(
                reinterpret_cast<const hf_gmem_u32x2*>(su + pf * 512u + lane8));
            __builtin_nontemporal_store(hf_u32x4{o[0], o[1], o[2], o[3]},
                                        reinterpret_cast<hf_gmem_u32x4*>(ou + r * 1024u + lane16));
        } else if constexpr (TAB::kPairs) {
            const uint32_t lo = __builtin_amdgcn_perm(P[1], P[0], 0x06050201u);
            const uint32_t hi = __builtin_amdgcn_perm(P[3], P[2], 0x06050201u);
            __builtin_nontemporal_store(hf_u32x2{lo, hi},
                                        reinterpret_cast<hf_gmem_u32x2*>(ou + r * 512u + lane8));
        } else {
            const uint32_t a = __builtin_amdgcn_perm(e[1], e[0], 0x0C0C0501u);
            const uint32_t b = __builtin_amdgcn_perm(e[3], e[2], 0x0C0C0501u);
            const uint32_t c = __builtin_amdgcn_perm(e[5], e[4], 0x0C0C0501u);
            const uint32_t d = __builtin_amdgcn_perm(e[7], e[6], 0x0C0C0501u);
            __builtin_nontemporal_store(hf_u32x2{a | (b << 16), c | (d << 16)},
                                        reinterpret_cast<hf_gmem_u32x2*>(ou + r * 512u + lane8));
        }
    };
    for (uint32_t r = 0; r < HF_ROUNDS; r += 2) {
        round(r, sdA);
        round(r + 1, sdB);
    }
    return gs + lpos;
}

// The last, partial chunk: same schedule with per-symbol validity; speed does not matter.
template <int LAYOUT, class TAB>
__device__ u64 decode_chunk_tail(const uint32_t* __restrict__ payload, const u64 payload_words,
                                 const u64 off0, const u64 off1,
                                 const uint8_t* __restrict__ side_chunk,
                                 void* __restrict__ out_chunk, const uint32_t nvalid,
                                 const TAB& T, uint32_t* __restrict__ stg, uint32_t* err,
                                 const int lane, const bool staged = false) {
    u64 gs = off0 & ~3ull;
    if (!staged) stage_window<TAB::kStage>(payload, payload_words, off1, gs, stg, lane);
    uint32_t lpos = (uint32_t)(off0 - gs);
    u64 win = 0;
    int room = 32;
    const uint32_t rounds = (nvalid + 511u) / 512u;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t g = r * 64u + lane;
        if (lpos + 256u > (uint32_t)TAB::kStage && gs + TAB::kStage < off1) {
            gs = (gs + lpos) & ~3ull;
            lpos = lpos & 3u;
            stage_window<TAB::kStage>(payload, payload_words, off1, gs, stg, lane);
        }
        uint32_t s[8];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            const uint32_t i0 = g * 8 + 2 * jp;
            const bool va = i0 < nvalid, vb = i0 + 1 < nvalid;
            refill<TAB::kStage>(va && room > 0, win, room, lpos, stg);
            s[2 * jp] = va ? decode_one(win, room, T, err) >> 8 : 0u;
            s[2 * jp + 1] = vb ? decode_one(win, room, T, err) >> 8 : 0u;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (g * 8 + i < nvalid) {
                if (LAYOUT == HF_LAYOUT_BF16_EXP) {
                    const uint32_t b = side_chunk[(size_t)g * 8 + i];
                    reinterpret_cast<uint16_t*>(out_chunk)[(size_t)g * 8 + i] =
                        (uint16_t)(((b & 0x80u) << 8) | (s[i] << 7) | (b & 0x7Fu));
                } else {
                    reinterpret_cast<uint8_t*>(out_chunk)[(size_t)g * 8 + i] = (uint8_t)s[i];
                }
            }
        }
    }
    return gs + lpos;
}

// LDS of the stream-wide decoder.  Byte streams may be decoded through the multi-symbol table
// (4-byte entries), so their first level takes 16 KiB; 16-bit elements are bound by HBM either
// way (the multi-symbol path measured 2 % slower there: twice the table to copy per workgroup,
// nothing gained per symbol), so their kernel has the one-symbol table only.
// Second-level tables: the one-symbol path keeps all 256 -- behind the first level, which for byte
// streams is the upper half of the 16 KiB (the one-symbol table fills the lower half only); the
// multi-symbol path keeps kSubPair behind its table.
template <bool PAIRS_POSSIBLE>
struct DecLds {
    union {
        uint16_t lut[HF_LUT_SIZE];
        uint32_t lut2[PAIRS_POSSIBLE ? HF_LUT_SIZE : HF_LUT_SIZE / 2];
    };
    uint16_t sub[(PAIRS_POSSIBLE ? kSubPair : kSubAll) * 16];
    uint32_t stage[kWaves][kStageWords + kStageSlack];
};

// the chunks of one wavefront, through one kind of table
template <int LAYOUT, class TAB>
__device__ __forceinline__ void decode_chunks_of_wave(
    const uint32_t* __restrict__ payload, const uint64_t payload_words,
    const uint32_t* __restrict__ chunk_off, const uint32_t* __restrict__ chunk_words,
    const uint8_t* __restrict__ side, const uint64_t n,
    const uint32_t nchunks, void* __restrict__ out, const TAB& T, uint32_t* __restrict__ stg,
    uint32_t* err, const uint32_t wave, const uint32_t nwaves, bool staged, const int lane) {
    const size_t elem_bytes = LAYOUT == HF_LAYOUT_BF16_EXP ? 2 : 1;
    for (uint32_t c = wave; c < nchunks; c += nwaves) {
        const uint32_t off0 = __builtin_amdgcn_readfirstlane(chunk_off[c]);
        uint32_t off1 = __builtin_amdgcn_readfirstlane(chunk_off[c + 1]);
        // offsets that run backwards or past the payload buffer are never followed
        bool sane = off1 >= off0 && (uint64_t)off1 <= payload_words;
        if (chunk_words) {
            // a slotted stream: the chunk's words are the first chunk_words[c] of its slot
            const uint32_t w = __builtin_amdgcn_readfirstlane(chunk_words[c]);
            sane = sane && w <= off1 - off0;
            off1 = off0 + w;
        }
        if (!sane) {
            if (lane == 0) atomicOr(err, HF_KERR_CORRUPT);
            staged = false;
            continue;
        }
        const uint64_t first = (uint64_t)c * HF_CHUNK_SYMS;
        const uint64_t left = n - first;
        const uint32_t nvalid = left < HF_CHUNK_SYMS ? (uint32_t)left : HF_CHUNK_SYMS;
        const uint8_t* side_chunk = side ? side + first : nullptr;
        void* out_chunk = reinterpret_cast<uint8_t*>(out) + first * elem_bytes;
        u64 end;
        if (nvalid == HF_CHUNK_SYMS)
            end = decode_chunk_full<LAYOUT>(payload, payload_words, off0, off1, side_chunk,
                                            out_chunk, T, stg, err, lane, staged);
        else
            end = decode_chunk_tail<LAYOUT>(payload, payload_words, off0, off1, side_chunk,
                                            out_chunk, nvalid, T, stg, err, lane, staged);
        staged = false;  // only the wavefront's first chunk was staged ahead
        // a well-formed chunk is consumed exactly
        if (end != off1 && lane == 0) atomicOr(err, HF_KERR_CORRUPT);
    }
}

template <int LAYOUT>
__global__ __launch_bounds__(kWaves * 64) __attribute__((amdgpu_waves_per_eu(4, 6))) void
hf_k_decode(const uint32_t* __restrict__ payload, const uint64_t payload_words,
            const uint32_t* __restrict__ chunk_off, const uint32_t* __restrict__ chunk_words,
            const uint8_t* __restrict__ side,
            const uint64_t n, const uint32_t nchunks, const hf_cb_dev* __restrict__ cb,
            void* __restrict__ out, uint32_t* err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hf_dec_smem[];
    constexpr bool kPairsPossible = LAYOUT != HF_LAYOUT_BF16_EXP;
    typedef DecLds<kPairsPossible> Lds;
    // where the one-symbol path finds its second-level tables / where the multi-symbol path does
    constexpr uint32_t kSubAddr = (uint32_t)(HF_LUT_SIZE * 2);
    constexpr uint32_t kSubAddrPair = (uint32_t)offsetof(Lds, sub);
    static_assert(offsetof(Lds, lut) == kLutAddr && (!kPairsPossible || kSubAddrPair == HF_LUT_SIZE * 4) &&
                      (kPairsPossible || offsetof(Lds, sub) == kSubAddr),
                  "the tables' LDS addresses are compile-time constants");
    Lds& S = *reinterpret_cast<Lds*>(hf_dec_smem);
    const int lane = hf_lane();
    // the table lookups take the block to start at LDS address 0 (it is the kernel's only LDS)
    if (lds_address(hf_dec_smem) != 0u) {
        if (threadIdx.x == 0) atomicOr(err, HF_KERR_CORRUPT);
        return;
    }
    const int wv = threadIdx.x >> 6;
    uint32_t* stg = S.stage[wv];
    const uint32_t wave = blockIdx.x * kWaves + wv;
    const uint32_t nwaves = gridDim.x * kWaves;
    // A workgroup lives for one chunk per wavefront, so what it waits for at its start counts.
    // Everything it needs is requested in one go, without a branch in between: the chunk's
    // offsets first (the window's addresses depend on them), the first-level table (16-byte
    // pieces), the number of second-level tables, then the chunk's first payload window and the
    // second-level tables in use; the LDS stores and the barrier follow.  Two memory round trips
    // instead of five.
    constexpr int kThreads = kWaves * 64;
    static_assert(kThreads * 16 >= HF_LUT_SIZE * 2 && kThreads * 32 >= HF_LUT_SIZE * 4 &&
                  kThreads >= (int)kSubAll * 2, "one piece of the one-symbol table, two of the multi-symbol one, "
                                   "one of the second level per thread");
    const uint32_t cw = wave < nchunks ? wave : nchunks - 1u;  // (nchunks >= 1)
    const uint32_t o0 = __builtin_amdgcn_readfirstlane(chunk_off[cw]);
    uint32_t o1 = __builtin_amdgcn_readfirstlane(chunk_off[cw + 1]);
    bool slot_ok = true;
    if (chunk_words) {  // a slotted stream (see decode_chunks_of_wave)
        const uint32_t w = __builtin_amdgcn_readfirstlane(chunk_words[cw]);
        slot_ok = o1 >= o0 && w <= o1 - o0;
        o1 = o0 + w;
    }
    const uint32_t n_sub = (uint32_t)__builtin_amdgcn_readfirstlane((int)cb->n_sub);
    const bool pairs = kPairsPossible && n_sub <= kSubPair &&
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)cb->pair_hits) >= HF_PAIR_HITS_MIN;
    // first level: 512 pieces of the one-symbol table, or 1024 of the multi-symbol one
    const uint4* t_src = pairs ? reinterpret_cast<const uint4*>(cb->lut2)
                               : reinterpret_cast<const uint4*>(cb->lut);
    const uint32_t t_n = pairs ? HF_LUT_SIZE * 4 / 16 : HF_LUT_SIZE * 2 / 16;
    const bool has_a = threadIdx.x < t_n, has_b = threadIdx.x + kThreads < t_n;
    const uint4 t_a = t_src[has_a ? threadIdx.x : 0];
    const uint4 t_b = t_src[has_b ? threadIdx.x + kThreads : 0];
    const uint32_t nsub = n_sub < kSubAll ? n_sub : kSubAll;
    bool staged = wave < nchunks && slot_ok && o1 >= o0 && (uint64_t)o1 <= payload_words;
    uint4 w[kStageWords / 256];
    stage_window_load<kStageWords>(payload, payload_words, staged ? o1 : 0u,
                                   staged ? (u64)o0 & ~3ull : 0ull, lane, w);
    const bool has_sub = threadIdx.x < nsub * 2;  // 16 entries = two 16-byte pieces per table
    const uint4 t_sub = reinterpret_cast<const uint4*>(cb->sub)[has_sub ? threadIdx.x : 0];
    if (has_a) reinterpret_cast<uint4*>(S.lut2)[threadIdx.x] = t_a;
    if (has_b) reinterpret_cast<uint4*>(S.lut2)[threadIdx.x + kThreads] = t_b;
    // (a second-level table no first-level entry names is never read, whatever it holds)
    if (has_sub)
        reinterpret_cast<uint4*>(hf_dec_smem + (pairs ? kSubAddrPair : kSubAddr))[threadIdx.x] = t_sub;
    stage_window_store<kStageWords>(stg, lane, w);  // (unused when !staged)
    __syncthreads();
    if (kPairsPossible && pairs) {
        PairTables T;
        decode_chunks_of_wave<LAYOUT>(payload, payload_words, chunk_off, chunk_words, side, n,
                                      nchunks, out, T, stg, err, wave, nwaves, staged, lane);
    } else {
        DecTables<kSubAddr> T;
        decode_chunks_of_wave<LAYOUT>(payload, payload_words, chunk_off, chunk_words, side, n,
                                      nchunks, out, T, stg, err, wave, nwaves, staged, lane);
    }
}

// ---- per-chunk codebooks -------------------------------------------------------------------
constexpr int kWavesC = 4;

// Rebuild one chunk's decode tables from its canonical order.  Lane l holds ranks 4l .. 4l+3.
// The codes of ranks 0, 1, 2, ... tile the left-aligned 16-bit code space front to back, rank p
// covering 2^(16 - len_p) values, so a prefix sum of those widths gives every code.
constexpr int kTailEntries = 512;  // direct table for the long codes when their space fits

// `tmp` (256 words) may alias the payload window: it is only used before the window is staged.
// Returns false (wave-uniform), with nothing but cleared tables behind it, when the
// record is not a canonical order: lengths of 1..16 that never decrease with the rank (a zero,
// "symbol absent", only after them) and fill no more than the code space.  That is what keeps
// every span below inside its table, so a damaged record cannot reach a neighbour's tables.
__device__ bool build_chunk_tables(const hf_chunk_cb* __restrict__ ccb, uint16_t* __restrict__ lut,
                                   uint8_t* __restrict__ r2s, uint32_t* __restrict__ tmp,
                                   uint16_t* __restrict__ tail, ChunkTables& T, const int lane) {
    constexpr int B = ChunkTables::kBits;
    constexpr int kLong = ChunkTables::kLong;
    const uint32_t lens4 = reinterpret_cast<const uint32_t*>(ccb->len_by_rank)[lane];
    const uint32_t syms4 = reinterpret_cast<const uint32_t*>(ccb->rank2sym)[lane];
    reinterpret_cast<uint32_t*>(r2s)[lane] = syms4;
    // zero the first level: entries no short code covers stay "escape"
#pragma unroll
    for (int k = 0; k < (1 << B) * 2 / 16 / 64; ++k)
        reinterpret_cast<uint4*>(lut)[k * 64 + lane] = make_uint4(0, 0, 0, 0);
    uint32_t len[4], width[4], start[4];
    uint32_t lane_total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        len[i] = (lens4 >> (8 * i)) & 0xFFu;
        width[i] = (len[i] >= 1 && len[i] <= 16) ? (1u << (16 - len[i])) : 0u;
        start[i] = lane_total;
        lane_total += width[i];
    }
    uint32_t incl = lane_total;  // inclusive scan over lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    const uint32_t lane_base = incl - lane_total;
    bool valid;
    {
        const uint32_t before = __shfl_up(len[3], 1);
        bool ok = incl <= 65536u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t a = i ? len[i - 1] : (lane ? before : 1u);
            ok = ok && len[i] <= 16u && (len[i] == 0u || (a != 0u && a <= len[i]));
        }
        valid = __ballot(!ok) == 0ull;
        // a refused record builds nothing: with every length cleared no span below is written
        if (!valid) len[0] = len[1] = len[2] = len[3] = 0u;
    }
    // blocks of the long codes: where each long length starts in the code space and the rank of
    // its first code.  Everything before them is the short codes taken together.
    uint32_t cnt_below = 0, short_width = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool is_short = len[i] >= 1 && len[i] <= (uint32_t)B;
        cnt_below += (uint32_t)__popcll(__ballot(is_short));
        short_width += is_short ? width[i] : 0u;
    }
    uint32_t start_acc = hf_wave_sum_u32(short_width);
#pragma unroll
    for (uint32_t d = B + 1; d <= 16; ++d) {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) c += (uint32_t)__popcll(__ballot(len[i] == d));
        T.start[d - B - 1] = start_acc;
        T.off[d - B - 1] = cnt_below;
        T.end[d - B - 1] = start_acc + (c << (16 - d));
        cnt_below += c;
        start_acc += c << (16 - d);
    }
    // The long codes occupy [start[0], end[kLong-1]) of the 16-bit code space; when that is at
    // most kTailEntries values (rare symbols: the usual case) they get a direct table, so a long
    // code costs one more lookup instead of the block arithmetic.
    const uint32_t tail_span = T.end[kLong - 1] - T.start[0];
    T.tail_n = tail_span <= (uint32_t)kTailEntries ? tail_span : 0u;
    if (T.tail_n != 0) {
#pragma unroll
        for (int k = 0; k < kTailEntries * 2 / 16 / 32; ++k)
            if (k * 64 + lane < kTailEntries * 2 / 16)
                reinterpret_cast<uint4*>(tail)[k * 64 + lane] = make_uint4(0, 0, 0, 0);
    }
    // per-rank fill descriptor for the cooperative phase: first entry | len << 12 | sym << 24
    uint32_t n_wide = 0;  // ranks with span >= 64 (len <= B - 6); they are the first ranks
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t s16 = lane_base + start[i];
        const uint32_t sym = (syms4 >> (8 * i)) & 0xFFu;
        tmp[lane * 4 + i] = ((s16 >> (16 - B)) & ((1u << B) - 1u)) | (len[i] << 12) | (sym << 24);
        n_wide += (uint32_t)__popcll(__ballot(len[i] >= 1 && len[i] <= B - 6));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // wide codes: the whole wavefront fills one code's span, 64 entries per step
    for (uint32_t q = 0; q < n_wide; ++q) {
        const uint32_t d = tmp[q];
        const uint32_t first = d & 0xFFFu, L = (d >> 12) & 0x1Fu;
        const uint16_t val = (uint16_t)(((d >> 24) << 8) | L);
        const uint32_t span = 1u << (B - L);
        for (uint32_t k = lane; k < span; k += 64) lut[first + k] = val;
    }
    // narrow codes (at most 32 entries each): every lane fills its own ranks.  A code's span
    // starts at a multiple of its size (canonical codes, lengths never decreasing), so the span
    // is written with the widest aligned store that fits: 16 bytes for 8 entries and more.
    auto fill = [](uint16_t* base, const uint32_t first, const uint32_t span, const uint16_t val) {
        const uint32_t v2 = (uint32_t)val | ((uint32_t)val << 16);
        if (span >= 8) {
            for (uint32_t k = 0; k < span; k += 8)
                *reinterpret_cast<uint4*>(base + first + k) = make_uint4(v2, v2, v2, v2);
        } else if (span == 4) {
            *reinterpret_cast<uint2*>(base + first) = make_uint2(v2, v2);
        } else if (span == 2) {
            *reinterpret_cast<uint32_t*>(base + first) = v2;
        } else {
            base[first] = val;
        }
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint16_t val = (uint16_t)((((syms4 >> (8 * i)) & 0xFFu) << 8) | len[i]);
        if (len[i] >= B - 5 && len[i] <= B) {
            const uint32_t first = ((lane_base + start[i]) >> (16 - B)) & ((1u << B) - 1u);
            fill(lut, first, 1u << (B - len[i]), val);
        }
        if (T.tail_n != 0 && len[i] > B && len[i] <= 16) {
            // (the long codes' part of the code space starts at a multiple of 2^(16-B), so a
            // span keeps its alignment relative to the table)
            fill(tail, lane_base + start[i] - T.start[0], 1u << (16 - len[i]), val);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    T.lut_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_address(lut));
    T.rank2sym = r2s;
    T.tail = tail;
    return valid;
}

template <int LAYOUT>
__global__ __launch_bounds__(kWavesC * 64) void hf_k_decode_chunks(
    const uint32_t* __restrict__ payload, const uint64_t payload_words,
    const uint32_t* __restrict__ chunk_off, const uint32_t* __restrict__ chunk_words,
    const uint8_t* __restrict__ side, const uint64_t n,
    const uint32_t nchunks, const hf_chunk_cb* __restrict__ cbs,
    const uint32_t* __restrict__ chunk_ids, const uint32_t* __restrict__ out_slots,
    const uint32_t n_slots, const uint32_t n_work, void* __restrict__ out, uint32_t* err) {
    // (aligned to a table's size: ChunkTables::lookup ORs the table's address into the index)
    __shared__ __attribute__((aligned(4096))) uint16_t lut_all[kWavesC][1 << ChunkTables::kBits];
    static_assert((2u << ChunkTables::kBits) == 4096u, "table size = its alignment");
    __shared__ __attribute__((aligned(16))) uint32_t stage[kWavesC][kStageWordsC + kStageSlack];
    __shared__ __attribute__((aligned(16))) uint8_t r2s_all[kWavesC][256];
    __shared__ __attribute__((aligned(16))) uint16_t tail_all[kWavesC][kTailEntries];
    const int lane = hf_lane();
    const int wv = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * kWavesC + wv;
    const uint32_t nwaves = gridDim.x * kWavesC;
    const size_t elem_bytes = LAYOUT == HF_LAYOUT_BF16_EXP ? 2 : 1;
    for (uint32_t k = wave; k < n_work; k += nwaves) {
        // decode-on-read: work item k is chunk chunk_ids[k], written to page k of `out`, or to
        // page out_slots[k] when the caller names the pages (a paged buffer of n_slots pages)
        const uint32_t c = chunk_ids ? __builtin_amdgcn_readfirstlane(chunk_ids[k]) : k;
        const uint32_t slot = out_slots ? __builtin_amdgcn_readfirstlane(out_slots[k]) : k;
        if (c >= nchunks || (out_slots && slot >= n_slots)) {
            if (lane == 0) atomicOr(err, HF_KERR_CORRUPT);
            continue;
        }
        const uint32_t off0 = __builtin_amdgcn_readfirstlane(chunk_off[c]);
        uint32_t off1 = __builtin_amdgcn_readfirstlane(chunk_off[c + 1]);
        bool sane = off1 >= off0 && (uint64_t)off1 <= payload_words;
        if (chunk_words) {
            // a slotted stream: the chunk's words are the first chunk_words[c] of its slot
            const uint32_t w = __builtin_amdgcn_readfirstlane(chunk_words[c]);
            sane = sane && w <= off1 - off0;
            off1 = off0 + w;
        }
        const uint64_t first = (uint64_t)c * HF_CHUNK_SYMS;
        const uint64_t left = n - first;
        const uint32_t nvalid = left < HF_CHUNK_SYMS ? (uint32_t)left : HF_CHUNK_SYMS;
        sane = sane && (chunk_ids == nullptr || nvalid == HF_CHUNK_SYMS);
        if (!sane) {
            if (lane == 0) atomicOr(err, HF_KERR_CORRUPT);
            continue;
        }
        ChunkTables T;
        // The first payload window is requested before the tables are built, so that the record's
        // and the window's loads are in flight together; it goes to LDS only afterwards, because
        // the fill descriptors of the build borrow the window's place.
        uint4 w0[kStageWordsC / 256];
        stage_window_load<kStageWordsC>(payload, payload_words, off1, (u64)off0 & ~3ull, lane, w0);
        if (!build_chunk_tables(cbs + c, lut_all[wv], r2s_all[wv], stage[wv], tail_all[wv], T,
                                lane)) {
            if (lane == 0) atomicOr(err, HF_KERR_CORRUPT);
            continue;
        }
        stage_window_store<kStageWordsC>(stage[wv], lane, w0);
        const uint8_t* side_chunk = side ? side + first : nullptr;
        const uint64_t out_first = chunk_ids ? (uint64_t)slot * HF_CHUNK_SYMS : first;
        void* out_chunk = reinterpret_cast<uint8_t*>(out) + out_first * elem_bytes;
        u64 end;
        if (nvalid == HF_CHUNK_SYMS)
            end = decode_chunk_full<LAYOUT>(payload, payload_words, off0, off1, side_chunk,
                                            out_chunk, T, stage[wv], err, lane, true);
        else
            end = decode_chunk_tail<LAYOUT>(payload, payload_words, off0, off1, side_chunk,
                                            out_chunk, nvalid, T, stage[wv], err, lane, true);
        if (end != off1 && lane == 0) atomicOr(err, HF_KERR_CORRUPT);
        // the tables are rebuilt for the next chunk: all reads of this one are done
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

void hf_launch_decode_chunks(const uint32_t* d_payload, uint64_t payload_words,
                             const uint32_t* d_chunk_off, const uint32_t* d_chunk_words,
                             const uint8_t* d_side, uint64_t n,
                             int layout, const hf_chunk_cb* cbs, const uint32_t* d_chunk_ids,
                             const uint32_t* d_out_slots, uint64_t n_slots, uint64_t n_ids,
                             void* d_out, uint32_t* d_err, int num_cu, hipStream_t s) {
    const uint64_t nchunks = (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS;
    const uint64_t n_work = d_chunk_ids ? n_ids : nchunks;
    if (n_work == 0) return;
    // one chunk per wavefront (five 30 KiB workgroups fit one CU's LDS)
    (void)num_cu;
    const uint64_t grid = (n_work + kWavesC - 1) / kWavesC;
    if (layout == HF_LAYOUT_BF16_EXP)
        hipLaunchKernelGGL(hf_k_decode_chunks<HF_LAYOUT_BF16_EXP>, dim3((unsigned)grid),
                           dim3(kWavesC * 64), 0, s, d_payload, payload_words, d_chunk_off,
                           d_chunk_words, d_side, n, (uint32_t)nchunks, cbs, d_chunk_ids, d_out_slots, (uint32_t)n_slots,
                           (uint32_t)n_work, d_out, d_err);
    else
        hipLaunchKernelGGL(hf_k_decode_chunks<HF_LAYOUT_U8>, dim3((unsigned)grid),
                           dim3(kWavesC * 64), 0, s, d_payload, payload_words, d_chunk_off,
                           d_chunk_words, (const uint8_t*)nullptr, n, (uint32_t)nchunks, cbs,
                           d_chunk_ids,
                           d_out_slots, (uint32_t)n_slots, (uint32_t)n_work, d_out, d_err);
}

void hf_launch_decode(const uint32_t* d_payload, uint64_t payload_words,
                      const uint32_t* d_chunk_off, const uint32_t* d_chunk_words,
                      const uint8_t* d_side, uint64_t n, int layout,
                      const hf_cb_dev* cb, void* d_out, uint32_t* d_err, int num_cu,
                      hipStream_t s) {
    const uint64_t nchunks = (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS;
    if (nchunks == 0) return;
    // One chunk per wavefront, workgroups come and go (three 52 KiB workgroups of eight wavefronts
    // fit one CU's LDS): measured 7-10 % faster on bf16 streams than a resident grid looping over
    // the chunks, the same gain a kernel that only moves these bytes in this pattern shows; the
    // price is one table copy out of L2 per eight chunks.
    (void)num_cu;
    const uint64_t grid = (nchunks + kWaves - 1) / kWaves;
    const size_t lds = layout == HF_LAYOUT_BF16_EXP ? sizeof(DecLds<false>) : sizeof(DecLds<true>);
    static_assert(sizeof(DecLds<true>) <= 64 * 1024, "dynamic LDS within the default limit");
    if (layout == HF_LAYOUT_BF16_EXP)
        hipLaunchKernelGGL(hf_k_decode<HF_LAYOUT_BF16_EXP>, dim3((unsigned)grid),
                           dim3(kWaves * 64), lds, s, d_payload, payload_words, d_chunk_off,
                           d_chunk_words, d_side, n, (uint32_t)nchunks, cb, d_out, d_err);
    else
        hipLaunchKernelGGL(hf_k_decode<HF_LAYOUT_U8>, dim3((unsigned)grid), dim3(kWaves * 64), lds,
                           s, d_payload, payload_words, d_chunk_off, d_chunk_words,
                           (const uint8_t*)nullptr, n, (uint32_t)nchunks, cb, d_out, d_err);
}
