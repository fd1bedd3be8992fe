// Device-side pieces shared by the decoders of the packed stream (hf_decode.hip) and of the paged
// stream (hf_pages.hip): table reads by LDS byte address, the per-block table view and the symbol
// steps that walk a lane's bit window.  Included inside each translation unit's own unnamed
// namespace.
#pragma once
#include "hf_common.h"

namespace {

typedef unsigned long long u64;

// a + (low byte of b), and "is the low byte of x zero" as a lane mask: one instruction each
// (byte operand selection), where a mask and an add / a compare would take two
__device__ __forceinline__ int add_byte0(const int a, const uint32_t b) {
    int r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
        : "=v"(r)
        : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ u64 byte0_is_zero(const uint32_t x) {
    u64 m;
    asm("v_cmp_eq_u32_sdwa %0, %1, 0 src0_sel:BYTE_0 src1_sel:DWORD" : "=s"(m) : "v"(x));
    return m;
}

// Table reads by LDS byte address.  Through a pointer the compiler adds the table's base to the
// index in a vector instruction of its own, even when that base is a constant; with the address
// formed here a constant base lands in the instruction's offset field and a wave-uniform one is
// OR-ed in by the instruction that scales the index.
typedef __attribute__((address_space(3))) const uint16_t hf_lds_u16;
typedef __attribute__((address_space(3))) const uint32_t hf_lds_cu32;
typedef __attribute__((address_space(3))) const uint8_t hf_lds_u8;
__device__ __forceinline__ uint32_t lds_u16(const uint32_t byte_addr) {
    return *reinterpret_cast<hf_lds_u16*>((uintptr_t)byte_addr);
}
__device__ __forceinline__ uint32_t lds_u32(const uint32_t byte_addr) {
    return *reinterpret_cast<hf_lds_cu32*>((uintptr_t)byte_addr);
}
__device__ __forceinline__ uint32_t lds_u8(const uint32_t byte_addr) {
    return *reinterpret_cast<hf_lds_u8*>((uintptr_t)byte_addr);
}
__device__ __forceinline__ uint32_t lds_address(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// Tables of one chunk's codebook, rebuilt by the decoding wavefront from the canonical order:
// a private first-level table of kBits bits (11, half the stream-wide one: the tables are per
// wavefront, and LDS per wavefront is what bounds how many chunks a CU decodes at once) plus,
// for the longer codes, either a direct table over their part of the 16-bit code space or, when
// that part is too large, the start/end of each length's block and the rank of its first code.
template <int BITS, int STAGE, bool PAIRCHECK = false>
struct ChunkTablesT {
    static constexpr int kBits = BITS;
    static constexpr int kStage = STAGE;
    static constexpr bool kPairs = false;
    // the escape test per symbol: an 11-bit first level meets long codes often enough that the
    // lookup a per-pair test wastes on them costs more than the test saves (0.64 against 0.66 ms
    // per 65536 fp8 pages)
    static constexpr bool kPairCheck = PAIRCHECK;
    static constexpr int kLong = 16 - kBits;  // code lengths kBits+1 .. 16
    uint32_t lut_addr;       // LDS address of this wavefront's 1 << kBits entries, a multiple of
                             // the table's size (wave-uniform)
    const uint8_t* rank2sym; // LDS, 256 entries
    const uint16_t* tail;    // LDS, direct table over the code space of the long codes
    uint32_t tail_n;         // entries of `tail` in use; 0: that space is too large, use the blocks
    uint32_t start[kLong], end[kLong], off[kLong];  // per long length (wave-uniform)

    __device__ __forceinline__ uint32_t lookup(const uint32_t top) const {
        // index, scaled to bytes, with the table's address OR-ed in: two instructions
        return lds_u16(((top >> (32 - kBits)) << 1) | lut_addr);
    }
    __device__ __forceinline__ uint32_t long_code(const uint32_t top, const uint32_t,
                                                  uint32_t& len) const {
        const uint32_t v16 = top >> 16;
        if (tail_n != 0) {  // wave-uniform: one lookup, as with the stream-wide tables
            const uint32_t idx = v16 - start[0];
            const uint32_t e2 = idx < tail_n ? tail[idx] : 0u;
            len = e2 & 0xFFu;
            return e2 >> 8;
        }
        uint32_t p = 0;
        len = 0;
#pragma unroll
        for (int k = kLong - 1; k >= 0; --k) {  // the lowest matching block wins
            if (v16 < end[k]) {
                len = (uint32_t)(kBits + 1 + k);
                p = off[k] + ((v16 - start[k]) >> (kLong - 1 - k));
            }
        }
        return rank2sym[p & 255u];
    }
};

// The bit window of a lane: `win` holds the buffered bits MSB-aligned, `room` = 32 - (number
// of buffered bits); a lane asks for a word when room > 0.
//
// Decodes one symbol and returns its table entry (symbol << 8) | length; the callers pick the
// symbol byte out with v_perm_b32 when they pack the output.
template <class TAB>
__device__ __forceinline__ uint32_t decode_one(u64& win, int& room, const TAB& T, uint32_t* err) {
    const uint32_t top = (uint32_t)(win >> 32);
    uint32_t e = T.lookup(top);
    if (__builtin_expect((e & 0xFFu) == 0, 0)) {
        uint32_t len;
        const uint32_t sym = T.long_code(top, e >> 8, len);
        if (len == 0) {  // no code matches: flag it and keep moving
            atomicOr(err, HF_KERR_CORRUPT);
            len = 1;
        }
        e = (sym << 8) | len;
    }
    win <<= (e & 63u);  // lengths are <= 16; the 64-bit shift reads only these six bits
    room += (int)(e & 0xFFu);
    return e;
}

// Both symbols of a pair, one lookup each, the escape test once for the two: a lane whose first
// lookup met an escape entry (length 0) has not moved, so its second lookup returns the same
// entry -- "the second length is zero" covers both.  Entries e0, e1: symbol in byte 1.
template <class TAB>
__device__ __forceinline__ void decode_pair_single(u64& win, int& room, const TAB& T, uint32_t* err,
                                                   uint32_t& e0, uint32_t& e1) {
    e0 = T.lookup((uint32_t)(win >> 32));
    win <<= (e0 & 63u);
    e1 = T.lookup((uint32_t)(win >> 32));
    if (__builtin_expect(byte0_is_zero(e1) != 0ull, 0)) {  // wave-uniform
        if ((e1 & 0xFFu) == 0) {
            if ((e0 & 0xFFu) == 0) {
                // the first code is the long one: nothing was consumed yet
                const uint32_t top = (uint32_t)(win >> 32);
                uint32_t len;
                const uint32_t sym = T.long_code(top, e0 >> 8, len);
                if (len == 0) {
                    atomicOr(err, HF_KERR_CORRUPT);
                    len = 1;
                }
                e0 = (sym << 8) | len;
                win <<= len;
                e1 = T.lookup((uint32_t)(win >> 32));
            }
            if ((e1 & 0xFFu) == 0) {
                const uint32_t top = (uint32_t)(win >> 32);
                uint32_t len;
                const uint32_t sym = T.long_code(top, e1 >> 8, len);
                if (len == 0) {
                    atomicOr(err, HF_KERR_CORRUPT);
                    len = 1;
                }
                e1 = (sym << 8) | len;
            }
        }
    }
    win <<= (e1 & 63u);
    room = add_byte0(room, e0 + e1);  // both lengths are <= 16: the byte sum does not carry
}

}  // namespace
