// Sub-field front ends: an element is split into its exponent field, which becomes the 8-bit
// symbol the coding kernels work on, and a side field of the remaining bits, kept raw.
// (reports/spec.md:22 of the reference lists FP16 / FP8 / FP4 operands for the compressor; its RTL
// stops at 8-bit symbols, symbol_stat.scala:6-13, and bitplane/bitplane.scala is an empty file.)
//
//   HF_LAYOUT_FP16_EXP      s eeeee mmmmmmmmmm   symbol = eeeee, side field = s | m (11 bits)
//   HF_LAYOUT_FP8_E4M3_EXP  s eeee mmm           symbol = s eeee, side field = mmm (3 bits)
//   HF_LAYOUT_FP4_E2M1_EXP  s ee m | s ee m      an element is a byte of two fp4 values (the first in
//                                                the low nibble): symbol = see | see << 3, side
//                                                field = m | m << 1 (2 bits)
//
// The side stream is the dense little-endian bit-packing of the side fields in element order, so
// a group of 8 elements fills exactly 11, 3 or 2 bytes.  One thread per group; a workgroup's side
// bytes pass through LDS so that they leave (and arrive) as whole dwords, coalesced.
// HBM traffic per element: split reads 2 (1) B and writes 1 + 11/8 (1 + 3/8, 1 + 2/8) B; merge
// the reverse.
//
// HF_LAYOUT_BITPLANE8 is a transform of another kind (the reference's bitplane/bitplane.scala is
// an empty file; reports/spec.md:22 names bit-plane compression): the bytes are transposed into
// bit planes, blocks of HF_BITPLANE_BLOCK bytes at a time, and there is no side stream.  A block
// of m bytes is G = m / 8 groups of 8 consecutive bytes; symbol block + p * G + g holds bit p of
// each byte of group g, byte i of the group in bit i.  With full blocks chunk 8k + p of the symbol
// stream is bit plane p of data chunks 8k .. 8k + 7, so per-chunk codebooks see one plane each.
// 1 B read and 1 B written per byte, either way.
#include "hf_common.h"

namespace {

constexpr int kThreads = 256;

template <int LAYOUT>
struct Fields;
template <>
struct Fields<HF_LAYOUT_FP16_EXP> {
    static constexpr int kBytes = 11;  // side bytes per group of 8 elements
    typedef uint16_t elem_t;
    static __device__ __forceinline__ uint32_t symbol(uint32_t w) { return (w >> 10) & 0x1Fu; }
    static __device__ __forceinline__ uint32_t side(uint32_t w) { return ((w >> 15) << 10) | (w & 0x3FFu); }
    static __device__ __forceinline__ uint32_t join(uint32_t s, uint32_t f) {
        return ((f >> 10) << 15) | ((s & 0x1Fu) << 10) | (f & 0x3FFu);
    }
};
template <>
struct Fields<HF_LAYOUT_FP8_E4M3_EXP> {
    static constexpr int kBytes = 3;
    typedef uint8_t elem_t;
    static __device__ __forceinline__ uint32_t symbol(uint32_t w) { return (w >> 3) & 0x1Fu; }
    static __device__ __forceinline__ uint32_t side(uint32_t w) { return w & 7u; }
    static __device__ __forceinline__ uint32_t join(uint32_t s, uint32_t f) {
        return ((s & 0x1Fu) << 3) | (f & 7u);
    }
};

template <>
struct Fields<HF_LAYOUT_FP4_E2M1_EXP> {
    static constexpr int kBytes = 2;
    typedef uint8_t elem_t;
    static __device__ __forceinline__ uint32_t symbol(uint32_t w) {
        return ((w >> 1) & 7u) | (((w >> 5) & 7u) << 3);
    }
    static __device__ __forceinline__ uint32_t side(uint32_t w) { return (w & 1u) | (((w >> 4) & 1u) << 1); }
    static __device__ __forceinline__ uint32_t join(uint32_t s, uint32_t f) {
        return (f & 1u) | ((s & 7u) << 1) | (((f >> 1) & 1u) << 4) | (((s >> 3) & 7u) << 5);
    }
};

template <int LAYOUT>
__global__ __launch_bounds__(kThreads) void hf_k_fields_split(const void* __restrict__ data,
                                                              const uint64_t n,
                                                              uint8_t* __restrict__ sym,
                                                              uint8_t* __restrict__ side) {
    typedef Fields<LAYOUT> F;
    typedef typename F::elem_t elem_t;
    constexpr int B = F::kBytes;
    __shared__ __attribute__((aligned(16))) uint8_t stage[kThreads * B];
    const uint64_t ngroups = (n + 7) / 8;
    const uint64_t g0 = (uint64_t)blockIdx.x * kThreads;
    const uint64_t g = g0 + threadIdx.x;
    if (g < ngroups) {
        const elem_t* in = reinterpret_cast<const elem_t*>(data) + g * 8;
        uint32_t w[8];
        if (g * 8 + 8 <= n) {
            if (sizeof(elem_t) == 2) {
                const uint4 v = *reinterpret_cast<const uint4*>(in);
                const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) w[2 * i] = d[i] & 0xFFFFu, w[2 * i + 1] = d[i] >> 16;
            } else {
                const uint2 v = *reinterpret_cast<const uint2*>(in);
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = (v.x >> (8 * i)) & 0xFFu, w[4 + i] = (v.y >> (8 * i)) & 0xFFu;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = g * 8 + i < n ? (uint32_t)in[i] : 0u;
        }
        // the 8 side fields, packed LSB first, as up to three dwords
        uint32_t acc[3] = {0u, 0u, 0u};
        uint32_t s_lo = 0, s_hi = 0;
        constexpr int FB = B;  // bits per field = bytes per group
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t f = F::side(w[i]);
            const int at = i * FB;
            acc[at >> 5] |= f << (at & 31);
            if ((at & 31) + FB > 32) acc[(at >> 5) + 1] |= f >> (32 - (at & 31));
            const uint32_t sy = F::symbol(w[i]);
            if (i < 4) s_lo |= sy << (8 * i); else s_hi |= sy << (8 * (i - 4));
        }
#pragma unroll
        for (int b = 0; b < B; ++b) stage[threadIdx.x * B + b] = (uint8_t)(acc[b >> 2] >> (8 * (b & 3)));
        if (g * 8 + 8 <= n) {
            *reinterpret_cast<uint2*>(sym + g * 8) = make_uint2(s_lo, s_hi);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (g * 8 + i < n) sym[g * 8 + i] = (uint8_t)((i < 4 ? s_lo : s_hi) >> (8 * (i & 3)));
        }
    }
    __syncthreads();
    // this workgroup's side bytes: whole dwords first, then the last 1..3 bytes of the stream
    const uint64_t here = ngroups - g0 < (uint64_t)kThreads ? ngroups - g0 : (uint64_t)kThreads;
    const uint32_t nbytes = (uint32_t)here * B;
    uint8_t* out = side + g0 * B;  // (256 * B bytes per workgroup: a multiple of 16)
    for (uint32_t d = threadIdx.x; d < nbytes / 4; d += kThreads)
        reinterpret_cast<uint32_t*>(out)[d] = reinterpret_cast<const uint32_t*>(stage)[d];
    if (threadIdx.x < (nbytes & 3u)) out[(nbytes & ~3u) + threadIdx.x] = stage[(nbytes & ~3u) + threadIdx.x];
}

template <int LAYOUT>
__global__ __launch_bounds__(kThreads) void hf_k_fields_merge(const uint8_t* __restrict__ sym,
                                                              const uint8_t* __restrict__ side,
                                                              const uint64_t n,
                                                              void* __restrict__ data) {
    typedef Fields<LAYOUT> F;
    typedef typename F::elem_t elem_t;
    constexpr int B = F::kBytes;
    __shared__ __attribute__((aligned(16))) uint8_t stage[kThreads * B + 4];
    const uint64_t ngroups = (n + 7) / 8;
    const uint64_t g0 = (uint64_t)blockIdx.x * kThreads;
    const uint64_t g = g0 + threadIdx.x;
    const uint64_t here = ngroups - g0 < (uint64_t)kThreads ? ngroups - g0 : (uint64_t)kThreads;
    const uint32_t nbytes = (uint32_t)here * B;
    const uint8_t* in_side = side + g0 * B;
    for (uint32_t d = threadIdx.x; d < nbytes / 4; d += kThreads)
        reinterpret_cast<uint32_t*>(stage)[d] = reinterpret_cast<const uint32_t*>(in_side)[d];
    if (threadIdx.x < (nbytes & 3u)) stage[(nbytes & ~3u) + threadIdx.x] = in_side[(nbytes & ~3u) + threadIdx.x];
    __syncthreads();
    if (g >= ngroups) return;
    uint32_t acc[3] = {0u, 0u, 0u};
#pragma unroll
    for (int b = 0; b < B; ++b) acc[b >> 2] |= (uint32_t)stage[threadIdx.x * B + b] << (8 * (b & 3));
    uint32_t s[8];
    if (g * 8 + 8 <= n) {
        const uint2 v = *reinterpret_cast<const uint2*>(sym + g * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = (v.x >> (8 * i)) & 0xFFu, s[4 + i] = (v.y >> (8 * i)) & 0xFFu;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = g * 8 + i < n ? (uint32_t)sym[g * 8 + i] : 0u;
    }
    uint32_t w[8];
    constexpr int FB = B;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int at = i * FB;
        uint32_t f = acc[at >> 5] >> (at & 31);
        if ((at & 31) + FB > 32) f |= acc[(at >> 5) + 1] << (32 - (at & 31));
        w[i] = F::join(s[i], f & ((1u << FB) - 1u));
    }
    elem_t* out = reinterpret_cast<elem_t*>(data) + g * 8;
    if (g * 8 + 8 <= n) {
        if (sizeof(elem_t) == 2)
            *reinterpret_cast<uint4*>(out) = make_uint4(w[0] | (w[1] << 16), w[2] | (w[3] << 16),
                                                        w[4] | (w[5] << 16), w[6] | (w[7] << 16));
        else
            *reinterpret_cast<uint2*>(out) =
                make_uint2(w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24),
                           w[4] | (w[5] << 8) | (w[6] << 16) | (w[7] << 24));
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (g * 8 + i < n) out[i] = (elem_t)w[i];
    }
}

// ---- bit planes ------------------------------------------------------------------------------
typedef unsigned long long u64;
constexpr uint32_t kPlaneGroups = HF_BITPLANE_BLOCK / 8;  // groups of a full block
constexpr uint32_t kPlaneQuads = kPlaneGroups / 4;        // threads of a full block

// 8 x 8 bit matrix transpose: bit i of byte p of the result = bit p of byte i of x
__device__ __forceinline__ u64 transpose8(u64 x) {
    x = (x & 0xAA55AA55AA55AA55ull) | ((x & 0x00AA00AA00AA00AAull) << 7) | ((x >> 7) & 0x00AA00AA00AA00AAull);
    x = (x & 0xCCCC3333CCCC3333ull) | ((x & 0x0000CCCC0000CCCCull) << 14) | ((x >> 14) & 0x0000CCCC0000CCCCull);
    x = (x & 0xF0F0F0F00F0F0F0Full) | ((x & 0x00000000F0F0F0F0ull) << 28) | ((x >> 28) & 0x00000000F0F0F0F0ull);
    return x;
}
// byte k of each of a, b, c, d, in that order (a 4 x 4 byte transpose takes four of these)
__device__ __forceinline__ uint32_t gather_byte(const uint32_t a, const uint32_t b, const uint32_t c,
                                                const uint32_t d, const uint32_t k) {
    const uint32_t sel = k | ((4u + k) << 8) | 0x0C0C0000u;  // (selector 0x0C reads as zero)
    const uint32_t ab = __builtin_amdgcn_perm(b, a, sel);
    const uint32_t cd = __builtin_amdgcn_perm(d, c, sel);
    return ab | (cd << 16);
}

// One thread moves four groups: 32 consecutive data bytes on one side, four bytes of each of the
// eight planes on the other.  INVERSE = false: data -> planes; true: planes -> data (the bit
// transpose is its own inverse, so the two directions differ in addressing only).
template <bool INVERSE>
__global__ __launch_bounds__(kThreads) void hf_k_bitplane(const uint8_t* __restrict__ in,
                                                          const uint64_t n,
                                                          uint8_t* __restrict__ out) {
    const uint64_t q = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    const uint64_t b = q / kPlaneQuads;               // block
    const uint32_t g0 = (uint32_t)(q % kPlaneQuads) * 4u;  // first group of this thread
    const uint64_t b0 = b * HF_BITPLANE_BLOCK;
    if (b0 >= n) return;
    const uint64_t m = n - b0 < HF_BITPLANE_BLOCK ? n - b0 : (uint64_t)HF_BITPLANE_BLOCK;
    const uint32_t G = (uint32_t)(m / 8);
    if (g0 >= G) return;
    const uint8_t* data_in = in + b0 + 8ull * g0;     // (INVERSE: unused)
    const uint8_t* planes_in = in + b0 + g0;          // plane p at + p * G
    uint8_t* data_out = out + b0 + 8ull * g0;
    uint8_t* planes_out = out + b0 + g0;
    const bool whole = g0 + 4u <= G && (G & 3u) == 0;  // four groups, plane addresses 4-byte aligned
    u64 t[4] = {0, 0, 0, 0};
    if (!INVERSE) {
        if (whole) {
            const uint4 v0 = reinterpret_cast<const uint4*>(data_in)[0];
            const uint4 v1 = reinterpret_cast<const uint4*>(data_in)[1];
            t[0] = (u64)v0.x | ((u64)v0.y << 32);
            t[1] = (u64)v0.z | ((u64)v0.w << 32);
            t[2] = (u64)v1.x | ((u64)v1.y << 32);
            t[3] = (u64)v1.z | ((u64)v1.w << 32);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (g0 + j < G) {
                    const uint2 v = reinterpret_cast<const uint2*>(data_in)[j];
                    t[j] = (u64)v.x | ((u64)v.y << 32);
                }
        }
    } else {
        uint32_t P[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint8_t* src = planes_in + (uint64_t)p * G;
            if (whole) {
                P[p] = *reinterpret_cast<const uint32_t*>(src);
            } else {
                P[p] = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (g0 + j < G) P[p] |= (uint32_t)src[j] << (8 * j);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            t[j] = (u64)gather_byte(P[0], P[1], P[2], P[3], j) |
                   ((u64)gather_byte(P[4], P[5], P[6], P[7], j) << 32);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = transpose8(t[j]);
    if (!INVERSE) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint32_t k = p & 3;
            const uint32_t w =
                p < 4 ? gather_byte((uint32_t)t[0], (uint32_t)t[1], (uint32_t)t[2], (uint32_t)t[3], k)
                      : gather_byte((uint32_t)(t[0] >> 32), (uint32_t)(t[1] >> 32),
                                    (uint32_t)(t[2] >> 32), (uint32_t)(t[3] >> 32), k);
            uint8_t* dst = planes_out + (uint64_t)p * G;
            if (whole) {
                *reinterpret_cast<uint32_t*>(dst) = w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (g0 + j < G) dst[j] = (uint8_t)(w >> (8 * j));
            }
        }
    } else {
        if (whole) {
            reinterpret_cast<uint4*>(data_out)[0] = make_uint4((uint32_t)t[0], (uint32_t)(t[0] >> 32),
                                                               (uint32_t)t[1], (uint32_t)(t[1] >> 32));
            reinterpret_cast<uint4*>(data_out)[1] = make_uint4((uint32_t)t[2], (uint32_t)(t[2] >> 32),
                                                               (uint32_t)t[3], (uint32_t)(t[3] >> 32));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (g0 + j < G)
                    reinterpret_cast<uint2*>(data_out)[j] =
                        make_uint2((uint32_t)t[j], (uint32_t)(t[j] >> 32));
        }
    }
}

}  // namespace

uint64_t hf_fields_side_bytes_impl(uint64_t n, int layout) {
    const uint64_t groups = (n + 7) / 8;
    if (layout == HF_LAYOUT_FP16_EXP) return groups * 11;
    if (layout == HF_LAYOUT_FP8_E4M3_EXP) return groups * 3;
    if (layout == HF_LAYOUT_FP4_E2M1_EXP) return groups * 2;
    return 0;
}

static unsigned bitplane_grid(uint64_t n) {
    const uint64_t full = n / HF_BITPLANE_BLOCK;
    const uint64_t rest_groups = (n % HF_BITPLANE_BLOCK) / 8;
    const uint64_t quads = full * kPlaneQuads + (rest_groups ? kPlaneQuads : 0);  // (whole blocks of threads)
    return (unsigned)((quads + kThreads - 1) / kThreads);
}

void hf_launch_fields_split(const void* d_data, uint64_t n, int layout, uint8_t* d_sym,
                            uint8_t* d_side, hipStream_t s) {
    if (n == 0) return;
    if (layout == HF_LAYOUT_BITPLANE8) {
        hipLaunchKernelGGL(hf_k_bitplane<false>, dim3(bitplane_grid(n)), dim3(kThreads), 0, s,
                           reinterpret_cast<const uint8_t*>(d_data), n, d_sym);
        return;
    }
    const unsigned grid = (unsigned)(((n + 7) / 8 + kThreads - 1) / kThreads);
    if (layout == HF_LAYOUT_FP16_EXP)
        hipLaunchKernelGGL(hf_k_fields_split<HF_LAYOUT_FP16_EXP>, dim3(grid), dim3(kThreads), 0, s,
                           d_data, n, d_sym, d_side);
    else if (layout == HF_LAYOUT_FP8_E4M3_EXP)
        hipLaunchKernelGGL(hf_k_fields_split<HF_LAYOUT_FP8_E4M3_EXP>, dim3(grid), dim3(kThreads), 0,
                           s, d_data, n, d_sym, d_side);
    else
        hipLaunchKernelGGL(hf_k_fields_split<HF_LAYOUT_FP4_E2M1_EXP>, dim3(grid), dim3(kThreads), 0,
                           s, d_data, n, d_sym, d_side);
}

void hf_launch_fields_merge(const uint8_t* d_sym, const uint8_t* d_side, uint64_t n, int layout,
                            void* d_out, hipStream_t s) {
    if (n == 0) return;
    if (layout == HF_LAYOUT_BITPLANE8) {
        hipLaunchKernelGGL(hf_k_bitplane<true>, dim3(bitplane_grid(n)), dim3(kThreads), 0, s, d_sym,
                           n, reinterpret_cast<uint8_t*>(d_out));
        return;
    }
    const unsigned grid = (unsigned)(((n + 7) / 8 + kThreads - 1) / kThreads);
    if (layout == HF_LAYOUT_FP16_EXP)
        hipLaunchKernelGGL(hf_k_fields_merge<HF_LAYOUT_FP16_EXP>, dim3(grid), dim3(kThreads), 0, s,
                           d_sym, d_side, n, d_out);
    else if (layout == HF_LAYOUT_FP8_E4M3_EXP)
        hipLaunchKernelGGL(hf_k_fields_merge<HF_LAYOUT_FP8_E4M3_EXP>, dim3(grid), dim3(kThreads), 0,
                           s, d_sym, d_side, n, d_out);
    else
        hipLaunchKernelGGL(hf_k_fields_merge<HF_LAYOUT_FP4_E2M1_EXP>, dim3(grid), dim3(kThreads), 0,
                           s, d_sym, d_side, n, d_out);
}
