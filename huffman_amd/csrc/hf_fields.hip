// Sub-field front ends: an element is split into its exponent field, which becomes the 8-bit
// symbol the coding kernels work on, and a side field of the remaining bits, kept raw.
// (reports/spec.md:22 of the reference lists FP16 / FP8 / FP4 operands for the compressor; its RTL
// stops at 8-bit symbols, symbol_stat.scala:6-13, and bitplane/bitplane.scala is an empty file.)
//
//   HF_LAYOUT_FP16_EXP      s eeeee mmmmmmmmmm   symbol = eeeee, side field = s | m (11 bits)
//   HF_LAYOUT_FP8_E4M3_EXP  s eeee mmm           symbol = s eeee, side field = mmm (3 bits)
//
// The side stream is the dense little-endian bit-packing of the side fields in element order, so
// a group of 8 elements fills exactly 11 or 3 bytes.  One thread per group; a workgroup's side
// bytes pass through LDS so that they leave (and arrive) as whole dwords, coalesced.
// HBM traffic per element: split reads 2 (1) B and writes 1 + 11/8 (1 + 3/8) B; merge the reverse.
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

}  // namespace

uint64_t hf_fields_side_bytes_impl(uint64_t n, int layout) {
    const uint64_t groups = (n + 7) / 8;
    if (layout == HF_LAYOUT_FP16_EXP) return groups * 11;
    if (layout == HF_LAYOUT_FP8_E4M3_EXP) return groups * 3;
    return 0;
}

void hf_launch_fields_split(const void* d_data, uint64_t n, int layout, uint8_t* d_sym,
                            uint8_t* d_side, hipStream_t s) {
    if (n == 0) return;
    const unsigned grid = (unsigned)(((n + 7) / 8 + kThreads - 1) / kThreads);
    if (layout == HF_LAYOUT_FP16_EXP)
        hipLaunchKernelGGL(hf_k_fields_split<HF_LAYOUT_FP16_EXP>, dim3(grid), dim3(kThreads), 0, s,
                           d_data, n, d_sym, d_side);
    else
        hipLaunchKernelGGL(hf_k_fields_split<HF_LAYOUT_FP8_E4M3_EXP>, dim3(grid), dim3(kThreads), 0,
                           s, d_data, n, d_sym, d_side);
}

void hf_launch_fields_merge(const uint8_t* d_sym, const uint8_t* d_side, uint64_t n, int layout,
                            void* d_out, hipStream_t s) {
    if (n == 0) return;
    const unsigned grid = (unsigned)(((n + 7) / 8 + kThreads - 1) / kThreads);
    if (layout == HF_LAYOUT_FP16_EXP)
        hipLaunchKernelGGL(hf_k_fields_merge<HF_LAYOUT_FP16_EXP>, dim3(grid), dim3(kThreads), 0, s,
                           d_sym, d_side, n, d_out);
    else
        hipLaunchKernelGGL(hf_k_fields_merge<HF_LAYOUT_FP8_E4M3_EXP>, dim3(grid), dim3(kThreads), 0,
                           s, d_sym, d_side, n, d_out);
}
