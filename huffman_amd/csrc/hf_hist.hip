// SymbolStat (reference: symbol_stat.scala:6-58): 256-bin histogram of a symbol stream.
//
// HBM-bound streaming read.  Every workgroup keeps 32 copies of the 256 counters in LDS
// (32 KiB): lane l adds into copy l % 32, so within a 32-lane group every lane owns its own
// LDS bank and neither bank conflicts nor same-address serialisation occur, however skewed
// the stream is.  One 64-bit global atomic per bin per workgroup merges the result.
#include "hf_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kCopies = 32;

template <int LAYOUT>
__device__ __forceinline__ void hist_vec(const uint4 x, uint32_t* h, const uint32_t copy) {
    const uint32_t w[4] = {x.x, x.y, x.z, x.w};
    if (LAYOUT == HF_LAYOUT_BF16_EXP) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t e0 = (w[i] >> 7) & 0xFFu;   // exponent of the low bf16
            const uint32_t e1 = (w[i] >> 23) & 0xFFu;  // exponent of the high bf16
            atomicAdd(&h[e0 * kCopies + copy], 1u);
            atomicAdd(&h[e1 * kCopies + copy], 1u);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            atomicAdd(&h[((w[i]) & 0xFFu) * kCopies + copy], 1u);
            atomicAdd(&h[((w[i] >> 8) & 0xFFu) * kCopies + copy], 1u);
            atomicAdd(&h[((w[i] >> 16) & 0xFFu) * kCopies + copy], 1u);
            atomicAdd(&h[(w[i] >> 24) * kCopies + copy], 1u);
        }
    }
}

// nvec 16-byte vectors, then `ntail_elems` trailing elements handled by one thread.
template <int LAYOUT>
__global__ __launch_bounds__(kThreads) void hf_k_hist(const uint4* __restrict__ data,
                                                      const uint64_t nvec,
                                                      const uint32_t ntail_elems,
                                                      unsigned long long* __restrict__ freq) {
    __shared__ uint32_t h[256 * kCopies];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 256 * kCopies; i += kThreads) h[i] = 0;
    __syncthreads();

    const uint32_t copy = tid & (kCopies - 1);
    const uint64_t stride = (uint64_t)gridDim.x * kThreads;
    uint64_t v = (uint64_t)blockIdx.x * kThreads + tid;
    // four independent 16-byte loads in flight per lane
    for (; v + 3 * stride < nvec; v += 4 * stride) {
        const uint4 a = data[v];
        const uint4 b = data[v + stride];
        const uint4 c = data[v + 2 * stride];
        const uint4 d = data[v + 3 * stride];
        hist_vec<LAYOUT>(a, h, copy);
        hist_vec<LAYOUT>(b, h, copy);
        hist_vec<LAYOUT>(c, h, copy);
        hist_vec<LAYOUT>(d, h, copy);
    }
    for (; v < nvec; v += stride) hist_vec<LAYOUT>(data[v], h, copy);

    if (blockIdx.x == 0 && tid == 0 && ntail_elems) {
        if (LAYOUT == HF_LAYOUT_BF16_EXP) {
            const uint16_t* t = reinterpret_cast<const uint16_t*>(data + nvec);
            for (uint32_t i = 0; i < ntail_elems; ++i)
                atomicAdd(&h[((t[i] >> 7) & 0xFFu) * kCopies], 1u);
        } else {
            const uint8_t* t = reinterpret_cast<const uint8_t*>(data + nvec);
            for (uint32_t i = 0; i < ntail_elems; ++i) atomicAdd(&h[t[i] * kCopies], 1u);
        }
    }
    __syncthreads();

    // thread t folds the 32 copies of bin t; the rotation keeps the reads conflict free
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < kCopies; ++i) sum += h[tid * kCopies + ((i + tid) & (kCopies - 1))];
    if (sum) atomicAdd(&freq[tid], (unsigned long long)sum);
}

}  // namespace

void hf_launch_hist(const void* d_data, uint64_t n, int layout, unsigned long long* d_freq256,
                    int num_cu, hipStream_t s) {
    if (n == 0) return;
    const uint64_t elems_per_vec = layout == HF_LAYOUT_BF16_EXP ? 8 : 16;
    const uint64_t nvec = n / elems_per_vec;
    const uint32_t ntail = (uint32_t)(n - nvec * elems_per_vec);
    // a workgroup's LDS counters are 32 bit: keep every workgroup below 2^31 symbols
    uint64_t grid = (nvec + kThreads - 1) / kThreads;
    const uint64_t cap = (uint64_t)num_cu * 4;  // 4 workgroups of 32 KiB LDS fit one CU
    if (grid > cap) grid = cap;
    if (grid == 0) grid = 1;
    while (n / grid >= (1ull << 31)) grid *= 2;
    const uint4* p = reinterpret_cast<const uint4*>(d_data);
    if (layout == HF_LAYOUT_BF16_EXP)
        hipLaunchKernelGGL(hf_k_hist<HF_LAYOUT_BF16_EXP>, dim3((unsigned)grid), dim3(kThreads), 0,
                           s, p, nvec, ntail, d_freq256);
    else
        hipLaunchKernelGGL(hf_k_hist<HF_LAYOUT_U8>, dim3((unsigned)grid), dim3(kThreads), 0, s, p,
                           nvec, ntail, d_freq256);
}
