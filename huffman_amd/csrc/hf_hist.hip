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
constexpr int kInFlight = 4;

// The input is read once: a non-temporal load keeps it from displacing anything in the caches
// and streams measurably faster (7.0 against 6.2 TB/s for a plain read of this box).
__device__ __forceinline__ uint4 stream_load(const uint4* p) {
    const hf_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const hf_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

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
    // a workgroup takes tiles of kInFlight * 256 consecutive vectors: kInFlight independent
    // 16-byte loads in flight per lane
    constexpr uint64_t kTile = (uint64_t)kInFlight * kThreads;
    const uint64_t ntiles = nvec / kTile;
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint4* src = data + t * kTile + tid;
        uint4 x[kInFlight];
#pragma unroll
        for (int k = 0; k < kInFlight; ++k) x[k] = stream_load(src + k * kThreads);
#pragma unroll
        for (int k = 0; k < kInFlight; ++k) hist_vec<LAYOUT>(x[k], h, copy);
    }
    for (uint64_t v = ntiles * kTile + (uint64_t)blockIdx.x * kThreads + tid; v < nvec;
         v += (uint64_t)gridDim.x * kThreads)
        hist_vec<LAYOUT>(stream_load(data + v), h, copy);

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

// One histogram per 16384-symbol chunk (a KV-cache page): one wavefront per chunk.  A chunk holds
// at most 16384 of any symbol, so a counter fits 16 bits and two bins share a 32-bit LDS word
// (bins 2k and 2k+1 in the halves of word k); that halves the table and pays for 16 copies of it
// in the 8 KiB that 8 copies of 32-bit counters took, lane l adding into copy l mod 16.  Measured
// on 65536 fp8 pages: 16 packed copies 0.198 ms, 8 unpacked 0.24, 8 packed 0.246, 32 packed (a
// bank per lane, but 16 KiB per wavefront and 8 wavefronts per CU instead of 20) 0.261.
// Folded and written as uint32[256].
constexpr int kChunkCopies = 16;
constexpr int kChunkWaves = 4;
constexpr int kChunkWords = 128 * kChunkCopies;  // 8 KiB per wavefront

template <int LAYOUT>
__global__ __launch_bounds__(kChunkWaves * 64) void hf_k_hist_chunks(
    const void* __restrict__ data, const uint64_t n, const uint32_t nchunks,
    uint32_t* __restrict__ freq_chunks) {
    __shared__ __attribute__((aligned(16))) uint32_t h_all[kChunkWaves][kChunkWords];
    const int lane = hf_lane();
    const int wv = threadIdx.x >> 6;
    uint32_t* h = h_all[wv];
    const uint32_t c = blockIdx.x * kChunkWaves + wv;
    if (c >= nchunks) return;
#pragma unroll
    for (int k = 0; k < kChunkWords / 4 / 64; ++k)
        reinterpret_cast<uint4*>(h)[k * 64 + lane] = make_uint4(0, 0, 0, 0);
    const uint64_t first = (uint64_t)c * HF_CHUNK_SYMS;
    const uint64_t left = n - first;
    const uint32_t nvalid = left < HF_CHUNK_SYMS ? (uint32_t)left : HF_CHUNK_SYMS;
    const uint32_t copy = lane & (kChunkCopies - 1);
    auto count = [&](const uint32_t sym) {
        atomicAdd(&h[(sym >> 1) * kChunkCopies + copy], 1u << ((sym & 1u) * 16u));
    };
    const uint4* const pv = reinterpret_cast<const uint4*>(
        reinterpret_cast<const uint8_t*>(data) + first * (LAYOUT == HF_LAYOUT_BF16_EXP ? 2 : 1));
    constexpr uint32_t kPerVec = LAYOUT == HF_LAYOUT_BF16_EXP ? 8 : 16;
    auto count_vec = [&](const uint4 x) {
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (LAYOUT == HF_LAYOUT_BF16_EXP) {
                count((w[i] >> 7) & 0xFFu);
                count((w[i] >> 23) & 0xFFu);
            } else {
                count(w[i] & 0xFFu);
                count((w[i] >> 8) & 0xFFu);
                count((w[i] >> 16) & 0xFFu);
                count(w[i] >> 24);
            }
        }
    };
    // the wavefront's own LDS operations complete in order; no other wavefront touches h
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (nvalid == HF_CHUNK_SYMS) {
        // a whole page: its loads are issued 16 at a time before any of them is consumed
        constexpr int kBatch = 16;
        constexpr int kBatches = HF_CHUNK_SYMS / kPerVec / 64 / kBatch;
#pragma unroll 1
        for (int b = 0; b < kBatches; ++b) {
            uint4 x[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) x[k] = stream_load(pv + (b * kBatch + k) * 64 + lane);
#pragma unroll
            for (int k = 0; k < kBatch; ++k) count_vec(x[k]);
        }
    } else {
        const uint32_t nvec = nvalid / kPerVec;
        for (uint32_t v = lane; v < nvec; v += 64) count_vec(pv[v]);
        for (uint32_t i = nvec * kPerVec + lane; i < nvalid; i += 64) {
            if (LAYOUT == HF_LAYOUT_BF16_EXP)
                count((reinterpret_cast<const uint16_t*>(pv)[i] >> 7) & 0xFFu);
            else
                count(reinterpret_cast<const uint8_t*>(pv)[i]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lane l folds the bin pairs l and l + 64 (bins 2l, 2l+1 and 128+2l, 129+2l); the rotation
    // keeps the reads of a 32-lane group on 32 different banks.  No half can carry into the
    // other: a chunk holds at most 16384 symbols.
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int pair = lane + 64 * k;
        uint32_t sum = 0;
#pragma unroll
        for (int i = 0; i < kChunkCopies; ++i)
            sum += h[pair * kChunkCopies + ((i + lane) & (kChunkCopies - 1))];
        reinterpret_cast<uint2*>(freq_chunks + (size_t)c * 256)[pair] =
            make_uint2(sum & 0xFFFFu, sum >> 16);
    }
}

// The same counts for a stream that also wants its own 256 (one codebook for the stream, pages
// sized from their counts): per-page counts AND the stream's totals from one read of the data.
// A page is counted by a WORKGROUP: its four wavefronts add into one table of 32 copies of the
// packed counters (a bank per lane of a 32-lane group: no conflict, no same-address serialisation,
// however skewed the page), word k of a copy holding bins 2k and 2k+1 as
// (count of both) | (count of 2k+1) << 16 -- the increment is then 1 | odd << 16, one instruction.
// 16 KiB of LDS per workgroup leaves room for as many workgroups as the registers allow.
// Workgroups stay resident and walk the pages, asking for their next page before they count the
// one they hold; thread t < 128 folds word t of all copies, keeps the running totals of its two
// bins in registers, and adds them to the stream's counts once, at the end.
constexpr int kPageCopies = 32;
constexpr int kPageWords = 128 * kPageCopies;

typedef __attribute__((address_space(3))) uint32_t hf_lds_w32;

template <int LAYOUT>
__global__ __launch_bounds__(256) void hf_k_hist_pages(
    const void* __restrict__ data, const uint64_t n, const uint32_t nchunks,
    uint32_t* __restrict__ freq_chunks, unsigned long long* __restrict__ freq256) {
    __shared__ __attribute__((aligned(16))) uint32_t h[kPageWords];
    const uint32_t tid = threadIdx.x;
    // byte address of this lane's column of the table: word k of copy (lane mod 32) at + 128 k
    const uint32_t col = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)h +
                         (tid & (kPageCopies - 1)) * 4u;
    auto add = [&](const uint32_t pair, const uint32_t inc) {
        __hip_atomic_fetch_add(reinterpret_cast<hf_lds_w32*>((uintptr_t)((pair * (kPageCopies * 4u)) + col)), inc,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // the symbols of one 32-bit word of input
    auto count_word = [&](const uint32_t w) {
        if (LAYOUT == HF_LAYOUT_BF16_EXP) {  // exponents: bits 7..14 and 23..30
            add(__builtin_amdgcn_ubfe(w, 8, 7), ((w & 0x80u) << 9) | 1u);
            add(__builtin_amdgcn_ubfe(w, 24, 7), (__builtin_amdgcn_ubfe(w, 23, 1) << 16) | 1u);
        } else {
            add(__builtin_amdgcn_ubfe(w, 1, 7), ((w & 1u) << 16) | 1u);
            add(__builtin_amdgcn_ubfe(w, 9, 7), ((w & 0x100u) << 8) | 1u);
            add(__builtin_amdgcn_ubfe(w, 17, 7), (w & 0x10000u) | 1u);
            add(w >> 25, (__builtin_amdgcn_ubfe(w, 24, 1) << 16) | 1u);
        }
    };
    auto zero = [&]() {
#pragma unroll
        for (int k = 0; k < kPageWords / 4 / 256; ++k)
            reinterpret_cast<uint4*>(h)[k * 256 + tid] = make_uint4(0, 0, 0, 0);
    };
    uint32_t acc0 = 0, acc1 = 0;  // bins 2 tid and 2 tid + 1 (tid < 128)
    // counted page -> its 256 counts; the table is cleared for the next page
    auto fold = [&](const uint32_t c) {
        __syncthreads();
        if (tid < 128) {
            uint32_t sum = 0;
#pragma unroll
            for (int i = 0; i < kPageCopies; ++i)
                sum += h[tid * kPageCopies + ((i + tid) & (kPageCopies - 1))];
            // (no half can carry into the other: a page holds at most 16384 symbols)
            const uint32_t odd = sum >> 16, even = (sum & 0xFFFFu) - odd;
            reinterpret_cast<uint2*>(freq_chunks + (size_t)c * 256)[tid] = make_uint2(even, odd);
            acc0 += even;
            acc1 += odd;
        }
        __syncthreads();
        zero();
        __syncthreads();
    };

    constexpr uint32_t kPerVec = LAYOUT == HF_LAYOUT_BF16_EXP ? 8 : 16;
    constexpr uint32_t kPageVecs = HF_CHUNK_SYMS / kPerVec;
    constexpr int kVecs = kPageVecs / 256;  // per thread and page: 8 (bf16) or 4
    const uint32_t nfull = (uint32_t)(n / HF_CHUNK_SYMS);
    const uint4* const base = reinterpret_cast<const uint4*>(data);
    // (a page behind the workgroup's last one is read from the stream's last full page instead and
    // never counted: the loads are issued on every path, so the counted waits stay exact and the
    // page requested ahead is still in flight while its predecessor is counted)
    auto load = [&](uint4 (&x)[kVecs], uint32_t c) {
        c = c < nfull ? c : nfull - 1u;
        const uint4* p = base + (size_t)c * kPageVecs + tid;
#pragma unroll
        for (int k = 0; k < kVecs; ++k) x[k] = stream_load(p + k * 256);
    };
    auto count = [&](const uint4 (&x)[kVecs]) {
#pragma unroll
        for (int k = 0; k < kVecs; ++k)
            count_word(x[k].x), count_word(x[k].y), count_word(x[k].z), count_word(x[k].w);
    };
    zero();
    __syncthreads();
    const uint32_t stride = gridDim.x;
    uint32_t c = blockIdx.x;
    if (c < nfull) {
        uint4 xa[kVecs], xb[kVecs];
        load(xa, c);
        for (;;) {
            load(xb, c + stride);
            count(xa);
            fold(c);
            c += stride;
            if (c >= nfull) break;
            load(xa, c + stride);
            count(xb);
            fold(c);
            c += stride;
            if (c >= nfull) break;
        }
    }
    // the stream's last, partial page
    if (nfull < nchunks && (nfull % stride) == blockIdx.x) {
        const uint32_t nvalid = (uint32_t)(n - (uint64_t)nfull * HF_CHUNK_SYMS);
        const uint4* const pv = base + (size_t)nfull * kPageVecs;
        const uint32_t nvec = nvalid / kPerVec;
        for (uint32_t v = tid; v < nvec; v += 256) {
            const uint4 x = pv[v];
            count_word(x.x), count_word(x.y), count_word(x.z), count_word(x.w);
        }
        for (uint32_t i = nvec * kPerVec + tid; i < nvalid; i += 256) {
            const uint32_t sym = LAYOUT == HF_LAYOUT_BF16_EXP
                                     ? (reinterpret_cast<const uint16_t*>(pv)[i] >> 7) & 0xFFu
                                     : reinterpret_cast<const uint8_t*>(pv)[i];
            add(sym >> 1, ((sym & 1u) << 16) | 1u);
        }
        fold(nfull);
    }
    if (freq256 && tid < 128) {
        if (acc0) atomicAdd(&freq256[2 * tid], (unsigned long long)acc0);
        if (acc1) atomicAdd(&freq256[2 * tid + 1], (unsigned long long)acc1);
    }
}

}  // namespace

// Per-chunk counts folded into a stream's: a workgroup per run of chunks, each of its four
// wavefronts takes every fourth row (256 counts, 16 bytes per lane), the four partial sums meet in
// LDS, and thread t adds bin t with one 64-bit atomic.
namespace {
constexpr uint32_t kFoldChunks = 64;
__global__ __launch_bounds__(256) void hf_k_fold_chunks(const uint32_t* __restrict__ freq_chunks,
                                                        const uint64_t nchunks,
                                                        unsigned long long* __restrict__ freq256) {
    __shared__ unsigned long long part[4][256];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t c0 = (uint64_t)blockIdx.x * kFoldChunks;
    const uint64_t c1 = c0 + kFoldChunks < nchunks ? c0 + kFoldChunks : nchunks;
    unsigned long long acc[4] = {0, 0, 0, 0};
    uint64_t c = c0 + wv;
    for (; c + 12 < c1; c += 16) {  // four rows in flight per wavefront
        uint4 r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            r[k] = reinterpret_cast<const uint4*>(freq_chunks + (c + 4 * k) * 256)[lane];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[0] += r[k].x, acc[1] += r[k].y, acc[2] += r[k].z, acc[3] += r[k].w;
        }
    }
    for (; c < c1; c += 4) {
        const uint4 r = reinterpret_cast<const uint4*>(freq_chunks + c * 256)[lane];
        acc[0] += r.x, acc[1] += r.y, acc[2] += r.z, acc[3] += r.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) part[wv][lane * 4 + k] = acc[k];
    __syncthreads();
    const unsigned long long sum = part[0][threadIdx.x] + part[1][threadIdx.x] +
                                   part[2][threadIdx.x] + part[3][threadIdx.x];
    if (sum) atomicAdd(&freq256[threadIdx.x], sum);
}
}  // namespace

void hf_launch_fold_chunks(const uint32_t* d_freq_chunks, uint64_t nchunks,
                           unsigned long long* d_freq256, hipStream_t s) {
    if (nchunks == 0) return;
    const unsigned grid = (unsigned)((nchunks + kFoldChunks - 1) / kFoldChunks);
    hipLaunchKernelGGL(hf_k_fold_chunks, dim3(grid), dim3(256), 0, s, d_freq_chunks, nchunks,
                       d_freq256);
}

void hf_launch_hist_chunks(const void* d_data, uint64_t n, int layout, uint32_t* d_freq_chunks,
                           hipStream_t s) {
    const uint64_t nchunks = (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS;
    if (nchunks == 0) return;
    const unsigned grid = (unsigned)((nchunks + kChunkWaves - 1) / kChunkWaves);
    if (layout == HF_LAYOUT_BF16_EXP)
        hipLaunchKernelGGL(hf_k_hist_chunks<HF_LAYOUT_BF16_EXP>, dim3(grid), dim3(kChunkWaves * 64),
                           0, s, d_data, n, (uint32_t)nchunks, d_freq_chunks);
    else
        hipLaunchKernelGGL(hf_k_hist_chunks<HF_LAYOUT_U8>, dim3(grid), dim3(kChunkWaves * 64), 0, s,
                           d_data, n, (uint32_t)nchunks, d_freq_chunks);
}

// per-page counts and (d_freq256 given) the stream's totals added into d_freq256
void hf_launch_hist_pages(const void* d_data, uint64_t n, int layout, uint32_t* d_freq_chunks,
                          unsigned long long* d_freq256, int num_cu, hipStream_t s) {
    const uint64_t nchunks = (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS;
    if (nchunks == 0) return;
    // resident workgroups, as many as a CU holds; a thread's 32-bit totals hold 2^18 pages
    // (registers bound the workgroups per CU: four for 16-bit elements, five for bytes)
    uint64_t grid = nchunks;
    const uint64_t cap = (uint64_t)num_cu * (layout == HF_LAYOUT_BF16_EXP ? 4 : 5);
    if (grid > cap) grid = cap;
    while (nchunks / grid >= (1ull << 18)) grid *= 2;
    if (layout == HF_LAYOUT_BF16_EXP)
        hipLaunchKernelGGL(hf_k_hist_pages<HF_LAYOUT_BF16_EXP>, dim3((unsigned)grid), dim3(256), 0, s,
                           d_data, n, (uint32_t)nchunks, d_freq_chunks, d_freq256);
    else
        hipLaunchKernelGGL(hf_k_hist_pages<HF_LAYOUT_U8>, dim3((unsigned)grid), dim3(256), 0, s,
                           d_data, n, (uint32_t)nchunks, d_freq_chunks, d_freq256);
}

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
