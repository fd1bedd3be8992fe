// Plain streaming rates of one GPU with hand-written kernels: what a copy, a read and a fill reach
// with ordinary and with non-temporal accesses, at several loads in flight per lane.  The numbers
// say how far the coding kernels are from what the memory system gives a kernel that does nothing
// but move bytes.   Build: hipcc -O3 --offload-arch=gfx950 tools/membw_hip.hip -o tools/membw_hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ u32x4 ld(const u32x4* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <int NT>
__device__ __forceinline__ void st(u32x4* p, u32x4 v) {
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

// each workgroup walks tiles of 256 * K vectors; K loads in flight per lane
template <int K, int NTL, int NTS>
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst,
                                              size_t nvec) {
    const size_t tile = (size_t)256 * K;
    for (size_t base = (size_t)blockIdx.x * tile; base + tile <= nvec; base += (size_t)gridDim.x * tile) {
        u32x4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = ld<NTL>(src + base + k * 256 + threadIdx.x);
#pragma unroll
        for (int k = 0; k < K; ++k) st<NTS>(dst + base + k * 256 + threadIdx.x, v[k]);
    }
}

template <int K, int NTL>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ src, uint32_t* __restrict__ out,
                                              size_t nvec) {
    const size_t tile = (size_t)256 * K;
    uint32_t acc = 0;
    for (size_t base = (size_t)blockIdx.x * tile; base + tile <= nvec; base += (size_t)gridDim.x * tile) {
        u32x4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = ld<NTL>(src + base + k * 256 + threadIdx.x);
#pragma unroll
        for (int k = 0; k < K; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) out[0] = acc;  // keeps the loads alive
}

template <int K, int NTS>
__global__ __launch_bounds__(256) void k_fill(u32x4* __restrict__ dst, size_t nvec) {
    const size_t tile = (size_t)256 * K;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (size_t base = (size_t)blockIdx.x * tile; base + tile <= nvec; base += (size_t)gridDim.x * tile) {
#pragma unroll
        for (int k = 0; k < K; ++k) st<NTS>(dst + base + k * 256 + threadIdx.x, v);
    }
}

// the decoder's mix: read R vectors, write W vectors per tile step (R:W = 2:3 ~ 1.33 B in, 2 B out)
template <int R, int W, int NTL, int NTS>
__global__ __launch_bounds__(256) void k_mix(const u32x4* __restrict__ src, u32x4* __restrict__ dst,
                                             size_t nsteps) {
    for (size_t s = blockIdx.x; s < nsteps; s += gridDim.x) {
        u32x4 v[R];
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = ld<NTL>(src + (s * R + k) * 256 + threadIdx.x);
#pragma unroll
        for (int k = 1; k < R; ++k) v[0] ^= v[k];  // every load is used
#pragma unroll
        for (int k = 0; k < W; ++k) st<NTS>(dst + (s * W + k) * 256 + threadIdx.x, v[0]);
    }
}

// The coding kernels' pattern: one wavefront per 16384-element chunk, walking its own region 64
// lanes x 16 bytes (16-bit side) or x 8 bytes (byte side) per round, eight rounds in flight.
// DEC: read the byte side, write the 16-bit side; otherwise the reverse.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int DEC, int NT, int B = 8>
__global__ __launch_bounds__(256) void k_wave_chunks(const uint8_t* __restrict__ src,
                                                     uint8_t* __restrict__ dst, size_t nchunks) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t c = wave; c < nchunks; c += nwaves) {
        for (int r0 = 0; r0 < 32; r0 += B) {
            if (DEC) {
                u32x2 v[B];
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const u32x2* q = (const u32x2*)(src + c * 16384 + (r0 + k) * 512 + lane * 8);
                    v[k] = NT ? __builtin_nontemporal_load(q) : *q;
                }
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    u32x4* q = (u32x4*)(dst + c * 32768 + (r0 + k) * 1024 + lane * 16);
                    const u32x4 o = {v[k].x, v[k].y, v[k].x ^ 1u, v[k].y ^ 2u};
                    if (NT) __builtin_nontemporal_store(o, q); else *q = o;
                }
            } else {
                u32x4 v[B];
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const u32x4* q = (const u32x4*)(src + c * 32768 + (r0 + k) * 1024 + lane * 16);
                    v[k] = NT ? __builtin_nontemporal_load(q) : *q;
                }
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    u32x2* q = (u32x2*)(dst + c * 16384 + (r0 + k) * 512 + lane * 8);
                    const u32x2 o = {v[k].x ^ v[k].z, v[k].y ^ v[k].w};
                    if (NT) __builtin_nontemporal_store(o, q); else *q = o;
                }
            }
        }
    }
}

// The encoder's direction with the byte side written 16 bytes per lane: the bytes of two rounds are
// paired up so that a wavefront's store covers 1 KiB instead of 512 bytes (what an exchange
// between neighbouring lanes would make possible in the encoder).
template <int NT>
__global__ __launch_bounds__(256) void k_wave_chunks_enc16(const uint8_t* __restrict__ src,
                                                           uint8_t* __restrict__ dst, size_t nchunks) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t c = wave; c < nchunks; c += nwaves) {
        for (int r0 = 0; r0 < 32; r0 += 8) {
            u32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const u32x4* q = (const u32x4*)(src + c * 32768 + (r0 + k) * 1024 + lane * 16);
                v[k] = NT ? __builtin_nontemporal_load(q) : *q;
            }
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                u32x4* q = (u32x4*)(dst + c * 16384 + (r0 + k) * 512 + lane * 16);
                const u32x4 o = {v[k].x ^ v[k].z, v[k].y ^ v[k].w, v[k + 1].x ^ v[k + 1].z,
                                 v[k + 1].y ^ v[k + 1].w};
                if (NT) __builtin_nontemporal_store(o, q); else *q = o;
            }
        }
    }
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / reps;
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 4096) << 20;  // MiB per buffer
    const size_t nvec = bytes / 16;
    void *src, *dst;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&dst, bytes));
    CK(hipMemset(src, 1, bytes));
    CK(hipMemset(dst, 2, bytes));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cu = p.multiProcessorCount;
    printf("%s, %d CUs, %zu MiB per buffer\n", p.name, cu, bytes >> 20);
    const int reps = 10;
    const double gb = (double)bytes / 1e9;
#define ROW(name, traffic_gb, ...)                                        \
    {                                                                     \
        double ms = time_ms([&] { __VA_ARGS__; }, reps);                  \
        printf("%-34s %8.3f ms %8.0f GB/s\n", name, ms, (traffic_gb) / ms * 1e3); \
    }
    for (int wg_per_cu : {8, 0}) {
        const unsigned g = wg_per_cu ? (unsigned)(cu * wg_per_cu) : (unsigned)(nvec / (256 * 4));
        printf("-- grid %u (%d workgroups per CU%s)\n", g, wg_per_cu, wg_per_cu ? "" : ": one tile each");
        ROW("copy  K=4 plain", 2 * gb, (k_copy<4, 0, 0><<<g, 256>>>((const u32x4*)src, (u32x4*)dst, nvec)));
        ROW("copy  K=4 nt-store", 2 * gb, (k_copy<4, 0, 1><<<g, 256>>>((const u32x4*)src, (u32x4*)dst, nvec)));
        ROW("copy  K=4 nt-load nt-store", 2 * gb, (k_copy<4, 1, 1><<<g, 256>>>((const u32x4*)src, (u32x4*)dst, nvec)));
        ROW("copy  K=8 plain", 2 * gb, (k_copy<8, 0, 0><<<g, 256>>>((const u32x4*)src, (u32x4*)dst, nvec)));
        ROW("copy  K=8 nt-load nt-store", 2 * gb, (k_copy<8, 1, 1><<<g, 256>>>((const u32x4*)src, (u32x4*)dst, nvec)));
        ROW("read  K=4 plain", gb, (k_read<4, 0><<<g, 256>>>((const u32x4*)src, (uint32_t*)dst, nvec)));
        ROW("read  K=8 plain", gb, (k_read<8, 0><<<g, 256>>>((const u32x4*)src, (uint32_t*)dst, nvec)));
        ROW("read  K=8 nt", gb, (k_read<8, 1><<<g, 256>>>((const u32x4*)src, (uint32_t*)dst, nvec)));
        ROW("fill  K=4 plain", gb, (k_fill<4, 0><<<g, 256>>>((u32x4*)dst, nvec)));
        ROW("fill  K=4 nt", gb, (k_fill<4, 1><<<g, 256>>>((u32x4*)dst, nvec)));
        {
            const size_t steps = nvec / (256 * 3);
            const double t = (double)steps * 256 * 16 * 5 / 1e9;
            ROW("mix 2 read : 3 written plain", t, (k_mix<2, 3, 0, 0><<<g, 256>>>((const u32x4*)src, (u32x4*)dst, steps)));
            ROW("mix 2 read : 3 written nt", t, (k_mix<2, 3, 1, 1><<<g, 256>>>((const u32x4*)src, (u32x4*)dst, steps)));
            ROW("mix 3 read : 2 written plain", t, (k_mix<3, 2, 0, 0><<<g, 256>>>((const u32x4*)src, (u32x4*)dst, steps)));
            ROW("mix 3 read : 2 written nt", t, (k_mix<3, 2, 1, 1><<<g, 256>>>((const u32x4*)src, (u32x4*)dst, steps)));
        }
    }
    {
        // the 16-bit side fills one buffer: bytes / 32 KiB chunks, 3 bytes moved per element
        const size_t nchunks = bytes / 32768;
        const double t = (double)nchunks * 16384 * 3 / 1e9;
        for (int wg_per_cu : {2, 6, 0}) {
            const unsigned g = wg_per_cu ? (unsigned)(cu * wg_per_cu) : (unsigned)(nchunks / 4);
            printf("-- wavefront per chunk, %d waves per CU (0: one chunk per wavefront, workgroups come and go)\n", wg_per_cu * 4);
            ROW("decoder-like (1 B in, 2 B out) plain", t, (k_wave_chunks<1, 0><<<g, 256>>>((const uint8_t*)src, (uint8_t*)dst, nchunks)));
            ROW("decoder-like (1 B in, 2 B out) nt", t, (k_wave_chunks<1, 1><<<g, 256>>>((const uint8_t*)src, (uint8_t*)dst, nchunks)));
            ROW("encoder-like (2 B in, 1 B out) plain", t, (k_wave_chunks<0, 0><<<g, 256>>>((const uint8_t*)src, (uint8_t*)dst, nchunks)));
            ROW("encoder-like (2 B in, 1 B out) nt", t, (k_wave_chunks<0, 1><<<g, 256>>>((const uint8_t*)src, (uint8_t*)dst, nchunks)));
            ROW("encoder-like, 16-byte stores plain", t, (k_wave_chunks_enc16<0><<<g, 256>>>((const uint8_t*)src, (uint8_t*)dst, nchunks)));
            ROW("encoder-like, 16-byte stores nt", t, (k_wave_chunks_enc16<1><<<g, 256>>>((const uint8_t*)src, (uint8_t*)dst, nchunks)));
        }
        {
            const unsigned g = (unsigned)(cu * 2);
            printf("-- wavefront per chunk, 8 waves per CU, rounds per burst 2 / 4 / 16 (non-temporal)\n");
            ROW("encoder-like burst 2", t, (k_wave_chunks<0, 1, 2><<<g, 256>>>((const uint8_t*)src, (uint8_t*)dst, nchunks)));
            ROW("encoder-like burst 4", t, (k_wave_chunks<0, 1, 4><<<g, 256>>>((const uint8_t*)src, (uint8_t*)dst, nchunks)));
            ROW("encoder-like burst 16", t, (k_wave_chunks<0, 1, 16><<<g, 256>>>((const uint8_t*)src, (uint8_t*)dst, nchunks)));
            ROW("decoder-like burst 2", t, (k_wave_chunks<1, 1, 2><<<g, 256>>>((const uint8_t*)src, (uint8_t*)dst, nchunks)));
            ROW("decoder-like burst 4", t, (k_wave_chunks<1, 1, 4><<<g, 256>>>((const uint8_t*)src, (uint8_t*)dst, nchunks)));
            ROW("decoder-like burst 16", t, (k_wave_chunks<1, 1, 16><<<g, 256>>>((const uint8_t*)src, (uint8_t*)dst, nchunks)));
        }
        const size_t steps = nvec / (256 * 2);
        const double t2 = (double)steps * 256 * 16 * 3 / 1e9;
        for (int wg_per_cu : {8, 0}) {
            const unsigned g = wg_per_cu ? (unsigned)(cu * wg_per_cu) : (unsigned)steps;
            printf("-- workgroup tiles, grid %u\n", g);
            ROW("mix 1 read : 2 written plain", t2, (k_mix<1, 2, 0, 0><<<g, 256>>>((const u32x4*)src, (u32x4*)dst, steps)));
            ROW("mix 1 read : 2 written nt", t2, (k_mix<1, 2, 1, 1><<<g, 256>>>((const u32x4*)src, (u32x4*)dst, steps)));
            ROW("mix 2 read : 1 written plain", t2, (k_mix<2, 1, 0, 0><<<g, 256>>>((const u32x4*)src, (u32x4*)dst, steps)));
            ROW("mix 2 read : 1 written nt", t2, (k_mix<2, 1, 1, 1><<<g, 256>>>((const u32x4*)src, (u32x4*)dst, steps)));
        }
    }
    ROW("hipMemcpyDtoD", 2 * gb, CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0)));
    CK(hipFree(src));
    CK(hipFree(dst));
    return 0;
}
