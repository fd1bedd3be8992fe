// Cycle stamps around the 256-key sorts of the codebook kernels (developer probe): one wavefront,
// 4 keys per lane, shader-clock stamps (s_memtime) before and after, repeated and averaged.
// Build (from the repository root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ihuffman_amd/csrc tools/sort_cycles.hip -o tools/sort_cycles
// -DSORT_SOURCE='"file"' times the sorts of another version of hf_codebook.hip.
#ifndef SORT_SOURCE
#define SORT_SOURCE "../huffman_amd/csrc/hf_codebook.hip"
#endif
#include SORT_SOURCE
#include <stdio.h>
#include <stdlib.h>

// a stamp that the keys pass through: the sort between two stamps can neither start before the
// first nor end after the second
template <class T>
__device__ __forceinline__ unsigned long long stamp(T (&k)[4]) {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(t), "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3])::"memory");
    return t;
}

__global__ __launch_bounds__(64) void probe(const uint32_t* in, uint32_t* out, unsigned long long* cyc,
                                            const int reps) {
    const int lane = threadIdx.x;
    unsigned long long c32 = 0, c64 = 0;
    for (int r = 0; r < reps; ++r) {
        uint32_t k[4];
        u64 q[4];
        for (int i = 0; i < 4; ++i) {
            k[i] = in[(r * 256 + lane * 4 + i)];
            q[i] = ((u64)k[i] << 20) | (u64)(lane * 4 + i);
        }
        const unsigned long long t0 = stamp(k);
        sort256_desc_u32(k, lane);
        const unsigned long long t1 = stamp(k);
        const unsigned long long t2 = stamp(q);
        sort256_desc(q, lane);
        const unsigned long long t3 = stamp(q);
        c32 += t1 - t0;
        c64 += t3 - t2;
        for (int i = 0; i < 4; ++i) out[r * 256 + lane * 4 + i] = k[i] ^ (uint32_t)(q[i] >> 20);
    }
    if (lane == 0) cyc[0] = c32, cyc[1] = c64;
}

int main() {
    const int reps = 64;
    uint32_t* h = (uint32_t*)malloc(reps * 256 * 4);
    srand(5);
    for (int i = 0; i < reps * 256; ++i) h[i] = ((uint32_t)(rand() & 0xFFFFFF) << 8) | (255u - (i & 255));
    uint32_t *din, *dout;
    unsigned long long* dc;
    hipMalloc(&din, reps * 256 * 4);
    hipMalloc(&dout, reps * 256 * 4);
    hipMalloc(&dc, 16);
    hipMemcpy(din, h, reps * 256 * 4, hipMemcpyHostToDevice);
    unsigned long long c[2];
    for (int pass = 0; pass < 2; ++pass) {  // (the first pass pays the instruction cache misses)
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, dout, dc, reps);
        hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
    }
    uint32_t* o = (uint32_t*)malloc(reps * 256 * 4);
    hipMemcpy(o, dout, reps * 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < reps; ++r)
        for (int i = 0; i < 256; ++i) bad += o[r * 256 + i] != 0;  // both sorts give the same order
    printf("sort256_desc_u32: %.0f ticks   sort256_desc (64-bit keys): %.0f ticks of s_memtime   (%d sorts "
           "each, orders %s)\n", (double)c[0] / reps, (double)c[1] / reps, reps, bad ? "DIFFER" : "agree");
    return bad != 0;
}
