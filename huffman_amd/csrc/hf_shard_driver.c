/*
 * hf_shard_driver -- one process, N GPUs: a tensor partitioned into contiguous chunk-aligned
 * shards, one context and one RCCL rank per GPU, the global codebook from one all-reduce of the
 * 256 symbol counts.  Plain C over the C ABI of libhuffhip.so (include/huffhip.h) and RCCL: what a
 * host that is not Python (the reference's own side is Scala: JNA or Panama over the same calls)
 * does to run a sharded tensor.  The reference has no multi-device notion (one SymbolStat feeding
 * one tree builder, top_encoder.scala:29-41); the exchange step is this build's.
 *
 * usage: hf_shard_driver [--gpus N] [--elems TOTAL] [--seed S]
 * Prints one line per rank (shard range, local and global symbol counts, codebook hash, payload
 * words, round trip) and a summary line; exit status 0 iff every rank round-trips and all ranks
 * hold the same codebook.
 */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/huffhip.h"

#define MAX_GPUS 16
#define CHECK_HIP(x)                                                                 \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 2;                                                                \
        }                                                                            \
    } while (0)
#define CHECK_HF(x)                                                              \
    do {                                                                         \
        int r_ = (x);                                                            \
        if (r_ != 0) {                                                           \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hf_error_string(r_)); \
            return 2;                                                            \
        }                                                                        \
    } while (0)
#define CHECK_NCCL(x)                                                                 \
    do {                                                                              \
        ncclResult_t r_ = (x);                                                        \
        if (r_ != ncclSuccess) {                                                      \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, ncclGetErrorString(r_)); \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

/* element range of `rank`: contiguous, chunk aligned except for the global tail, sizes differing
 * by at most one chunk (the rule of huffman_amd/dist.py: shard_range) */
static void shard_range(uint64_t n, int rank, int world, uint64_t* lo, uint64_t* hi) {
    const uint64_t nchunks = (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS;
    const uint64_t base = nchunks / (uint64_t)world, extra = nchunks % (uint64_t)world;
    const uint64_t r = (uint64_t)rank;
    const uint64_t c_lo = r * base + (r < extra ? r : extra);
    const uint64_t c_hi = c_lo + base + (r < extra ? 1 : 0);
    *lo = c_lo * HF_CHUNK_SYMS < n ? c_lo * HF_CHUNK_SYMS : n;
    *hi = c_hi * HF_CHUNK_SYMS < n ? c_hi * HF_CHUNK_SYMS : n;
}

/* bf16 bit patterns of roughly Gaussian weights (sum of four uniforms, scaled to sigma 0.02) */
static uint64_t rng_state;
static uint32_t rng_next(void) {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(rng_state >> 33);
}
static uint16_t gaussian_bf16(void) {
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += (float)(rng_next() & 0xFFFFFF) / 16777216.0f;
    const float x = (s - 2.0f) * 0.0346f; /* sigma of the sum is 0.577 */
    uint32_t u;
    memcpy(&u, &x, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

static uint64_t fnv1a(const void* p, size_t n, uint64_t h) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

int main(int argc, char** argv) {
    int want = 0;
    uint64_t total = 8ull << 20;
    uint64_t seed = 1234;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--gpus")) want = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--elems")) total = strtoull(argv[i + 1], NULL, 10);
        else if (!strcmp(argv[i], "--seed")) seed = strtoull(argv[i + 1], NULL, 10);
        else {
            fprintf(stderr, "usage: %s [--gpus N] [--elems TOTAL] [--seed S]\n", argv[0]);
            return 2;
        }
    }
    int ndev = 0;
    CHECK_HIP(hipGetDeviceCount(&ndev));
    const int world = want > 0 ? want : ndev;
    if (world < 1 || world > ndev || world > MAX_GPUS) {
        fprintf(stderr, "%d GPUs asked for, %d visible (at most %d)\n", world, ndev, MAX_GPUS);
        return 2;
    }
    int devs[MAX_GPUS];
    ncclComm_t comms[MAX_GPUS];
    for (int r = 0; r < world; ++r) devs[r] = r;
    CHECK_NCCL(ncclCommInitAll(comms, world, devs));

    hipStream_t stream[MAX_GPUS];
    hf_ctx* ctx[MAX_GPUS];
    uint64_t lo[MAX_GPUS], hi[MAX_GPUS], local_sum[MAX_GPUS], words[MAX_GPUS];
    uint16_t* h_data[MAX_GPUS];
    void *d_data[MAX_GPUS], *d_freq[MAX_GPUS], *d_cb[MAX_GPUS], *d_side[MAX_GPUS], *d_pay[MAX_GPUS],
        *d_off[MAX_GPUS], *d_out[MAX_GPUS];
    uint64_t cap[MAX_GPUS];
    rng_state = seed;
    for (int r = 0; r < world; ++r) {
        shard_range(total, r, world, &lo[r], &hi[r]);
        const uint64_t n = hi[r] - lo[r], nch = hf_num_chunks(n);
        CHECK_HIP(hipSetDevice(r));
        CHECK_HIP(hipStreamCreate(&stream[r]));
        CHECK_HF(hf_ctx_create(&ctx[r], r, stream[r]));
        h_data[r] = (uint16_t*)malloc(2 * (n ? n : 1));
        for (uint64_t i = 0; i < n; ++i) h_data[r][i] = gaussian_bf16();
        cap[r] = hf_payload_bound_words(n) + 4;
        CHECK_HIP(hipMalloc(&d_data[r], 2 * n + 16));
        CHECK_HIP(hipMalloc(&d_out[r], 2 * n + 16));
        CHECK_HIP(hipMalloc(&d_side[r], n + 16));
        CHECK_HIP(hipMalloc(&d_pay[r], 4 * cap[r]));
        CHECK_HIP(hipMalloc(&d_off[r], 4 * (nch + 4)));
        CHECK_HIP(hipMalloc(&d_freq[r], 256 * 8));
        CHECK_HIP(hipMalloc(&d_cb[r], HF_CODEBOOK_DEV_BYTES));
        CHECK_HIP(hipMemcpyAsync(d_data[r], h_data[r], 2 * n, hipMemcpyHostToDevice, stream[r]));
        CHECK_HIP(hipMemsetAsync(d_freq[r], 0, 256 * 8, stream[r]));
        /* SymbolStat of this shard, on its own GPU */
        CHECK_HF(hf_symbol_stat_dev(ctx[r], d_data[r], n, HF_LAYOUT_BF16_EXP, (uint64_t*)d_freq[r]));
        uint64_t f[256];
        CHECK_HIP(hipMemcpyAsync(f, d_freq[r], sizeof f, hipMemcpyDeviceToHost, stream[r]));
        CHECK_HIP(hipStreamSynchronize(stream[r]));
        local_sum[r] = 0;
        for (int s = 0; s < 256; ++s) local_sum[r] += f[s];
    }
    /* the one exchange step: the counts summed over all ranks, one communicator per GPU */
    CHECK_NCCL(ncclGroupStart());
    for (int r = 0; r < world; ++r)
        CHECK_HF(hf_symbol_stat_allreduce(ctx[r], comms[r], (uint64_t*)d_freq[r]));
    CHECK_NCCL(ncclGroupEnd());

    int ok = 1;
    uint64_t hash0 = 0;
    for (int r = 0; r < world; ++r) {
        const uint64_t n = hi[r] - lo[r];
        CHECK_HIP(hipSetDevice(r));
        uint64_t f[256], gsum = 0;
        CHECK_HIP(hipMemcpyAsync(f, d_freq[r], sizeof f, hipMemcpyDeviceToHost, stream[r]));
        CHECK_HF(hf_system_codebook_dev(ctx[r], (const uint64_t*)d_freq[r], HF_MODE_HUFFMAN, 0, d_cb[r]));
        uint16_t code[256];
        uint8_t len[256];
        int mode = 0;
        uint32_t ent = 0;
        CHECK_HF(hf_codebook_download(ctx[r], d_cb[r], code, len, &mode, &ent));
        for (int s = 0; s < 256; ++s) gsum += f[s];
        const uint64_t h = fnv1a(len, sizeof len, fnv1a(code, sizeof code, 1469598103934665603ull));
        if (r == 0) hash0 = h;
        words[r] = 0;
        CHECK_HF(hf_encode(ctx[r], d_data[r], n, HF_LAYOUT_BF16_EXP, d_cb[r], (uint8_t*)d_side[r],
                           (uint32_t*)d_pay[r], cap[r], (uint32_t*)d_off[r], &words[r]));
        CHECK_HF(hf_decode(ctx[r], (const uint32_t*)d_pay[r], cap[r], (const uint32_t*)d_off[r],
                           (const uint8_t*)d_side[r], n, HF_LAYOUT_BF16_EXP, d_cb[r], d_out[r], 1));
        uint16_t* back = (uint16_t*)malloc(2 * (n ? n : 1));
        CHECK_HIP(hipMemcpyAsync(back, d_out[r], 2 * n, hipMemcpyDeviceToHost, stream[r]));
        CHECK_HIP(hipStreamSynchronize(stream[r]));
        const int same = memcmp(back, h_data[r], 2 * n) == 0;
        free(back);
        ok = ok && same && h == hash0 && gsum == total;
        printf("rank %d: elements [%llu, %llu) local_count_sum %llu global_count_sum %llu "
               "codebook_hash %016llx mode %d entropy_q16 %u payload_words %llu round_trip %s\n",
               r, (unsigned long long)lo[r], (unsigned long long)hi[r],
               (unsigned long long)local_sum[r], (unsigned long long)gsum, (unsigned long long)h,
               mode, ent, (unsigned long long)words[r], same ? "ok" : "MISMATCH");
    }
    uint64_t all_words = 0;
    for (int r = 0; r < world; ++r) all_words += words[r];
    printf("summary: world %d elements %llu payload_words %llu same_codebook_everywhere %s all_ok %s\n",
           world, (unsigned long long)total, (unsigned long long)all_words,
           ok ? "yes" : "no", ok ? "yes" : "no");
    for (int r = 0; r < world; ++r) {
        CHECK_HIP(hipSetDevice(r));
        hf_ctx_destroy(ctx[r]);
        hipFree(d_data[r]); hipFree(d_out[r]); hipFree(d_side[r]); hipFree(d_pay[r]);
        hipFree(d_off[r]); hipFree(d_freq[r]); hipFree(d_cb[r]);
        hipStreamDestroy(stream[r]);
        ncclCommDestroy(comms[r]);
        free(h_data[r]);
    }
    return ok ? 0 : 1;
}
