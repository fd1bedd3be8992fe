// C ABI of libhuffhip.so (declarations and reference citations: include/huffhip.h).
// Host side only: argument checks, the per-context workspace, stream-ordered launches.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <new>

#include "hf_common.h"

struct hf_ctx {
    int device;
    hipStream_t stream;
    int num_cu;
    hf_stage_io* d_io;
    hf_stage_io* h_io;  // pinned
    hf_cb_dev* d_cb;    // scratch codebook for hf_codebook_download staging
    hf_cb_dev* h_cb;    // pinned
    unsigned long long* d_freq;
    unsigned long long* h_freq;  // pinned
    unsigned long long* d_desc;
    size_t desc_cap;    // entries
    uint32_t* d_ctl;    // [1] error bits, [2] total words of the last encode; 64-byte block
    uint32_t* h_ctl;    // pinned, 16 words
    uint32_t* d_spill;
    void* d_cbscratch;  // per-block codebook batch scratch, allocated on first use
    unsigned long long* d_scan_aux;  // group totals of the slot-offset scan, allocated on first use
    size_t scan_aux_bytes;
    void* d_page_tab;   // decode tables of a stream-wide codebook for the page decoder (one call's)
    void* d_static_cbscratch;  // this device's copy of `scratch` below (never freed)
    int static_slot;           // slice of it this context holds, -1 before its first static call
    hipEvent_t ev0, ev1;
    int timing;
    int timed;
    int last_hip;
};

// Scratch of hf_system_codebook_chunks_static, the entry point that allocates nothing:
// HF_STATIC_CB_SLOTS slices, each with room for HF_STATIC_CB_MAX_CHUNKS per-block codebooks in
// flight between the three codebook kernels.  Every GPU the library runs on holds its own copy
// until the process ends; host code never uses the array's name as an address, only the pointer
// hipGetSymbolAddress returned for the device.
static constexpr size_t kStaticSliceBytes = (size_t)HF_STATIC_CB_MAX_CHUNKS * HF_CB_SCRATCH_PER_CHUNK;
static __device__ __align__(256) unsigned char scratch[HF_STATIC_CB_SLOTS * kStaticSliceBytes];
// which slices of each GPU's copy are held by a live context (bit i = slice i)
static constexpr int kMaxDevices = 64;
static std::atomic<unsigned> g_static_slices[kMaxDevices];

namespace {

#define HF_HIP(expr)                                   \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) {                        \
            ctx->last_hip = (int)e_;                   \
            return HF_ERR_HIP;                         \
        }                                              \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
inline bool layout_ok(int l) {
    return l == HF_LAYOUT_U8 || l == HF_LAYOUT_BF16_EXP || l == HF_LAYOUT_FP8;
}
inline int norm_layout(int l) { return l == HF_LAYOUT_FP8 ? HF_LAYOUT_U8 : l; }
// layouts whose elements are split into a symbol byte and a bit-packed side field
inline bool field_layout(int l) {
    return l == HF_LAYOUT_FP16_EXP || l == HF_LAYOUT_FP8_E4M3_EXP || l == HF_LAYOUT_FP4_E2M1_EXP;
}
// layouts that pass through hf_fields_split / hf_fields_merge around the byte-symbol kernels
inline bool front_end_layout(int l) { return field_layout(l) || l == HF_LAYOUT_BITPLANE8; }
// bytes of the side section a stream of n elements carries
inline uint64_t side_bytes(uint64_t n, int l) {
    return l == HF_LAYOUT_BF16_EXP ? n : field_layout(l) ? hf_fields_side_bytes_impl(n, l) : 0;
}
inline uint64_t mask_w(int W) { return W >= 64 ? ~0ull : ((1ull << W) - 1ull); }
inline int log2ceil_i(int x) {
    int b = 0;
    while ((1 << b) < x) ++b;
    return b;
}

__global__ void hf_k_zero_words(uint32_t* p, const uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
void zero_words(uint32_t* d_p, uint64_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(hf_k_zero_words, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_p, n);
}

int kerr_to_code(uint32_t bits) {
    if (bits & HF_KERR_TIMEOUT) return HF_ERR_TIMEOUT;
    if (bits & HF_KERR_OVERFLOW) return HF_ERR_OVERFLOW;
    if (bits & HF_KERR_CORRUPT) return HF_ERR_CORRUPT;
    return HF_OK;
}

// In-kernel errors are sticky: the kernels OR their bits into d_ctl[1], and the word is cleared
// only when it is reported -- by a call that synchronises, by hf_sync, or by hf_flush -- so that
// the errors of calls that only enqueue work are not lost.  Expects the stream synchronised and
// h_ctl holding a copy of d_ctl.
int report_kernel_errors(hf_ctx* ctx) {
    const uint32_t bits = ctx->h_ctl[1];
    if (bits) HF_HIP(hipMemsetAsync(ctx->d_ctl + 1, 0, sizeof(uint32_t), ctx->stream));
    return kerr_to_code(bits);
}

// Makes the context's device current for the duration of a call and puts the caller's device
// back on the way out, so that a call on a context of another GPU does not move the thread's
// later allocations and launches there.
struct device_guard {
    int prev = -1;
    int enter(hf_ctx* ctx) {
        int cur = -1;
        HF_HIP(hipGetDevice(&cur));
        if (cur != ctx->device) {
            HF_HIP(hipSetDevice(ctx->device));
            prev = cur;
        }
        return HF_OK;
    }
    ~device_guard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
#define HF_USE_DEVICE(ctx)  \
    device_guard dev_guard_; \
    if (int rc_ = dev_guard_.enter(ctx)) return rc_

void tic(hf_ctx* ctx) {
    if (ctx->timing) {
        (void)hipEventRecord(ctx->ev0, ctx->stream);
    }
}
void toc(hf_ctx* ctx) {
    if (ctx->timing) {
        (void)hipEventRecord(ctx->ev1, ctx->stream);
        ctx->timed = 1;
    }
}

// ---- the context's lazily sized workspaces ------------------------------------------------------
// Each grows on demand: a call that needs more than the context holds synchronises the stream,
// frees and allocates.  hf_ctx_reserve makes all of them up front, so that later calls within the
// reserved size only enqueue work (what stream capture and allocation-free loops need).

// tile descriptors of the look-back encoder (two words in front of them hold the tile ticket)
int ensure_desc(hf_ctx* ctx, uint64_t nchunks) {
    if (nchunks <= ctx->desc_cap) return HF_OK;
    // grown geometrically (at least 64 Ki descriptors, 512 KiB), so that a context sees a
    // handful of allocations over its life, whatever sizes it is asked for
    if (ctx->d_desc) {
        HF_HIP(hipStreamSynchronize(ctx->stream));
        HF_HIP(hipFree(ctx->d_desc));
        ctx->d_desc = nullptr;
    }
    size_t cap = ctx->desc_cap ? ctx->desc_cap * 2 : (size_t)65536;
    ctx->desc_cap = 0;
    while (cap < nchunks + 2) cap *= 2;
    HF_HIP(hipMalloc(&ctx->d_desc, (cap + 2) * sizeof(unsigned long long)));
    ctx->desc_cap = cap;
    return HF_OK;
}
// spill area of the look-back encoder (~100 MB: two staged chunks per resident wavefront)
int ensure_spill(hf_ctx* ctx) {
    if (!ctx->d_spill) HF_HIP(hipMalloc(&ctx->d_spill, hf_encode_spill_bytes(ctx->num_cu)));
    return HF_OK;
}
// batch scratch of the per-block codebook kernels (68 MB)
int ensure_cbscratch(hf_ctx* ctx) {
    if (!ctx->d_cbscratch) HF_HIP(hipMalloc(&ctx->d_cbscratch, hf_chunk_cb_scratch_bytes()));
    return HF_OK;
}
// group totals of the offset scan of the slotted and paged encoders, and behind them the dump
// area for words of a chunk that outgrows its slot
int ensure_scan_aux(hf_ctx* ctx, uint64_t nchunks) {
    const size_t aux = hf_scan_aux_bytes(nchunks + 1);
    if (aux <= ctx->scan_aux_bytes) return HF_OK;
    if (ctx->d_scan_aux) {
        HF_HIP(hipStreamSynchronize(ctx->stream));
        HF_HIP(hipFree(ctx->d_scan_aux));
        ctx->d_scan_aux = nullptr;
        ctx->scan_aux_bytes = 0;
    }
    size_t cap = (size_t)1 << 16;  // 8 Mi chunks' worth; doubled until it holds this call
    while (cap < aux) cap *= 2;
    HF_HIP(hipMalloc(&ctx->d_scan_aux, cap + hf_slot_dump_bytes()));
    ctx->scan_aux_bytes = cap;
    return HF_OK;
}

// run one table stage: h_io (pinned) -> device -> kernel -> h_io, synchronous
int run_stage(hf_ctx* ctx, int N, int op, int lmax, unsigned long long thr) {
    ctx->h_io->status = -1;
    HF_HIP(hipMemcpyAsync(ctx->d_io, ctx->h_io, sizeof(hf_stage_io), hipMemcpyHostToDevice,
                          ctx->stream));
    hf_launch_stage(ctx->d_io, N, op, lmax, thr, ctx->stream);
    HF_HIP(hipGetLastError());
    HF_HIP(hipMemcpyAsync(ctx->h_io, ctx->d_io, sizeof(hf_stage_io), hipMemcpyDeviceToHost,
                          ctx->stream));
    HF_HIP(hipStreamSynchronize(ctx->stream));
    return ctx->h_io->status == 0 ? HF_OK : HF_ERR_HIP;
}

}  // namespace

extern "C" {

int hf_abi_version(void) { return HF_ABI_VERSION; }

const char* hf_error_string(int code) {
    switch (code) {
        case HF_OK: return "ok";
        case HF_ERR_ARG: return "bad argument";
        case HF_ERR_ALIGN: return "device pointer not 16-byte aligned";
        case HF_ERR_RANGE: return "value out of range";
        case HF_ERR_HIP: return "HIP runtime error";
        case HF_ERR_OVERFLOW: return "payload buffer too small";
        case HF_ERR_TIMEOUT: return "in-kernel wait timed out";
        case HF_ERR_CORRUPT: return "corrupt payload or offsets";
        case HF_ERR_NOMEM: return "out of memory";
        case HF_ERR_RCCL: return "RCCL not loadable or collective failed";
        default: return "unknown error";
    }
}

int hf_ctx_create(hf_ctx** out, int device, void* stream) {
    if (!out) return HF_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return HF_ERR_HIP;
    hf_ctx* ctx = new (std::nothrow) hf_ctx();
    if (!ctx) return HF_ERR_NOMEM;
    memset(ctx, 0, sizeof(*ctx));
    ctx->device = device;
    ctx->static_slot = -1;
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    bool ok = hipSetDevice(device) == hipSuccess;
    hipDeviceProp_t prop;
    ok = ok && hipGetDeviceProperties(&prop, device) == hipSuccess;
    ctx->num_cu = ok ? prop.multiProcessorCount : 0;
    ok = ok && hipMalloc(&ctx->d_io, sizeof(hf_stage_io)) == hipSuccess;
    ok = ok && hipHostMalloc(&ctx->h_io, sizeof(hf_stage_io)) == hipSuccess;
    ok = ok && hipMalloc(&ctx->d_cb, HF_CODEBOOK_DEV_BYTES) == hipSuccess;
    ok = ok && hipHostMalloc(&ctx->h_cb, HF_CODEBOOK_DEV_BYTES) == hipSuccess;
    ok = ok && hipMalloc(&ctx->d_freq, 256 * sizeof(unsigned long long)) == hipSuccess;
    ok = ok && hipHostMalloc(&ctx->h_freq, 256 * sizeof(unsigned long long)) == hipSuccess;
    ok = ok && hipMalloc(&ctx->d_ctl, 64) == hipSuccess;
    ok = ok && hipMalloc(&ctx->d_page_tab, hf_page_tab_image_bytes()) == hipSuccess;
    ok = ok && hipHostMalloc(&ctx->h_ctl, 64) == hipSuccess;
    ok = ok && hipMemset(ctx->d_ctl, 0, 64) == hipSuccess;  // ticket and the sticky error word
    // (the encoder's spill area, ~100 MB, is allocated by the first packed encode)
    ok = ok && hipEventCreate(&ctx->ev0) == hipSuccess;
    ok = ok && hipEventCreate(&ctx->ev1) == hipSuccess;
    // this device's address of the static scratch block (the device is current here)
    ok = ok && hipGetSymbolAddress(&ctx->d_static_cbscratch, HIP_SYMBOL(scratch)) == hipSuccess &&
         ctx->d_static_cbscratch != nullptr;
    if (!ok) {
        hf_ctx_destroy(ctx);
        if (prev_dev >= 0 && prev_dev != device) (void)hipSetDevice(prev_dev);
        return HF_ERR_HIP;
    }
    if (prev_dev >= 0 && prev_dev != device) (void)hipSetDevice(prev_dev);
    *out = ctx;
    return HF_OK;
}

int hf_ctx_destroy(hf_ctx* ctx) {
    if (!ctx) return HF_ERR_ARG;
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    (void)hipSetDevice(ctx->device);
    if (ctx->stream || ctx->d_io) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_io) (void)hipFree(ctx->d_io);
    if (ctx->h_io) (void)hipHostFree(ctx->h_io);
    if (ctx->d_cb) (void)hipFree(ctx->d_cb);
    if (ctx->h_cb) (void)hipHostFree(ctx->h_cb);
    if (ctx->d_freq) (void)hipFree(ctx->d_freq);
    if (ctx->h_freq) (void)hipHostFree(ctx->h_freq);
    if (ctx->d_desc) (void)hipFree(ctx->d_desc);
    if (ctx->d_ctl) (void)hipFree(ctx->d_ctl);
    if (ctx->d_page_tab) (void)hipFree(ctx->d_page_tab);
    if (ctx->h_ctl) (void)hipHostFree(ctx->h_ctl);
    if (ctx->d_spill) (void)hipFree(ctx->d_spill);
    if (ctx->d_cbscratch) (void)hipFree(ctx->d_cbscratch);
    if (ctx->d_scan_aux) (void)hipFree(ctx->d_scan_aux);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->static_slot >= 0)
        g_static_slices[ctx->device].fetch_and(~(1u << ctx->static_slot), std::memory_order_acq_rel);
    if (prev_dev >= 0 && prev_dev != ctx->device) (void)hipSetDevice(prev_dev);
    delete ctx;
    return HF_OK;
}

int hf_sync(hf_ctx* ctx) {
    if (!ctx) return HF_ERR_ARG;
    HF_USE_DEVICE(ctx);
    HF_HIP(hipMemcpyAsync(ctx->h_ctl, ctx->d_ctl, 64, hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(hipStreamSynchronize(ctx->stream));
    return report_kernel_errors(ctx);
}

int hf_flush(hf_ctx* ctx) {
    if (!ctx) return HF_ERR_ARG;
    HF_USE_DEVICE(ctx);
    HF_HIP(hipStreamSynchronize(ctx->stream));
    HF_HIP(hipMemsetAsync(ctx->d_freq, 0, 256 * sizeof(unsigned long long), ctx->stream));
    HF_HIP(hipMemsetAsync(ctx->d_ctl, 0, 64, ctx->stream));
    HF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->last_hip = 0;
    ctx->timed = 0;
    return HF_OK;
}

int hf_ctx_reserve(hf_ctx* ctx, uint64_t max_n, unsigned flags) {
    if (!ctx || (flags & ~(unsigned)HF_RESERVE_ALL) != 0) return HF_ERR_ARG;
    if (max_n > (1ull << 40)) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    const uint64_t nchunks = hf_num_chunks(max_n);
    if (flags & HF_RESERVE_ENCODE) {
        if (int rc = ensure_desc(ctx, nchunks)) return rc;
        if (int rc = ensure_spill(ctx)) return rc;
    }
    if (flags & HF_RESERVE_CODEBOOKS)
        if (int rc = ensure_cbscratch(ctx)) return rc;
    if (flags & HF_RESERVE_PLACED) {
        if (int rc = ensure_scan_aux(ctx, nchunks)) return rc;
        if (int rc = ensure_desc(ctx, nchunks)) return rc;  // (pages placed while they are encoded)
    }
    return HF_OK;
}

int hf_zero_dev(hf_ctx* ctx, void* d_ptr, uint64_t nwords) {
    if (!ctx || (nwords && !d_ptr)) return HF_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(d_ptr) & 3u) return HF_ERR_ALIGN;
    if (nwords > (1ull << 40)) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    zero_words(static_cast<uint32_t*>(d_ptr), nwords, ctx->stream);
    HF_HIP(hipGetLastError());
    return HF_OK;
}

int hf_last_hip_error(hf_ctx* ctx) { return ctx ? ctx->last_hip : 0; }

int hf_set_timing(hf_ctx* ctx, int enabled) {
    if (!ctx) return HF_ERR_ARG;
    ctx->timing = enabled != 0;
    ctx->timed = 0;
    return HF_OK;
}

int hf_last_kernel_ms(hf_ctx* ctx, float* ms) {
    if (!ctx || !ms) return HF_ERR_ARG;
    if (!ctx->timed) return HF_ERR_ARG;
    HF_USE_DEVICE(ctx);
    HF_HIP(hipEventSynchronize(ctx->ev1));
    HF_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return HF_OK;
}

/* ---- SymbolStat ---------------------------------------------------------------------------- */
int hf_symbol_stat_dev(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                       uint64_t* d_freq256) {
    if (!ctx || !d_freq256 || !layout_ok(layout) || (n && !d_data)) return HF_ERR_ARG;
    if (!aligned16(d_data) || !aligned16(d_freq256)) return HF_ERR_ALIGN;
    if (n > (1ull << 40)) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    tic(ctx);
    hf_launch_hist(d_data, n, norm_layout(layout), reinterpret_cast<unsigned long long*>(d_freq256),
                   ctx->num_cu, ctx->stream);
    const hipError_t launch_err = hipGetLastError();  // before the event record can mask it
    toc(ctx);
    HF_HIP(launch_err);
    return HF_OK;
}

// ncclAllReduce of the RCCL this process uses, looked up on first use: the library is half a
// gigabyte and only callers that shard over GPUs need it, so libhuffhip.so does not link it.
// (dlopen by soname returns the copy that is already loaded, whoever loaded it.)
typedef int (*hf_nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
static hf_nccl_allreduce_fn rccl_allreduce() {
    static std::atomic<void*> fn{nullptr};
    void* f = fn.load(std::memory_order_acquire);
    if (!f) {
        const char* name = getenv("HF_RCCL_LIB");
        void* h = dlopen(name && *name ? name : "librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return nullptr;
        f = dlsym(h, "ncclAllReduce");
        if (f) fn.store(f, std::memory_order_release);
    }
    return reinterpret_cast<hf_nccl_allreduce_fn>(f);
}

int hf_symbol_stat_allreduce(hf_ctx* ctx, void* nccl_comm, uint64_t* d_freq256) {
    if (!ctx || !nccl_comm || !d_freq256) return HF_ERR_ARG;
    if (!aligned16(d_freq256)) return HF_ERR_ALIGN;
    const hf_nccl_allreduce_fn allreduce = rccl_allreduce();
    if (!allreduce) return HF_ERR_RCCL;
    HF_USE_DEVICE(ctx);
    tic(ctx);
    // ncclUint64 = 5, ncclSum = 0 (rccl.h: ncclDataType_t, ncclRedOp_t)
    const int rc = allreduce(d_freq256, d_freq256, 256, 5, 0, nccl_comm, ctx->stream);
    toc(ctx);
    if (rc != 0) {
        ctx->last_hip = rc;
        return HF_ERR_RCCL;
    }
    return HF_OK;
}

int hf_symbol_stat(hf_ctx* ctx, const void* d_data, uint64_t n, int layout, int N, int wt_width,
                   uint64_t* h_freq) {
    if (!ctx || !h_freq || !layout_ok(layout) || (n && !d_data)) return HF_ERR_ARG;
    if (N < 2 || N > 256 || !is_pow2(N) || wt_width < 1 || wt_width > 64) return HF_ERR_ARG;
    HF_USE_DEVICE(ctx);
    HF_HIP(hipMemsetAsync(ctx->d_freq, 0, 256 * sizeof(unsigned long long), ctx->stream));
    if (int rc = hf_symbol_stat_dev(ctx, d_data, n, layout,
                                    reinterpret_cast<uint64_t*>(ctx->d_freq)))
        return rc;
    HF_HIP(hipMemcpyAsync(ctx->h_freq, ctx->d_freq, 256 * sizeof(unsigned long long),
                          hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(hipStreamSynchronize(ctx->stream));
    // the RTL indexes its N counters with the low log2(N) bits of the symbol
    // (symbol_stat.scala:36-37) and its W-bit adds wrap
    const uint64_t m = mask_w(wt_width);
    for (int j = 0; j < N; ++j) h_freq[j] = 0;
    for (int s = 0; s < 256; ++s) h_freq[s & (N - 1)] = (h_freq[s & (N - 1)] + ctx->h_freq[s]) & m;
    return HF_OK;
}

/* ---- sub-field front ends ------------------------------------------------------------------- */
uint64_t hf_fields_side_bytes(uint64_t n, int layout) { return hf_fields_side_bytes_impl(n, layout); }

int hf_fields_split(hf_ctx* ctx, const void* d_data, uint64_t n, int layout, uint8_t* d_sym,
                    uint8_t* d_side) {
    if (!ctx || !front_end_layout(layout) || (n && (!d_data || !d_sym))) return HF_ERR_ARG;
    if (field_layout(layout) ? (n && !d_side) : (n % 8 != 0)) return HF_ERR_ARG;
    if (!aligned16(d_data) || !aligned16(d_sym) || !aligned16(d_side)) return HF_ERR_ALIGN;
    if (n > (1ull << 40)) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    tic(ctx);
    hf_launch_fields_split(d_data, n, layout, d_sym, d_side, ctx->stream);
    const hipError_t launch_err = hipGetLastError();
    toc(ctx);
    HF_HIP(launch_err);
    return HF_OK;
}

int hf_fields_merge(hf_ctx* ctx, const uint8_t* d_sym, const uint8_t* d_side, uint64_t n, int layout,
                    void* d_out) {
    if (!ctx || !front_end_layout(layout) || (n && (!d_out || !d_sym))) return HF_ERR_ARG;
    if (field_layout(layout) ? (n && !d_side) : (n % 8 != 0)) return HF_ERR_ARG;
    if (!aligned16(d_out) || !aligned16(d_sym) || !aligned16(d_side)) return HF_ERR_ALIGN;
    if (n > (1ull << 40)) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    tic(ctx);
    hf_launch_fields_merge(d_sym, d_side, n, layout, d_out, ctx->stream);
    const hipError_t launch_err = hipGetLastError();
    toc(ctx);
    HF_HIP(launch_err);
    return HF_OK;
}

/* ---- SymbolSort ---------------------------------------------------------------------------- */
int hf_symbol_sort(hf_ctx* ctx, const uint64_t* h_freq, int N, uint64_t* h_sorted,
                   uint16_t* h_rank2sym) {
    if (!ctx || !h_freq || !h_sorted || N < 2 || N > 256) return HF_ERR_ARG;
    for (int i = 0; i < N; ++i)
        if (h_freq[i] >> 56) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    memcpy(ctx->h_io->freq, h_freq, sizeof(uint64_t) * (size_t)N);
    if (int rc = run_stage(ctx, N, HF_OP_SORT, 0, 0)) return rc;
    memcpy(h_sorted, ctx->h_io->sorted, sizeof(uint64_t) * (size_t)N);
    if (h_rank2sym) memcpy(h_rank2sym, ctx->h_io->rank2sym, sizeof(uint16_t) * (size_t)N);
    return HF_OK;
}

/* ---- ShannonEntropy + AdaptiveThreshold ------------------------------------------------------ */
int hf_entropy(hf_ctx* ctx, const uint64_t* h_freq, int N, uint32_t* entropy_q16,
               int* compression_mode) {
    if (!ctx || !h_freq || !entropy_q16 || N < 2 || N > 256) return HF_ERR_ARG;
    uint64_t tot = 0;
    (void)tot;
    for (int i = 0; i < N; ++i) {
        if (h_freq[i] >> 56) return HF_ERR_RANGE;
        tot += h_freq[i];
    }
    HF_USE_DEVICE(ctx);
    memcpy(ctx->h_io->freq, h_freq, sizeof(uint64_t) * (size_t)N);
    if (int rc = run_stage(ctx, N, HF_OP_ENTROPY, 0, 0)) return rc;
    *entropy_q16 = ctx->h_io->entropy_q16;
    if (compression_mode) *compression_mode = (int)ctx->h_io->mode;
    return HF_OK;
}

int hf_adaptive_threshold(uint32_t entropy_q16, int depth, int wt_width, hf_adaptive_config* out) {
    if (!out || depth < 2 || depth > 256 || wt_width < 2 || wt_width > 64) return HF_ERR_ARG;
    // (1 << (wtWidth - 2)) is a Scala Int: the shift count is taken mod 32, and a negative
    // result cannot become a UInt literal (adaptive_threshold.scala:53 fails to elaborate)
    const int sh = (wt_width - 2) & 31;
    if (sh == 31) return HF_ERR_RANGE;
    const uint32_t dmask = (1u << log2ceil_i(depth)) - 1u;
    memset(out, 0, sizeof(*out));
    if (entropy_q16 < 0x8000u) {  // setLowEntropyConfig, :41-46
        out->encoding_mode = 1;
        out->compression_threshold_q16 = 19660u;  // (0.3 * 65536).toInt
        out->tree_depth_limit = (uint32_t)(depth / 4) & dmask;
        out->min_freq_threshold = 1;
    } else if (entropy_q16 > 0x60000u) {  // setHighEntropyConfig, :49-54
        out->encoding_mode = 0;
        out->compression_threshold_q16 = 58982u;  // (0.9 * 65536).toInt
        out->tree_depth_limit = (uint32_t)(depth / 8) & dmask;
        out->min_freq_threshold = (1ull << sh) & mask_w(wt_width);
    } else {  // setMediumEntropyConfig, :57-62
        out->encoding_mode = 1;
        out->compression_threshold_q16 = 39321u;  // (0.6 * 65536).toInt
        out->tree_depth_limit = (uint32_t)(depth / 2) & dmask;
        out->min_freq_threshold = 2ull & mask_w(wt_width);
    }
    return HF_OK;
}

/* ---- HuffmanTreeBuilder ---------------------------------------------------------------------- */
int hf_tree_build(hf_ctx* ctx, const uint64_t* h_freqs, int N, int wt_width, uint64_t* h_weights,
                  uint16_t* h_lefts, uint16_t* h_rights, uint16_t* h_root, uint16_t* h_depths) {
    if (!ctx || !h_freqs || N < 2 || N > 256) return HF_ERR_ARG;
    if (wt_width != 32 && wt_width != 64) return HF_ERR_ARG;
    // the merge order equals the RTL's only while no W-bit sum wraps and no valid weight
    // reaches the all-ones mask value (tree_build.scala:48-52)
    const uint64_t lim = wt_width == 32 ? 0xFFFFFFFFull : (1ull << 56);
    uint64_t tot = 0;
    for (int i = 0; i < N; ++i) {
        if (h_freqs[i] >= lim) return HF_ERR_RANGE;
        tot += h_freqs[i];
        if (tot >= lim) return HF_ERR_RANGE;
    }
    HF_USE_DEVICE(ctx);
    memcpy(ctx->h_io->freq, h_freqs, sizeof(uint64_t) * (size_t)N);
    if (int rc = run_stage(ctx, N, HF_OP_TREE, 0, 0)) return rc;
    const int M = 2 * N - 1;
    if (h_weights) memcpy(h_weights, ctx->h_io->weights, sizeof(uint64_t) * (size_t)M);
    if (h_lefts) memcpy(h_lefts, ctx->h_io->lefts, sizeof(uint16_t) * (size_t)M);
    if (h_rights) memcpy(h_rights, ctx->h_io->rights, sizeof(uint16_t) * (size_t)M);
    if (h_root) *h_root = (uint16_t)ctx->h_io->root;
    if (h_depths) memcpy(h_depths, ctx->h_io->depths, sizeof(uint16_t) * (size_t)N);
    return HF_OK;
}

/* ---- tree_adjust ----------------------------------------------------------------------------- */
int hf_tree_adjust(hf_ctx* ctx, const uint16_t* h_depths_in, int N, int max_depth,
                   uint16_t* h_depths_out) {
    if (!ctx || !h_depths_in || !h_depths_out || N < 2 || N > 256) return HF_ERR_ARG;
    if (max_depth < 1 || max_depth > HF_MAX_CODE_LEN || (1 << max_depth) < N) return HF_ERR_ARG;
    HF_USE_DEVICE(ctx);
    memcpy(ctx->h_io->depths, h_depths_in, sizeof(uint16_t) * (size_t)N);
    if (int rc = run_stage(ctx, N, HF_OP_ADJUST, max_depth, 0)) return rc;
    memcpy(h_depths_out, ctx->h_io->adj, sizeof(uint16_t) * (size_t)N);
    return HF_OK;
}

/* ---- EntropyAwareTreeBuilder ------------------------------------------------------------------- */
int hf_codebook(hf_ctx* ctx, const uint64_t* h_freq, int N, const hf_entropy_config* cfg,
                uint16_t* h_code, uint8_t* h_len) {
    if (!ctx || !h_freq || !h_code || !h_len || N < 2 || N > 256) return HF_ERR_ARG;
    int lmax = HF_MAX_CODE_LEN;
    unsigned long long thr = 0;
    if (cfg) {
        if (cfg->tree_depth_limit) lmax = (int)cfg->tree_depth_limit;
        thr = cfg->min_freq_threshold;
    }
    if (lmax < 1 || lmax > HF_MAX_CODE_LEN || (1 << lmax) < N) return HF_ERR_ARG;
    uint64_t tot = 0;
    for (int i = 0; i < N; ++i) {
        if (h_freq[i] >> 56) return HF_ERR_RANGE;
        tot += h_freq[i];
        if (tot >> 56) return HF_ERR_RANGE;
    }
    HF_USE_DEVICE(ctx);
    memcpy(ctx->h_io->freq, h_freq, sizeof(uint64_t) * (size_t)N);
    if (int rc = run_stage(ctx, N, HF_OP_CODEBOOK, lmax, thr)) return rc;
    memcpy(h_code, ctx->h_io->code, sizeof(uint16_t) * (size_t)N);
    memcpy(h_len, ctx->h_io->len, (size_t)N);
    return HF_OK;
}

/* ---- front half of EntropyAwareHuffmanSystem, on the device ------------------------------------ */
int hf_system_codebook_dev(hf_ctx* ctx, const uint64_t* d_freq256, int mode,
                           uint64_t min_freq_threshold, void* d_codebook) {
    if (!ctx || !d_freq256 || !d_codebook) return HF_ERR_ARG;
    if (mode != HF_MODE_BYPASS && mode != HF_MODE_HUFFMAN && mode != HF_MODE_AUTO &&
        mode != HF_MODE_BEST)
        return HF_ERR_ARG;
    if (!aligned16(d_freq256) || !aligned16(d_codebook)) return HF_ERR_ALIGN;
    HF_USE_DEVICE(ctx);
    hf_cb_dev* cb = reinterpret_cast<hf_cb_dev*>(d_codebook);
    tic(ctx);
    hf_launch_system(reinterpret_cast<const unsigned long long*>(d_freq256), mode,
                     min_freq_threshold, cb, ctx->stream);  // codebook and decode tables
    const hipError_t launch_err = hipGetLastError();  // before the event record can mask it
    toc(ctx);
    HF_HIP(launch_err);
    return HF_OK;
}

int hf_system_codebook_block_dev(hf_ctx* ctx, const void* d_data, uint64_t n, int layout, int mode,
                                 uint64_t min_freq_threshold, uint64_t* d_freq256,
                                 void* d_codebook) {
    if (!ctx || !d_freq256 || !d_codebook || !layout_ok(layout) || (n && !d_data))
        return HF_ERR_ARG;
    if (mode != HF_MODE_BYPASS && mode != HF_MODE_HUFFMAN && mode != HF_MODE_AUTO &&
        mode != HF_MODE_BEST)
        return HF_ERR_ARG;
    if (!aligned16(d_data) || !aligned16(d_freq256) || !aligned16(d_codebook)) return HF_ERR_ALIGN;
    if (n > HF_BLOCK_MAX_SYMS) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    tic(ctx);
    hf_launch_system_block(d_data, (uint32_t)n, norm_layout(layout), mode, min_freq_threshold,
                           reinterpret_cast<unsigned long long*>(d_freq256),
                           reinterpret_cast<hf_cb_dev*>(d_codebook), ctx->stream);
    const hipError_t launch_err = hipGetLastError();
    toc(ctx);
    HF_HIP(launch_err);
    return HF_OK;
}

/* ---- MemWrapper ---------------------------------------------------------------------------------- */
int hf_codebook_upload(hf_ctx* ctx, const uint16_t* h_code, const uint8_t* h_len, int mode,
                       void* d_codebook) {
    if (!ctx || !h_code || !h_len || !d_codebook) return HF_ERR_ARG;
    if (mode != HF_MODE_BYPASS && mode != HF_MODE_HUFFMAN) return HF_ERR_ARG;
    if (!aligned16(d_codebook)) return HF_ERR_ALIGN;
    for (int s = 0; s < 256; ++s)
        if (h_len[s] > HF_MAX_CODE_LEN) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    hf_cb_dev* h = ctx->h_cb;
    memset(h, 0, sizeof(hf_cb_dev));
    for (int s = 0; s < 256; ++s) {
        const uint32_t len = mode == HF_MODE_BYPASS ? 8u : h_len[s];
        const uint32_t code = mode == HF_MODE_BYPASS ? (uint32_t)s : h_code[s];
        h->enc[s] = (code & (len ? ((1u << len) - 1u) : 0u)) | (len << 16);
        h->code[s] = (uint16_t)(h->enc[s] & 0xFFFFu);
        h->len[s] = (uint8_t)len;
    }
    h->mode = (uint32_t)mode;
    hf_cb_dev* cb = reinterpret_cast<hf_cb_dev*>(d_codebook);
    HF_HIP(hipMemcpyAsync(cb, h, sizeof(hf_cb_dev), hipMemcpyHostToDevice, ctx->stream));
    hf_launch_tables(cb, ctx->stream);
    HF_HIP(hipGetLastError());
    HF_HIP(hipStreamSynchronize(ctx->stream));
    return HF_OK;
}

int hf_codebook_download(hf_ctx* ctx, const void* d_codebook, uint16_t* h_code, uint8_t* h_len,
                         int* mode, uint32_t* entropy_q16) {
    if (!ctx || !d_codebook) return HF_ERR_ARG;
    HF_USE_DEVICE(ctx);
    HF_HIP(hipMemcpyAsync(ctx->h_cb, d_codebook, sizeof(hf_cb_dev), hipMemcpyDeviceToHost,
                          ctx->stream));
    HF_HIP(hipStreamSynchronize(ctx->stream));
    if (h_code) memcpy(h_code, ctx->h_cb->code, sizeof(uint16_t) * 256);
    if (h_len) memcpy(h_len, ctx->h_cb->len, 256);
    if (mode) *mode = (int)ctx->h_cb->mode;
    if (entropy_q16) *entropy_q16 = ctx->h_cb->entropy_q16;
    return HF_OK;
}

/* ---- encoder ------------------------------------------------------------------------------------- */
int hf_encode_pairs(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                    const void* d_codebook, hf_code_beat* d_beats) {
    if (!ctx || !d_codebook || !layout_ok(layout) || (n && (!d_data || !d_beats)))
        return HF_ERR_ARG;
    if (!aligned16(d_data) || !aligned16(d_beats) || !aligned16(d_codebook)) return HF_ERR_ALIGN;
    HF_USE_DEVICE(ctx);
    hf_launch_pairs(d_data, n, norm_layout(layout), reinterpret_cast<const hf_cb_dev*>(d_codebook),
                    d_beats, ctx->stream);
    HF_HIP(hipGetLastError());
    return HF_OK;
}

uint64_t hf_num_chunks(uint64_t n) { return (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS; }

uint64_t hf_payload_bound_words(uint64_t n) { return hf_num_chunks(n) * 64ull * 129ull; }

static int encode_common(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                         const void* d_codebook, bool per_chunk, uint8_t* d_side,
                         uint32_t* d_payload, uint64_t payload_cap_words, uint32_t* d_chunk_off,
                         uint64_t* h_total_words) {
    if (!ctx || !d_codebook || !d_chunk_off || !layout_ok(layout)) return HF_ERR_ARG;
    if (n && (!d_data || !d_payload)) return HF_ERR_ARG;
    const int lay = norm_layout(layout);
    if (lay == HF_LAYOUT_BF16_EXP && n && !d_side) return HF_ERR_ARG;
    if (!aligned16(d_data) || !aligned16(d_side) || !aligned16(d_payload) ||
        !aligned16(d_chunk_off) || !aligned16(d_codebook))
        return HF_ERR_ALIGN;
    if (n > (1ull << 40)) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    const uint64_t nchunks = hf_num_chunks(n);
    if (int rc = ensure_desc(ctx, nchunks)) return rc;
    if (nchunks)
        if (int rc = ensure_spill(ctx)) return rc;
    // (cleared by a kernel of the library's own, not by hipMemsetAsync: under stream capture a
    // memset becomes a graph node of its own kind, and ROCm 7.2 replayed these small ones with a
    // stale fill pattern from the second replay on, which left the ticket full of garbage)
    if (nchunks == 0) {
        zero_words(d_chunk_off, 1, ctx->stream);
    } else {
        // the ticket and the tile descriptors (the error word in d_ctl is sticky and stays); a
        // stream of a single tile uses neither, and a small block is spared the launch
        if (!hf_encode_single_tile(nchunks))
            zero_words(reinterpret_cast<uint32_t*>(ctx->d_desc), (2 + ((nchunks + 1) & ~1ull)) * 2,
                       ctx->stream);
        unsigned long long* const desc = ctx->d_desc + 2;
        uint32_t* const ticket = reinterpret_cast<uint32_t*>(ctx->d_desc);
        tic(ctx);
        if (per_chunk)
            hf_launch_encode_chunks(d_data, n, lay,
                                    reinterpret_cast<const hf_chunk_cb*>(d_codebook), d_side,
                                    d_payload, payload_cap_words, d_chunk_off, desc, ticket,
                                    ctx->d_spill, ctx->d_ctl + 1, ctx->num_cu, ctx->stream);
        else
            hf_launch_encode(d_data, n, lay, reinterpret_cast<const hf_cb_dev*>(d_codebook), d_side,
                             d_payload, payload_cap_words, d_chunk_off, desc, ticket, ctx->d_spill,
                             ctx->d_ctl + 1, ctx->num_cu, ctx->stream);
        const hipError_t launch_err = hipGetLastError();  // before the event record can mask it
        toc(ctx);
        HF_HIP(launch_err);
    }
    if (h_total_words) {
        // error word and total (the kernel leaves the total in d_ctl[2] too) in one copy
        HF_HIP(hipMemcpyAsync(ctx->h_ctl, ctx->d_ctl, 64, hipMemcpyDeviceToHost, ctx->stream));
        HF_HIP(hipStreamSynchronize(ctx->stream));
        *h_total_words = nchunks ? ctx->h_ctl[2] : 0;
        return report_kernel_errors(ctx);
    }
    return HF_OK;
}

int hf_encode(hf_ctx* ctx, const void* d_data, uint64_t n, int layout, const void* d_codebook,
              uint8_t* d_side, uint32_t* d_payload, uint64_t payload_cap_words,
              uint32_t* d_chunk_off, uint64_t* h_total_words) {
    return encode_common(ctx, d_data, n, layout, d_codebook, false, d_side, d_payload,
                         payload_cap_words, d_chunk_off, h_total_words);
}

/* ---- per-block codebooks --------------------------------------------------------------------------- */
int hf_symbol_stat_chunks_dev(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                              uint32_t* d_freq_chunks) {
    if (!ctx || !layout_ok(layout) || (n && (!d_data || !d_freq_chunks))) return HF_ERR_ARG;
    if (!aligned16(d_data) || !aligned16(d_freq_chunks)) return HF_ERR_ALIGN;
    if (n > (1ull << 40)) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    tic(ctx);
    hf_launch_hist_chunks(d_data, n, norm_layout(layout), d_freq_chunks, ctx->stream);
    const hipError_t launch_err = hipGetLastError();  // before the event record can mask it
    toc(ctx);
    HF_HIP(launch_err);
    return HF_OK;
}

int hf_symbol_stat_pages_dev(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                             uint32_t* d_freq_chunks, uint64_t* d_freq256) {
    if (!ctx || !layout_ok(layout) || (n && (!d_data || !d_freq_chunks))) return HF_ERR_ARG;
    if (!aligned16(d_data) || !aligned16(d_freq_chunks) || !aligned16(d_freq256)) return HF_ERR_ALIGN;
    if (n > (1ull << 40)) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    tic(ctx);
    hf_launch_hist_pages(d_data, n, norm_layout(layout), d_freq_chunks,
                         reinterpret_cast<unsigned long long*>(d_freq256), ctx->num_cu, ctx->stream);
    const hipError_t launch_err = hipGetLastError();  // before the event record can mask it
    toc(ctx);
    HF_HIP(launch_err);
    return HF_OK;
}

int hf_system_codebook_chunks_dev(hf_ctx* ctx, const uint32_t* d_freq_chunks, uint64_t nchunks,
                                  int mode, uint64_t min_freq_threshold, void* d_chunk_codebooks) {
    if (!ctx || (nchunks && (!d_freq_chunks || !d_chunk_codebooks))) return HF_ERR_ARG;
    if (mode != HF_MODE_BYPASS && mode != HF_MODE_HUFFMAN && mode != HF_MODE_AUTO &&
        mode != HF_MODE_BEST)
        return HF_ERR_ARG;
    if (!aligned16(d_freq_chunks) || !aligned16(d_chunk_codebooks)) return HF_ERR_ALIGN;
    if (nchunks > 0x7FFFFFFFull) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    if (nchunks)
        if (int rc = ensure_cbscratch(ctx)) return rc;
    tic(ctx);
    hf_launch_system_chunks(d_freq_chunks, nchunks, mode, min_freq_threshold,
                            reinterpret_cast<hf_chunk_cb*>(d_chunk_codebooks), ctx->d_cbscratch,
                            HF_CB_BATCH, ctx->stream);
    const hipError_t launch_err = hipGetLastError();  // before the event record can mask it
    toc(ctx);
    HF_HIP(launch_err);
    return HF_OK;
}

// The same kernels with the batch scratch in the static device array: nothing is allocated or
// freed on this path (hf_ctx_create fetched the array's address for the context's device).  The
// kernels overwrite every scratch word they later read, so nothing has to be cleared between
// calls.  A context works in a slice of its own, claimed here on its first call and released by
// hf_ctx_destroy (which drains the stream first): calls of one context are ordered by its stream,
// calls of different contexts never touch the same bytes.
static int claim_static_slice(hf_ctx* ctx) {
    if (ctx->static_slot >= 0) return HF_OK;
    if (ctx->device >= kMaxDevices) return HF_ERR_NOMEM;
    std::atomic<unsigned>& held = g_static_slices[ctx->device];
    unsigned cur = held.load(std::memory_order_acquire);
    for (;;) {
        int slot = -1;
        for (unsigned i = 0; i < HF_STATIC_CB_SLOTS; ++i)
            if (!(cur >> i & 1u)) {
                slot = (int)i;
                break;
            }
        if (slot < 0) return HF_ERR_NOMEM;  // every slice of this GPU is held by a live context
        if (held.compare_exchange_weak(cur, cur | (1u << slot), std::memory_order_acq_rel)) {
            ctx->static_slot = slot;
            return HF_OK;
        }
    }
}

int hf_system_codebook_chunks_static(hf_ctx* ctx, const uint32_t* d_freq_chunks, uint64_t nchunks,
                                     int mode, uint64_t min_freq_threshold,
                                     void* d_chunk_codebooks) {
    if (!ctx || (nchunks && (!d_freq_chunks || !d_chunk_codebooks))) return HF_ERR_ARG;
    if (mode != HF_MODE_BYPASS && mode != HF_MODE_HUFFMAN && mode != HF_MODE_AUTO &&
        mode != HF_MODE_BEST)
        return HF_ERR_ARG;
    if (!aligned16(d_freq_chunks) || !aligned16(d_chunk_codebooks)) return HF_ERR_ALIGN;
    if (nchunks > HF_STATIC_CB_MAX_CHUNKS) return HF_ERR_RANGE;  // more than the block holds
    if (!ctx->d_static_cbscratch) return HF_ERR_HIP;
    if (nchunks == 0) return HF_OK;
    if (int rc = claim_static_slice(ctx)) return rc;
    HF_USE_DEVICE(ctx);
    tic(ctx);
    hf_launch_system_chunks(d_freq_chunks, nchunks, mode, min_freq_threshold,
                            reinterpret_cast<hf_chunk_cb*>(d_chunk_codebooks),
                            static_cast<unsigned char*>(ctx->d_static_cbscratch) +
                                (size_t)ctx->static_slot * kStaticSliceBytes,
                            HF_STATIC_CB_MAX_CHUNKS, ctx->stream);
    const hipError_t launch_err = hipGetLastError();
    toc(ctx);
    HF_HIP(launch_err);
    return HF_OK;
}

int hf_encode_chunks(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                     const void* d_chunk_codebooks, uint8_t* d_side, uint32_t* d_payload,
                     uint64_t payload_cap_words, uint32_t* d_chunk_off, uint64_t* h_total_words) {
    return encode_common(ctx, d_data, n, layout, d_chunk_codebooks, true, d_side, d_payload,
                         payload_cap_words, d_chunk_off, h_total_words);
}

static int decode_chunks_common(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                                const uint32_t* d_chunk_off, const uint32_t* d_chunk_words,
                                const uint8_t* d_side, uint64_t n,
                                int layout, const void* d_chunk_codebooks,
                                const uint32_t* d_chunk_ids, const uint32_t* d_out_slots,
                                uint64_t n_slots, uint64_t n_ids, void* d_out, int check) {
    if (!ctx || !layout_ok(layout)) return HF_ERR_ARG;
    if (d_out_slots && !d_chunk_ids) return HF_ERR_ARG;
    const uint64_t n_work = d_chunk_ids ? n_ids : hf_num_chunks(n);
    if (n_work && (!d_payload || !d_chunk_off || !d_out || !d_chunk_codebooks)) return HF_ERR_ARG;
    if (n_work && payload_words < 4) return HF_ERR_ARG;
    const int lay = norm_layout(layout);
    if (lay == HF_LAYOUT_BF16_EXP && n_work && !d_side) return HF_ERR_ARG;
    if (!aligned16(d_payload) || !aligned16(d_side) || !aligned16(d_out) ||
        !aligned16(d_chunk_off) || !aligned16(d_chunk_codebooks) || !aligned16(d_chunk_ids) ||
        !aligned16(d_out_slots) || !aligned16(d_chunk_words))
        return HF_ERR_ALIGN;
    if (n > (1ull << 40) || n_work > 0x7FFFFFFFull || n_slots > 0xFFFFFFFFull) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    if (n_work) {
        tic(ctx);
        hf_launch_decode_chunks(d_payload, payload_words, d_chunk_off, d_chunk_words, d_side, n, lay,
                                reinterpret_cast<const hf_chunk_cb*>(d_chunk_codebooks),
                                d_chunk_ids, d_out_slots, n_slots, n_ids, d_out, ctx->d_ctl + 1,
                                ctx->num_cu, ctx->stream);
        const hipError_t launch_err = hipGetLastError();  // before the event record can mask it
        toc(ctx);
        HF_HIP(launch_err);
    }
    if (check) {
        HF_HIP(hipMemcpyAsync(ctx->h_ctl, ctx->d_ctl, 64, hipMemcpyDeviceToHost, ctx->stream));
        HF_HIP(hipStreamSynchronize(ctx->stream));
        return report_kernel_errors(ctx);
    }
    return HF_OK;
}

int hf_decode_chunks(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                     const uint32_t* d_chunk_off, const uint8_t* d_side, uint64_t n, int layout,
                     const void* d_chunk_codebooks, const uint32_t* d_chunk_ids, uint64_t n_ids,
                     void* d_out, int check) {
    return decode_chunks_common(ctx, d_payload, payload_words, d_chunk_off, nullptr, d_side, n,
                                layout, d_chunk_codebooks, d_chunk_ids, nullptr, 0, n_ids, d_out,
                                check);
}

int hf_decode_chunks_to(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                        const uint32_t* d_chunk_off, const uint8_t* d_side, uint64_t n, int layout,
                        const void* d_chunk_codebooks, const uint32_t* d_chunk_ids,
                        const uint32_t* d_out_slots, uint64_t n_ids, void* d_out, uint64_t n_slots,
                        int check) {
    if (!d_chunk_ids || !d_out_slots) return HF_ERR_ARG;
    return decode_chunks_common(ctx, d_payload, payload_words, d_chunk_off, nullptr, d_side, n,
                                layout, d_chunk_codebooks, d_chunk_ids, d_out_slots, n_slots, n_ids,
                                d_out, check);
}

/* ---- slotted placement of a per-block stream --------------------------------------------------- */
// (a chunk of 16384 symbols has at most 16384 * 16 bits of codes)
uint64_t hf_slotted_bound_words(uint64_t n) {
    return hf_num_chunks(n) * (uint64_t)HF_SLOT_WORDS(HF_CHUNK_SYMS * 16u);
}

// d_codebook: per_chunk ? hf_chunk_cb records : one hf_cb_dev (both begin with their code table)
static int encode_slotted_common(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                                 const void* d_codebook, bool per_chunk,
                                 const uint32_t* d_freq_chunks, uint8_t* d_side,
                                 uint32_t* d_payload, uint64_t payload_cap_words,
                                 uint32_t* d_chunk_off, uint32_t* d_chunk_words,
                                 uint64_t* h_total_words) {
    if (!ctx || !d_chunk_off || !layout_ok(layout)) return HF_ERR_ARG;
    if (n && (!d_data || !d_payload || !d_codebook || !d_freq_chunks || !d_chunk_words))
        return HF_ERR_ARG;
    const int lay = norm_layout(layout);
    if (lay == HF_LAYOUT_BF16_EXP && n && !d_side) return HF_ERR_ARG;
    if (!aligned16(d_data) || !aligned16(d_side) || !aligned16(d_payload) ||
        !aligned16(d_chunk_off) || !aligned16(d_chunk_words) || !aligned16(d_codebook) ||
        !aligned16(d_freq_chunks))
        return HF_ERR_ALIGN;
    if (n > (1ull << 40)) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    const uint64_t nchunks = hf_num_chunks(n);
    if (int rc = ensure_scan_aux(ctx, nchunks)) return rc;
    uint32_t* const dump = reinterpret_cast<uint32_t*>(
        reinterpret_cast<unsigned char*>(ctx->d_scan_aux) + ctx->scan_aux_bytes);
    if (nchunks == 0) {
        zero_words(d_chunk_off, 1, ctx->stream);
    } else {
        tic(ctx);
        static_assert(offsetof(hf_chunk_cb, enc) == 0 && offsetof(hf_cb_dev, enc) == 0 &&
                          sizeof(hf_chunk_cb) % 4 == 0,
                      "a record's code table is its first kilobyte");
        hf_launch_encode_chunks_slotted(d_data, n, lay,
                                        reinterpret_cast<const uint32_t*>(d_codebook),
                                        per_chunk ? (uint32_t)(sizeof(hf_chunk_cb) / 4) : 0u,
                                        d_freq_chunks, d_side, d_payload, payload_cap_words,
                                        d_chunk_off, d_chunk_words, ctx->d_scan_aux, dump,
                                        ctx->d_ctl + 1, ctx->num_cu, ctx->stream);
        const hipError_t launch_err = hipGetLastError();
        toc(ctx);
        HF_HIP(launch_err);
    }
    if (h_total_words) {
        // error word and total (the scan leaves the last slot offset in d_ctl[2] too) in one copy
        HF_HIP(hipMemcpyAsync(ctx->h_ctl, ctx->d_ctl, 64, hipMemcpyDeviceToHost, ctx->stream));
        HF_HIP(hipStreamSynchronize(ctx->stream));
        *h_total_words = nchunks ? ctx->h_ctl[2] : 0;
        return report_kernel_errors(ctx);
    }
    return HF_OK;
}

int hf_encode_chunks_slotted(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                             const void* d_chunk_codebooks, const uint32_t* d_freq_chunks,
                             uint8_t* d_side, uint32_t* d_payload, uint64_t payload_cap_words,
                             uint32_t* d_chunk_off, uint32_t* d_chunk_words,
                             uint64_t* h_total_words) {
    return encode_slotted_common(ctx, d_data, n, layout, d_chunk_codebooks, true, d_freq_chunks,
                                 d_side, d_payload, payload_cap_words, d_chunk_off, d_chunk_words,
                                 h_total_words);
}

int hf_encode_slotted(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                      const void* d_codebook, const uint32_t* d_freq_chunks, uint8_t* d_side,
                      uint32_t* d_payload, uint64_t payload_cap_words, uint32_t* d_chunk_off,
                      uint32_t* d_chunk_words, uint64_t* h_total_words) {
    return encode_slotted_common(ctx, d_data, n, layout, d_codebook, false, d_freq_chunks, d_side,
                                 d_payload, payload_cap_words, d_chunk_off, d_chunk_words,
                                 h_total_words);
}

int hf_symbol_stat_fold_dev(hf_ctx* ctx, const uint32_t* d_freq_chunks, uint64_t nchunks,
                            uint64_t* d_freq256) {
    if (!ctx || (nchunks && (!d_freq_chunks || !d_freq256))) return HF_ERR_ARG;
    if (!aligned16(d_freq_chunks) || !aligned16(d_freq256)) return HF_ERR_ALIGN;
    if (nchunks > (1ull << 26)) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    if (nchunks) {
        tic(ctx);
        hf_launch_fold_chunks(d_freq_chunks, nchunks,
                              reinterpret_cast<unsigned long long*>(d_freq256), ctx->stream);
        const hipError_t launch_err = hipGetLastError();
        toc(ctx);
        HF_HIP(launch_err);
    }
    return HF_OK;
}

int hf_decode_chunks_slotted(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                             const uint32_t* d_chunk_off, const uint32_t* d_chunk_words,
                             const uint8_t* d_side, uint64_t n, int layout,
                             const void* d_chunk_codebooks, const uint32_t* d_chunk_ids,
                             const uint32_t* d_out_slots, uint64_t n_ids, void* d_out,
                             uint64_t n_slots, int check) {
    if (!d_chunk_words && (d_chunk_ids ? n_ids : hf_num_chunks(n))) return HF_ERR_ARG;
    return decode_chunks_common(ctx, d_payload, payload_words, d_chunk_off, d_chunk_words, d_side,
                                n, layout, d_chunk_codebooks, d_chunk_ids, d_out_slots, n_slots,
                                n_ids, d_out, check);
}

int hf_chunk_codebook_download(hf_ctx* ctx, const void* d_chunk_codebooks, uint64_t chunk,
                               uint16_t* h_code, uint8_t* h_len, int* mode, uint32_t* entropy_q16) {
    if (!ctx || !d_chunk_codebooks) return HF_ERR_ARG;
    HF_USE_DEVICE(ctx);
    static_assert(sizeof(hf_chunk_cb) <= HF_CODEBOOK_DEV_BYTES, "staging buffer too small");
    const hf_chunk_cb* src = reinterpret_cast<const hf_chunk_cb*>(d_chunk_codebooks) + chunk;
    HF_HIP(hipMemcpyAsync(ctx->h_cb, src, sizeof(hf_chunk_cb), hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(hipStreamSynchronize(ctx->stream));
    const hf_chunk_cb* h = reinterpret_cast<const hf_chunk_cb*>(ctx->h_cb);
    for (int s = 0; s < 256; ++s) {
        if (h_code) h_code[s] = (uint16_t)(h->enc[s] & 0xFFFFu);
        if (h_len) h_len[s] = (uint8_t)(h->enc[s] >> 16);
    }
    if (mode) *mode = (int)h->mode;
    if (entropy_q16) *entropy_q16 = h->entropy_q16;
    return HF_OK;
}

/* ---- paged stream (format version 2) ---------------------------------------------------------------- */
uint64_t hf_pages_bound_words(uint64_t n) { return hf_num_chunks(n) * (uint64_t)HF_PAGE_MAX_WORDS; }

static int encode_pages_common(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                               const void* d_chunk_codebooks, const void* d_codebook,
                               const uint32_t* d_freq_chunks, uint8_t* d_side, uint32_t* d_payload,
                               uint64_t payload_cap_words, uint32_t* d_page_off,
                               uint64_t* h_total_words) {
    const void* const cbp = d_codebook ? d_codebook : d_chunk_codebooks;
    if (!ctx || !d_page_off || !layout_ok(layout)) return HF_ERR_ARG;
    // (a stream with one codebook may come without per-page counts: its pages are then placed
    // while they are encoded)
    const bool once = d_codebook && !d_freq_chunks;
    if (n && (!d_data || !d_payload || !cbp || (!d_freq_chunks && !once))) return HF_ERR_ARG;
    const int lay = norm_layout(layout);
    if (lay == HF_LAYOUT_BF16_EXP && n && !d_side) return HF_ERR_ARG;
    if (!aligned16(d_data) || !aligned16(d_side) || !aligned16(d_payload) ||
        !aligned16(d_page_off) || !aligned16(cbp) || !aligned16(d_freq_chunks))
        return HF_ERR_ALIGN;
    if (n > (1ull << 40)) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    const uint64_t nchunks = hf_num_chunks(n);
    if (once) {
        if (int rc = ensure_desc(ctx, nchunks)) return rc;
    } else if (int rc = ensure_scan_aux(ctx, nchunks)) {
        return rc;
    }
    if (nchunks == 0) {
        zero_words(d_page_off, 1, ctx->stream);
    } else if (once) {
        tic(ctx);
        zero_words(reinterpret_cast<uint32_t*>(ctx->d_desc + 2), nchunks * 2, ctx->stream);
        hf_launch_encode_pages_once(d_data, n, lay, reinterpret_cast<const hf_cb_dev*>(d_codebook),
                                    d_side, d_payload, payload_cap_words, d_page_off, ctx->d_desc + 2,
                                    ctx->d_ctl + 1, ctx->stream);
        const hipError_t launch_err = hipGetLastError();
        toc(ctx);
        HF_HIP(launch_err);
    } else {
        tic(ctx);
        hf_launch_encode_pages(d_data, n, lay, reinterpret_cast<const hf_chunk_cb*>(d_chunk_codebooks),
                               reinterpret_cast<const hf_cb_dev*>(d_codebook), d_freq_chunks, d_side,
                               d_payload, payload_cap_words, d_page_off, ctx->d_scan_aux,
                               ctx->d_ctl + 1, ctx->stream);
        const hipError_t launch_err = hipGetLastError();
        toc(ctx);
        HF_HIP(launch_err);
    }
    if (h_total_words) {
        // error word and total (the scan leaves the last offset in d_ctl[2] too) in one copy
        HF_HIP(hipMemcpyAsync(ctx->h_ctl, ctx->d_ctl, 64, hipMemcpyDeviceToHost, ctx->stream));
        HF_HIP(hipStreamSynchronize(ctx->stream));
        *h_total_words = nchunks ? ctx->h_ctl[2] : 0;
        return report_kernel_errors(ctx);
    }
    return HF_OK;
}

int hf_encode_pages(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                    const void* d_chunk_codebooks, const uint32_t* d_freq_chunks, uint8_t* d_side,
                    uint32_t* d_payload, uint64_t payload_cap_words, uint32_t* d_page_off,
                    uint64_t* h_total_words) {
    if (n && !d_chunk_codebooks) return HF_ERR_ARG;
    return encode_pages_common(ctx, d_data, n, layout, d_chunk_codebooks, nullptr, d_freq_chunks,
                               d_side, d_payload, payload_cap_words, d_page_off, h_total_words);
}

int hf_encode_pages_shared(hf_ctx* ctx, const void* d_data, uint64_t n, int layout,
                           const void* d_codebook, const uint32_t* d_freq_chunks, uint8_t* d_side,
                           uint32_t* d_payload, uint64_t payload_cap_words, uint32_t* d_page_off,
                           uint64_t* h_total_words) {
    if (!d_codebook) return HF_ERR_ARG;
    return encode_pages_common(ctx, d_data, n, layout, nullptr, d_codebook, d_freq_chunks, d_side,
                               d_payload, payload_cap_words, d_page_off, h_total_words);
}

static int decode_pages_common(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                               const uint32_t* d_page_off, const uint8_t* d_side, uint64_t n,
                               int layout, const void* d_codebook, const uint32_t* d_page_ids,
                               const uint32_t* d_out_slots, uint64_t n_ids, void* d_out,
                               uint64_t n_slots, int check) {
    if (!ctx || !layout_ok(layout)) return HF_ERR_ARG;
    if (d_out_slots && !d_page_ids) return HF_ERR_ARG;
    const uint64_t n_work = d_page_ids ? n_ids : hf_num_chunks(n);
    if (n_work && (!d_payload || !d_page_off || !d_out)) return HF_ERR_ARG;
    const int lay = norm_layout(layout);
    if (lay == HF_LAYOUT_BF16_EXP && n_work && !d_side) return HF_ERR_ARG;
    if (!aligned16(d_payload) || !aligned16(d_side) || !aligned16(d_out) || !aligned16(d_page_off) ||
        !aligned16(d_page_ids) || !aligned16(d_out_slots) || !aligned16(d_codebook))
        return HF_ERR_ALIGN;
    if (n > (1ull << 40) || n_work > 0x7FFFFFFFull || n_slots > 0xFFFFFFFFull) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    if (n_work) {
        tic(ctx);
        hf_launch_decode_pages(d_payload, payload_words, d_page_off, d_side, n, lay,
                               reinterpret_cast<const hf_cb_dev*>(d_codebook), ctx->d_page_tab,
                               d_page_ids, d_out_slots, n_slots, n_ids, d_out, ctx->d_ctl + 1,
                               ctx->stream);
        const hipError_t launch_err = hipGetLastError();
        toc(ctx);
        HF_HIP(launch_err);
    }
    if (check) {
        HF_HIP(hipMemcpyAsync(ctx->h_ctl, ctx->d_ctl, 64, hipMemcpyDeviceToHost, ctx->stream));
        HF_HIP(hipStreamSynchronize(ctx->stream));
        return report_kernel_errors(ctx);
    }
    return HF_OK;
}

int hf_decode_pages(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                    const uint32_t* d_page_off, const uint8_t* d_side, uint64_t n, int layout,
                    const uint32_t* d_page_ids, const uint32_t* d_out_slots, uint64_t n_ids,
                    void* d_out, uint64_t n_slots, int check) {
    return decode_pages_common(ctx, d_payload, payload_words, d_page_off, d_side, n, layout, nullptr,
                               d_page_ids, d_out_slots, n_ids, d_out, n_slots, check);
}

int hf_decode_pages_shared(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                           const uint32_t* d_page_off, const uint8_t* d_side, uint64_t n, int layout,
                           const void* d_codebook, const uint32_t* d_page_ids,
                           const uint32_t* d_out_slots, uint64_t n_ids, void* d_out,
                           uint64_t n_slots, int check) {
    if (!d_codebook) return HF_ERR_ARG;
    return decode_pages_common(ctx, d_payload, payload_words, d_page_off, d_side, n, layout,
                               d_codebook, d_page_ids, d_out_slots, n_ids, d_out, n_slots, check);
}

// The record of a stored page, read back and expanded on the host: ranks the record lists, then
// the symbols it does not list in ascending order; lengths from the counts per length; canonical
// codes in rank order.
int hf_page_codebook_download(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                              const uint32_t* d_page_off, uint64_t page, uint16_t* h_code,
                              uint8_t* h_len, int* mode, uint32_t* entropy_q16) {
    if (!ctx || !d_payload || !d_page_off) return HF_ERR_ARG;
    HF_USE_DEVICE(ctx);
    uint32_t* w = reinterpret_cast<uint32_t*>(ctx->h_io);
    HF_HIP(hipMemcpyAsync(w, d_page_off + page, 8, hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(hipStreamSynchronize(ctx->stream));
    const uint64_t off0 = w[0], off1 = w[1];
    if (off1 < off0 || off1 > payload_words || off1 - off0 < HF_PAGE_HDR_WORDS + HF_PAGE_GAP_WORDS)
        return HF_ERR_CORRUPT;
    const uint64_t words = off1 - off0;
    const uint64_t take = words < HF_PAGE_HDR_WORDS + 64u ? words : HF_PAGE_HDR_WORDS + 64u;
    HF_HIP(hipMemcpyAsync(w, d_payload + off0, 4 * take, hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(hipStreamSynchronize(ctx->stream));
    const uint32_t head = w[0] & 0x1FFu, bits = w[0] >> 10;
    if ((w[0] >> 29) != 0 || head > 255u ||
        HF_PAGE_HDR_WORDS + (head + 3u) / 4u + HF_PAGE_GAP_WORDS + ((uint64_t)bits + 31u) / 32u != words)
        return HF_ERR_CORRUPT;
    uint8_t r2s[256];
    bool seen[256] = {false};
    const uint8_t* listed = reinterpret_cast<const uint8_t*>(w + HF_PAGE_HDR_WORDS);
    uint32_t p = 0;
    for (; p < head; ++p) {
        if (seen[listed[p]]) return HF_ERR_CORRUPT;
        seen[listed[p]] = true;
        r2s[p] = listed[p];
    }
    for (int s = 0; s < 256; ++s)
        if (!seen[s]) r2s[p++] = (uint8_t)s;
    uint32_t total = 0, kraft = 0, next = 0;
    uint32_t nlen[17] = {0};
    for (uint32_t d = 1; d <= 16; ++d) {
        nlen[d] = (w[2 + (d - 1) / 2] >> (16u * ((d - 1) & 1u))) & 0xFFFFu;
        total += nlen[d];
        kraft += nlen[d] << (16 - d);
    }
    if (total != 256u || kraft > 65536u) return HF_ERR_CORRUPT;
    p = 0;
    for (uint32_t d = 1; d <= 16; ++d)
        for (uint32_t k = 0; k < nlen[d]; ++k, ++p) {
            if (h_code) h_code[r2s[p]] = (uint16_t)(next >> (16 - d));
            if (h_len) h_len[r2s[p]] = (uint8_t)d;
            next += 1u << (16 - d);
        }
    if (mode) *mode = (int)((w[0] >> 9) & 1u);
    if (entropy_q16) *entropy_q16 = w[1];
    return HF_OK;
}

/* ---- decoder ------------------------------------------------------------------------------------- */
static int decode_common(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                         const uint32_t* d_chunk_off, const uint32_t* d_chunk_words,
                         const uint8_t* d_side, uint64_t n, int layout, const void* d_codebook,
                         void* d_out, int check) {
    if (!ctx || !d_codebook || !layout_ok(layout)) return HF_ERR_ARG;
    if (n && (!d_payload || !d_chunk_off || !d_out)) return HF_ERR_ARG;
    if (n && payload_words < 4) return HF_ERR_ARG;  // the kernel reads the buffer 16 bytes at a time
    const int lay = norm_layout(layout);
    if (lay == HF_LAYOUT_BF16_EXP && n && !d_side) return HF_ERR_ARG;
    if (!aligned16(d_payload) || !aligned16(d_side) || !aligned16(d_out) ||
        !aligned16(d_chunk_off) || !aligned16(d_codebook) || !aligned16(d_chunk_words))
        return HF_ERR_ALIGN;
    if (n > (1ull << 40)) return HF_ERR_RANGE;
    HF_USE_DEVICE(ctx);
    if (n) {
        tic(ctx);
        hf_launch_decode(d_payload, payload_words, d_chunk_off, d_chunk_words, d_side, n, lay,
                         reinterpret_cast<const hf_cb_dev*>(d_codebook), d_out, ctx->d_ctl + 1,
                         ctx->num_cu, ctx->stream);
        const hipError_t launch_err = hipGetLastError();  // before the event record can mask it
        toc(ctx);
        HF_HIP(launch_err);
    }
    if (check) {
        HF_HIP(hipMemcpyAsync(ctx->h_ctl, ctx->d_ctl, 64, hipMemcpyDeviceToHost, ctx->stream));
        HF_HIP(hipStreamSynchronize(ctx->stream));
        return report_kernel_errors(ctx);
    }
    return HF_OK;
}

int hf_decode(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
              const uint32_t* d_chunk_off, const uint8_t* d_side, uint64_t n, int layout,
              const void* d_codebook, void* d_out, int check) {
    return decode_common(ctx, d_payload, payload_words, d_chunk_off, nullptr, d_side, n, layout,
                         d_codebook, d_out, check);
}

int hf_decode_slotted(hf_ctx* ctx, const uint32_t* d_payload, uint64_t payload_words,
                      const uint32_t* d_chunk_off, const uint32_t* d_chunk_words,
                      const uint8_t* d_side, uint64_t n, int layout, const void* d_codebook,
                      void* d_out, int check) {
    if (n && !d_chunk_words) return HF_ERR_ARG;
    return decode_common(ctx, d_payload, payload_words, d_chunk_off, d_chunk_words, d_side, n,
                         layout, d_codebook, d_out, check);
}

/* ---- container ---------------------------------------------------------------------------------- */
namespace {

inline uint64_t up16(uint64_t x) { return (x + 15ull) & ~15ull; }

// section offsets of a container; false when the request is not representable
bool plan_container(uint64_t n, int layout, int kind, int placement, uint64_t payload_words,
                    hf_container_info* info) {
    if ((!layout_ok(layout) && !front_end_layout(layout)) ||
        (kind != HF_CODEBOOK_STREAM && kind != HF_CODEBOOK_PER_CHUNK))
        return false;
    if (placement != HF_PLACEMENT_DENSE && placement != HF_PLACEMENT_SLOTTED &&
        placement != HF_PLACEMENT_PAGED)
        return false;
    if (layout == HF_LAYOUT_BITPLANE8 && n % 8 != 0) return false;
    if (n > (1ull << 40) || payload_words > 0xFFFFFFFFull) return false;
    const uint64_t nchunks = hf_num_chunks(n);
    memset(info, 0, sizeof(*info));
    info->version = placement == HF_PLACEMENT_DENSE ? 1u : 2u;
    info->layout = (uint32_t)layout;
    info->codebook_kind = (uint32_t)kind;
    info->placement = (uint32_t)placement;
    info->chunk_syms = HF_CHUNK_SYMS;
    info->max_code_len = HF_MAX_CODE_LEN;
    info->n = n;
    info->nchunks = nchunks;
    info->payload_words = payload_words;
    uint64_t off = HF_CONTAINER_HEADER_BYTES;
    // (pages with records carry their codebooks; pages without are coded with the stream's one)
    if (placement != HF_PLACEMENT_PAGED || kind == HF_CODEBOOK_STREAM) {
        info->off_codebook = off;
        off += kind == HF_CODEBOOK_STREAM ? 768ull : nchunks * HF_CONTAINER_CHUNK_CODEBOOK_BYTES;
        off = up16(off);
    }
    info->off_chunk_off = off;
    off = up16(off + 4ull * (nchunks + 1));
    if (placement == HF_PLACEMENT_SLOTTED) {
        info->off_chunk_words = off;
        off = up16(off + 4ull * nchunks);
    }
    if (layout == HF_LAYOUT_BF16_EXP || side_bytes(n, layout) != 0) {
        info->off_side = off;
        off = up16(off + side_bytes(n, layout));
    }
    info->off_payload = off;
    const uint64_t pay = 4ull * payload_words;
    off += pay < 16 ? 16 : up16(pay);
    info->total_bytes = off;
    return true;
}

void put32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }
void put64(uint8_t* p, uint64_t v) { memcpy(p, &v, 8); }
uint32_t get32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
uint64_t get64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

}  // namespace

int hf_container_plan(uint64_t n, int layout, int codebook_kind, uint64_t payload_words,
                      hf_container_info* info) {
    return hf_container_plan_placed(n, layout, codebook_kind, HF_PLACEMENT_DENSE, payload_words,
                                    info);
}

int hf_container_plan_placed(uint64_t n, int layout, int codebook_kind, int placement,
                             uint64_t payload_words, hf_container_info* info) {
    if (!info) return HF_ERR_ARG;
    return plan_container(n, layout, codebook_kind, placement, payload_words, info) ? HF_OK
                                                                                    : HF_ERR_ARG;
}

int hf_container_pack(hf_ctx* ctx, hf_container_info* info, const void* d_codebook,
                      const uint32_t* d_chunk_off, const uint8_t* d_side,
                      const uint32_t* d_payload, void* d_blob, uint64_t blob_cap_bytes) {
    return hf_container_pack_placed(ctx, info, d_codebook, d_chunk_off, nullptr, d_side, d_payload,
                                    d_blob, blob_cap_bytes);
}

int hf_container_pack_placed(hf_ctx* ctx, hf_container_info* info, const void* d_codebook,
                             const uint32_t* d_chunk_off, const uint32_t* d_chunk_words,
                             const uint8_t* d_side, const uint32_t* d_payload, void* d_blob,
                             uint64_t blob_cap_bytes) {
    if (!ctx || !info || !d_chunk_off || !d_blob) return HF_ERR_ARG;
    hf_container_info pl;
    if (!plan_container(info->n, (int)info->layout, (int)info->codebook_kind, (int)info->placement,
                        info->payload_words, &pl))
        return HF_ERR_ARG;
    if (pl.off_codebook && !d_codebook) return HF_ERR_ARG;
    if (pl.payload_words && !d_payload) return HF_ERR_ARG;
    if (pl.off_side && pl.n && !d_side) return HF_ERR_ARG;
    if (pl.off_chunk_words && pl.nchunks && !d_chunk_words) return HF_ERR_ARG;
    if (!aligned16(d_blob) || !aligned16(d_codebook)) return HF_ERR_ALIGN;
    if (blob_cap_bytes < pl.total_bytes) return HF_ERR_OVERFLOW;
    HF_USE_DEVICE(ctx);
    uint8_t* blob = reinterpret_cast<uint8_t*>(d_blob);
    hipStream_t s = ctx->stream;
    tic(ctx);
    // a section, then zeros up to the next one
    uint64_t end = HF_CONTAINER_HEADER_BYTES;
    auto pad_to = [&](uint64_t next) -> hipError_t {
        return next > end ? hipMemsetAsync(blob + end, 0, next - end, s) : hipSuccess;
    };
    if (pl.off_codebook) {
        if (pl.codebook_kind == HF_CODEBOOK_STREAM) {
            const hf_cb_dev* cb = reinterpret_cast<const hf_cb_dev*>(d_codebook);
            HF_HIP(hipMemcpyAsync(blob + pl.off_codebook, cb->code, 512, hipMemcpyDeviceToDevice, s));
            HF_HIP(hipMemcpyAsync(blob + pl.off_codebook + 512, cb->len, 256, hipMemcpyDeviceToDevice, s));
            HF_HIP(hipMemcpyAsync(ctx->h_ctl, &cb->mode, 8, hipMemcpyDeviceToHost, s));  // mode, entropy
            end = pl.off_codebook + 768;
        } else {
            hf_launch_cb_to_wire(reinterpret_cast<const hf_chunk_cb*>(d_codebook), pl.nchunks,
                                 reinterpret_cast<hf_wire_chunk_cb*>(blob + pl.off_codebook), s);
            HF_HIP(hipGetLastError());
            end = pl.off_codebook + pl.nchunks * HF_CONTAINER_CHUNK_CODEBOOK_BYTES;
        }
    }
    HF_HIP(pad_to(pl.off_chunk_off));
    HF_HIP(hipMemcpyAsync(blob + pl.off_chunk_off, d_chunk_off, 4ull * (pl.nchunks + 1),
                          hipMemcpyDeviceToDevice, s));
    end = pl.off_chunk_off + 4ull * (pl.nchunks + 1);
    if (pl.off_chunk_words) {
        HF_HIP(pad_to(pl.off_chunk_words));
        if (pl.nchunks)
            HF_HIP(hipMemcpyAsync(blob + pl.off_chunk_words, d_chunk_words, 4ull * pl.nchunks,
                                  hipMemcpyDeviceToDevice, s));
        end = pl.off_chunk_words + 4ull * pl.nchunks;
    }
    if (pl.off_side) {
        HF_HIP(pad_to(pl.off_side));
        const uint64_t sb = side_bytes(pl.n, (int)pl.layout);
        if (sb) HF_HIP(hipMemcpyAsync(blob + pl.off_side, d_side, sb, hipMemcpyDeviceToDevice, s));
        end = pl.off_side + sb;
    }
    HF_HIP(pad_to(pl.off_payload));
    if (pl.payload_words)
        HF_HIP(hipMemcpyAsync(blob + pl.off_payload, d_payload, 4ull * pl.payload_words,
                              hipMemcpyDeviceToDevice, s));
    end = pl.off_payload + 4ull * pl.payload_words;
    HF_HIP(pad_to(pl.total_bytes));
    toc(ctx);
    HF_HIP(hipStreamSynchronize(s));
    if (pl.off_codebook && pl.codebook_kind == HF_CODEBOOK_STREAM) {
        pl.mode = ctx->h_ctl[0];
        pl.entropy_q16 = ctx->h_ctl[1];
    } else {
        pl.mode = 0xFFFFFFFFu;
    }
    // header
    uint8_t* h = reinterpret_cast<uint8_t*>(ctx->h_io);
    memset(h, 0, HF_CONTAINER_HEADER_BYTES);
    memcpy(h, "HUFHIP1", 8);
    put32(h + 8, pl.version);
    put32(h + 12, HF_CONTAINER_HEADER_BYTES);
    put32(h + 16, pl.layout);
    put32(h + 20, pl.codebook_kind);
    put32(h + 24, pl.chunk_syms);
    put32(h + 28, pl.max_code_len);
    put32(h + 32, pl.mode);
    put32(h + 36, pl.entropy_q16);
    put64(h + 40, pl.n);
    put64(h + 48, pl.nchunks);
    put64(h + 56, pl.payload_words);
    put64(h + 64, pl.off_codebook);
    put64(h + 72, pl.off_chunk_off);
    put64(h + 80, pl.off_side);
    put64(h + 88, pl.off_payload);
    put64(h + 96, pl.total_bytes);
    if (pl.version >= 2) {
        put32(h + 104, pl.placement);
        put64(h + 112, pl.off_chunk_words);
    }
    HF_HIP(hipMemcpyAsync(blob, h, HF_CONTAINER_HEADER_BYTES, hipMemcpyHostToDevice, s));
    HF_HIP(hipStreamSynchronize(s));
    *info = pl;
    return HF_OK;
}

int hf_container_open(hf_ctx* ctx, const void* d_blob, uint64_t blob_bytes,
                      hf_container_info* info) {
    if (!ctx || !d_blob || !info) return HF_ERR_ARG;
    if (!aligned16(d_blob)) return HF_ERR_ALIGN;
    if (blob_bytes < HF_CONTAINER_HEADER_BYTES) return HF_ERR_CORRUPT;
    HF_USE_DEVICE(ctx);
    uint8_t* h = reinterpret_cast<uint8_t*>(ctx->h_io);
    HF_HIP(hipMemcpyAsync(h, d_blob, HF_CONTAINER_HEADER_BYTES, hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(hipStreamSynchronize(ctx->stream));
    if (memcmp(h, "HUFHIP1", 8) != 0) return HF_ERR_CORRUPT;
    const uint32_t version = get32(h + 8);
    if ((version != 1 && version != 2) || get32(h + 12) != HF_CONTAINER_HEADER_BYTES)
        return HF_ERR_CORRUPT;
    if (get32(h + 24) != HF_CHUNK_SYMS || get32(h + 28) != HF_MAX_CODE_LEN) return HF_ERR_CORRUPT;
    // (a version-1 blob is dense and has zeros where version 2 keeps its two fields; a dense
    // stream is never written as version 2)
    const uint32_t placement = version == 1 ? (uint32_t)HF_PLACEMENT_DENSE : get32(h + 104);
    if (version == 2 && placement == HF_PLACEMENT_DENSE) return HF_ERR_CORRUPT;
    hf_container_info pl;
    if (!plan_container(get64(h + 40), (int)get32(h + 16), (int)get32(h + 20), (int)placement,
                        get64(h + 56), &pl))
        return HF_ERR_CORRUPT;
    if (get64(h + 48) != pl.nchunks || get64(h + 64) != pl.off_codebook ||
        get64(h + 72) != pl.off_chunk_off || get64(h + 80) != pl.off_side ||
        get64(h + 88) != pl.off_payload || get64(h + 96) != pl.total_bytes ||
        pl.total_bytes > blob_bytes)
        return HF_ERR_CORRUPT;
    if (version == 2 && (get64(h + 112) != pl.off_chunk_words || get32(h + 108) != 0))
        return HF_ERR_CORRUPT;
    for (int i = version == 2 ? 120 : 104; i < HF_CONTAINER_HEADER_BYTES; ++i)
        if (h[i] != 0) return HF_ERR_CORRUPT;
    pl.mode = get32(h + 32);
    pl.entropy_q16 = get32(h + 36);
    if (pl.codebook_kind == HF_CODEBOOK_STREAM && pl.mode > 1u) return HF_ERR_CORRUPT;
    *info = pl;
    return HF_OK;
}

int hf_container_codebook(hf_ctx* ctx, const void* d_blob, const hf_container_info* info,
                          void* d_codebook_out) {
    if (!ctx || !d_blob || !info) return HF_ERR_ARG;
    if (!aligned16(d_blob) || !aligned16(d_codebook_out)) return HF_ERR_ALIGN;
    hf_container_info pl;
    if (!plan_container(info->n, (int)info->layout, (int)info->codebook_kind, (int)info->placement,
                        info->payload_words, &pl) ||
        pl.off_codebook != info->off_codebook)
        return HF_ERR_ARG;
    if (!pl.off_codebook) return HF_OK;  // paged with records: the codebooks are in the pages
    if (!d_codebook_out) return HF_ERR_ARG;
    HF_USE_DEVICE(ctx);
    const uint8_t* blob = reinterpret_cast<const uint8_t*>(d_blob);
    if (pl.codebook_kind == HF_CODEBOOK_STREAM) {
        if (info->mode > 1u) return HF_ERR_CORRUPT;
        uint8_t* w = reinterpret_cast<uint8_t*>(ctx->h_io);
        HF_HIP(hipMemcpyAsync(w, blob + pl.off_codebook, 768, hipMemcpyDeviceToHost, ctx->stream));
        HF_HIP(hipStreamSynchronize(ctx->stream));
        uint16_t code[256];
        uint8_t len[256];
        memcpy(code, w, 512);
        memcpy(len, w + 512, 256);
        for (int s = 0; s < 256; ++s)
            if (len[s] > HF_MAX_CODE_LEN) return HF_ERR_CORRUPT;
        if (int rc = hf_codebook_upload(ctx, code, len, (int)info->mode, d_codebook_out)) return rc;
        hf_cb_dev* cb = reinterpret_cast<hf_cb_dev*>(d_codebook_out);
        ctx->h_ctl[0] = info->entropy_q16;
        HF_HIP(hipMemcpyAsync(&cb->entropy_q16, ctx->h_ctl, 4, hipMemcpyHostToDevice, ctx->stream));
        HF_HIP(hipStreamSynchronize(ctx->stream));
        return HF_OK;
    }
    hf_launch_wire_to_cb(reinterpret_cast<const hf_wire_chunk_cb*>(blob + pl.off_codebook),
                         pl.nchunks, reinterpret_cast<hf_chunk_cb*>(d_codebook_out),
                         ctx->d_ctl + 1, ctx->stream);
    HF_HIP(hipGetLastError());
    HF_HIP(hipMemcpyAsync(ctx->h_ctl, ctx->d_ctl, 64, hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(hipStreamSynchronize(ctx->stream));
    return report_kernel_errors(ctx);
}

/* ---- TreeQualityAnalyzer --------------------------------------------------------------------------
 * Host arithmetic with the RTL's widths (tree_q_analyser.scala:21-47): total is a W-bit sum,
 * each freq*len product and their sum keep W+6 bits, weighted_length is a 32-bit wire,
 * avg_length_calc keeps (weighted << 8) / total, the three outputs are 16 bits wide. */
int hf_quality(hf_ctx* ctx, const uint64_t* h_freq, const uint8_t* h_len, int N, int wt_width,
               uint16_t* avg_len_q8, uint16_t* efficiency, uint16_t* compression_ratio) {
    (void)ctx;
    if (!h_freq || !h_len || N < 2 || N > 256 || wt_width < 1 || wt_width > 64) return HF_ERR_ARG;
    const uint64_t m = mask_w(wt_width);
    uint64_t total = 0;
    unsigned __int128 acc = 0;
    for (int i = 0; i < N; ++i) {
        const uint64_t f = h_freq[i] & m;
        total = (total + f) & m;
        acc += (unsigned __int128)f * (uint64_t)(h_len[i] & 63u);
    }
    acc &= (((unsigned __int128)1) << (wt_width + 6)) - 1;
    const uint64_t weighted = (uint64_t)acc & 0xFFFFFFFFull;
    if (total == 0) return HF_ERR_RANGE;  // the RTL divides by zero here
    const uint64_t avg = (weighted << 8) / total;
    if (avg == 0) return HF_ERR_RANGE;
    const uint64_t bits = (uint64_t)log2ceil_i(N);
    const uint64_t theo = (bits << 8) & 0xFFFFull;
    if (avg_len_q8) *avg_len_q8 = (uint16_t)avg;
    if (efficiency) *efficiency = (uint16_t)((theo << 8) / avg);
    if (compression_ratio) *compression_ratio = (uint16_t)((avg << 8) / (bits << 8));
    return HF_OK;
}

}  // extern "C"
