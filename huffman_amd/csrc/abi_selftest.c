/*
 * abi_selftest.c -- the host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer
 * (make -C huffman_amd/csrc sanitize-host).  It needs no GPU: it drives the entry points that are
 * host arithmetic and the argument checks every other entry point runs before its first HIP call.
 * With a GPU present hf_ctx_create succeeds and the context is destroyed again.
 */
#include <stdio.h>
#include <string.h>

#include "../../include/huffhip.h"

#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            fprintf(stderr, "abi selftest: %s failed, line %d\n", #cond, __LINE__); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

int main(void) {
    CHECK(hf_abi_version() == HF_ABI_VERSION);
    for (int code = -12; code <= 1; ++code) CHECK(hf_error_string(code) != NULL);
    CHECK(hf_num_chunks(0) == 0 && hf_num_chunks(1) == 1 && hf_num_chunks(16385) == 2);
    CHECK(hf_payload_bound_words(16384) == 64ull * 129ull);

    /* TreeQualityAnalyzer and AdaptiveThreshold: host arithmetic over every width */
    uint64_t freq[256];
    uint8_t len[256];
    for (int i = 0; i < 256; ++i) {
        freq[i] = (uint64_t)(i * 2654435761u) >> 7;
        len[i] = (uint8_t)(1 + i % 16);
    }
    for (int W = 1; W <= 64; ++W) {
        uint16_t a, e, r;
        const int rc = hf_quality(NULL, freq, len, 256, W, &a, &e, &r);
        CHECK(rc == HF_OK || rc == HF_ERR_RANGE);
        hf_adaptive_config cfg;
        for (uint32_t ent = 0; ent < 0x80000u; ent += 0x3FFFu) {
            const int rc2 = hf_adaptive_threshold(ent, 256, W, &cfg);
            CHECK(rc2 == HF_OK || rc2 == HF_ERR_RANGE || (W < 2 && rc2 == HF_ERR_ARG));
        }
    }
    CHECK(hf_quality(NULL, NULL, len, 256, 32, NULL, NULL, NULL) == HF_ERR_ARG);
    CHECK(hf_adaptive_threshold(0, 256, 32, NULL) == HF_ERR_ARG);

    /* container planning: host arithmetic */
    hf_container_info info;
    for (uint64_t n = 0; n < 100000; n += 7777)
        for (int kind = 0; kind < 2; ++kind)
            for (int layout = 0; layout < 3; ++layout) {
                CHECK(hf_container_plan(n, layout, kind, n / 3, &info) == HF_OK);
                CHECK(info.total_bytes % 16 == 0 && info.off_payload % 16 == 0);
                CHECK(info.total_bytes >= info.off_payload + 16);
            }
    /* ... with a placement (wire format version 2): slotted streams carry a words array, paged
       streams with records carry no codebook section, paged streams without carry the one */
    for (uint64_t n = 0; n < 100000; n += 7777)
        for (int kind = 0; kind < 2; ++kind)
            for (int placement = 0; placement < 3; ++placement) {
                CHECK(hf_container_plan_placed(n, HF_LAYOUT_U8, kind, placement, n / 3, &info) == HF_OK);
                CHECK(info.placement == (uint32_t)placement && info.total_bytes % 16 == 0);
                CHECK((info.off_chunk_words != 0) == (placement == HF_PLACEMENT_SLOTTED));
                CHECK((info.off_codebook == 0) ==
                      (placement == HF_PLACEMENT_PAGED && kind == HF_CODEBOOK_PER_CHUNK));
            }
    CHECK(hf_container_plan_placed(1, 0, 0, 3, 1, &info) == HF_ERR_ARG);
    CHECK(hf_pages_bound_words(0) == 0 && hf_pages_bound_words(16385) == 2ull * hf_pages_bound_words(1));
    CHECK(hf_container_plan(1, 9, 0, 1, &info) == HF_ERR_ARG);
    CHECK(hf_container_plan(1, 0, 0, 1, NULL) == HF_ERR_ARG);

    /* every context-taking entry point rejects a null context before it touches HIP */
    CHECK(hf_flush(NULL) == HF_ERR_ARG && hf_sync(NULL) == HF_ERR_ARG);
    CHECK(hf_symbol_sort(NULL, freq, 256, freq, NULL) == HF_ERR_ARG);
    CHECK(hf_entropy(NULL, freq, 256, NULL, NULL) == HF_ERR_ARG);
    CHECK(hf_encode(NULL, NULL, 0, 0, NULL, NULL, NULL, 0, NULL, NULL) == HF_ERR_ARG);
    CHECK(hf_decode(NULL, NULL, 0, NULL, NULL, 0, 0, NULL, NULL, 0) == HF_ERR_ARG);
    CHECK(hf_decode_chunks_to(NULL, NULL, 0, NULL, NULL, 0, 0, NULL, NULL, NULL, 0, NULL, 0, 0) ==
          HF_ERR_ARG);
    CHECK(hf_last_hip_error(NULL) == 0);
    CHECK(hf_ctx_reserve(NULL, 0, HF_RESERVE_ALL) == HF_ERR_ARG);
    CHECK(hf_zero_dev(NULL, NULL, 0) == HF_ERR_ARG);
    CHECK(hf_symbol_stat_pages_dev(NULL, NULL, 0, 0, NULL, NULL) == HF_ERR_ARG);
    CHECK(hf_symbol_stat_allreduce(NULL, NULL, NULL) == HF_ERR_ARG);
    CHECK(hf_encode_pages(NULL, NULL, 0, 0, NULL, NULL, NULL, NULL, 0, NULL, NULL) == HF_ERR_ARG);
    CHECK(hf_encode_pages_shared(NULL, NULL, 0, 0, NULL, NULL, NULL, NULL, 0, NULL, NULL) == HF_ERR_ARG);
    CHECK(hf_decode_pages(NULL, NULL, 0, NULL, NULL, 0, 0, NULL, NULL, 0, NULL, 0, 0) == HF_ERR_ARG);
    CHECK(hf_decode_pages_shared(NULL, NULL, 0, NULL, NULL, 0, 0, NULL, NULL, NULL, 0, NULL, 0, 0) ==
          HF_ERR_ARG);
    CHECK(hf_page_codebook_download(NULL, NULL, 0, NULL, 0, NULL, NULL, NULL, NULL) == HF_ERR_ARG);
    CHECK(hf_system_codebook_chunks_static(NULL, NULL, 0, HF_MODE_HUFFMAN, 0, NULL) == HF_ERR_ARG);

    hf_ctx* ctx = NULL;
    const int rc = hf_ctx_create(&ctx, 0, NULL);
    if (rc == HF_OK) {
        CHECK(ctx != NULL);
        CHECK(hf_flush(ctx) == HF_OK);
        CHECK(hf_ctx_reserve(ctx, 1u << 20, HF_RESERVE_ALL) == HF_OK);
        CHECK(hf_ctx_reserve(ctx, 1, 0x80u) == HF_ERR_ARG);
        CHECK(hf_zero_dev(ctx, NULL, 0) == HF_OK);
        CHECK(hf_symbol_stat_allreduce(ctx, NULL, NULL) == HF_ERR_ARG);
        CHECK(hf_ctx_destroy(ctx) == HF_OK);
    } else {
        CHECK(rc == HF_ERR_HIP && ctx == NULL);  /* no GPU here: a clean error, no fallback */
    }
    puts("abi selftest ok");
    return 0;
}
