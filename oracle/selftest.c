/*
 * selftest.c -- drives every oracle entry point over seeded inputs, for a build with
 * -fsanitize=address,undefined (make -C oracle sanitize): the oracle is what every parity test
 * trusts, so its own memory safety and freedom from undefined behaviour are checked here.
 * TEST INFRASTRUCTURE ONLY, like the oracle itself.  Exit status 0 = all invariants held.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hf_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}

#define CHECK(cond)                                                      \
    do {                                                                 \
        if (!(cond)) {                                                   \
            fprintf(stderr, "selftest: %s failed, line %d\n", #cond, __LINE__); \
            return 1;                                                    \
        }                                                                \
    } while (0)

static int one_stream(size_t n, int skew) {
    uint8_t* sym = malloc(n ? n : 1);
    for (size_t i = 0; i < n; ++i) {
        uint32_t r = rnd();
        sym[i] = skew == 0 ? (uint8_t)r : skew == 1 ? (uint8_t)__builtin_ctz(r | 0x80000000u) : 77;
    }
    uint64_t freq[256];
    orc_aw_symbol_stat(sym, n, 256, 64, freq);
    uint64_t sorted[256];
    uint16_t ids[256];
    orc_aw_symbol_sort(freq, 256, sorted, ids);
    for (int i = 1; i < 256; ++i) CHECK(sorted[i - 1] >= sorted[i]);
    uint16_t code[256];
    uint8_t len[256];
    CHECK(orc_rp_system_codebook(freq, 256, 64, 0, 16, code, len) == 0);
    double kraft = 0;
    for (int i = 0; i < 256; ++i) {
        CHECK(len[i] >= 1 && len[i] <= 16);
        kraft += 1.0 / (double)(1u << len[i]);
    }
    CHECK(kraft <= 1.0 + 1e-12);
    (void)orc_rp_entropy_q16_u64(freq, 256);
    (void)orc_rp_best_mode(freq, len, 256);
    uint16_t q[3];
    (void)orc_aw_quality(freq, len, 256, 64, &q[0], &q[1], &q[2]);
    const size_t nch = orc_num_chunks(n);
    const uint64_t bound = orc_payload_bound_words(n);
    uint32_t* off = malloc((nch + 1) * sizeof(uint32_t));
    const uint64_t words = orc_rp_encode(sym, n, code, len, NULL, off);
    CHECK(words <= bound);
    uint32_t* payload = malloc((words ? words : 1) * sizeof(uint32_t));
    CHECK(orc_rp_encode(sym, n, code, len, payload, off) == words);
    CHECK(off[nch] == words);
    uint8_t* back = malloc(n ? n : 1);
    CHECK(orc_rp_decode(payload, off, n, code, len, back) == 0);
    CHECK(memcmp(back, sym, n) == 0);
    uint8_t* beats = malloc(4 * (n ? n : 1));
    orc_aw_encode_pairs(sym, n, code, len, 1, 8, beats);
    /* container of the stream (one codebook, byte layout) */
    const uint64_t cbytes = orc_container_bytes(n, ORC_LAYOUT_U8, 0, words);
    uint8_t* blob = malloc(cbytes);
    CHECK(orc_container_pack(blob, n, ORC_LAYOUT_U8, 0, words, 1, 0, code, len, NULL, NULL, NULL,
                             NULL, off, NULL, payload) == cbytes);
    free(blob);
    free(beats);
    free(back);
    free(payload);
    free(off);
    free(sym);
    return 0;
}

static int tables(void) {
    for (int it = 0; it < 200; ++it) {
        const int N = 1 << (1 + rnd() % 8); /* 2 .. 256 */
        uint64_t f[256];
        const int wide = rnd() % 3;
        for (int i = 0; i < N; ++i)
            f[i] = wide == 0 ? rnd() % 4 : wide == 1 ? rnd() % 100000 : (uint64_t)rnd() << (rnd() % 12);
        uint64_t w[511];
        uint16_t l[511], r[511], root, h[511], depth[256], adj[256], code[256];
        orc_aw_tree_build(f, N, 64, w, l, r, &root, h);
        orc_rp_leaf_depths(l, r, root, N, depth);
        for (int lmax = 16; lmax >= 8; lmax -= 4) {
            if ((1 << lmax) < N) continue;
            CHECK(orc_rp_tree_adjust(depth, N, lmax, adj) >= 0);
            double kraft = 0;
            for (int i = 0; i < N; ++i) {
                CHECK(adj[i] <= lmax);
                kraft += adj[i] ? 1.0 / (double)(1u << adj[i]) : 1.0;
            }
            CHECK(N == 1 || kraft <= 1.0 + 1e-12);
            orc_rp_canonical_codes(adj, N, code);
            uint16_t cnt[17], cs[17];
            orc_aw_code_start(adj, N, cnt, cs);
            orc_aw_codes_as_written(adj, N, code);
        }
        uint16_t cb[256];
        uint8_t ln[256];
        CHECK(orc_rp_codebook(f, N, 64, rnd() % 3, 16, cb, ln) == 0);
        uint64_t f32[256], flt[256];
        for (int i = 0; i < N; ++i) f32[i] = f[i] & 0xFFFFFu;
        (void)orc_aw_entropy_q16(f32, N, 32);
        orc_aw_filter(f, N, 5, flt);
        for (uint32_t e = 0; e < 0x80000u; e += 0x7FFFu) {
            uint32_t m, t, d;
            uint64_t mf;
            (void)orc_aw_adaptive_mode(e);
            (void)orc_aw_adaptive_config(e, N, 2 + (int)(rnd() % 63), &m, &t, &d, &mf);
        }
    }
    return 0;
}

int main(void) {
    static const size_t sizes[] = {0, 1, 7, 8, 9, 511, 16383, 16384, 16385, 3 * 16384 + 77, 200000};
    for (size_t k = 0; k < sizeof(sizes) / sizeof(sizes[0]); ++k)
        for (int skew = 0; skew < 3; ++skew)
            if (one_stream(sizes[k], skew)) return 1;
    /* the 16-bit front end */
    uint16_t w16[1000];
    uint8_t s8[1000], side[1000];
    uint16_t back16[1000];
    for (int i = 0; i < 1000; ++i) w16[i] = (uint16_t)rnd();
    orc_symbolize(w16, 1000, ORC_LAYOUT_BF16_EXP, s8, side);
    orc_desymbolize(s8, side, 1000, ORC_LAYOUT_BF16_EXP, back16);
    if (memcmp(w16, back16, sizeof(w16)) != 0) return 1;
    if (tables()) return 1;
    puts("oracle selftest ok");
    return 0;
}
