/*
 * hf_oracle.c -- CPU restatement of the liuyuxivapor/huffman encoder pipeline (plain C11).
 *
 * TEST INFRASTRUCTURE ONLY: see hf_oracle.h.  Every function cites the reference lines it
 * follows.  The reference cannot be built here (Scala 2.13 + Chisel 6 need a JVM, sbt and
 * firtool; none is installed), so this restatement is pinned by the known-answer vectors in
 * the reference's own chiseltest specs (SymbolStat, SymbolSort) and, for everything the
 * reference leaves broken or absent, is "parity unpinned".
 */
#include "hf_oracle.h"

#include <stdlib.h>
#include <string.h>

static inline uint64_t mask_w(int W) { return W >= 64 ? ~0ull : ((1ull << W) - 1ull); }

static inline int log2ceil_i(int x) {
    int b = 0;
    while ((1 << b) < x) ++b;
    return b;
}

/* ======================================================================================= */
/* front end                                                                                */
/* ======================================================================================= */

/* Sub-field layouts (reports/spec.md:22 names FP16 / FP8 / FP4 operands; the reference itself
 * never gets past 8-bit symbols): the symbol is the exponent field, the remaining bits of the
 * element form a side field of `bits` bits, and the side stream is the dense little-endian
 * bit-packing of those fields in element order (field i at bits [i * bits, (i + 1) * bits)),
 * so that every 8 elements fill exactly `bits` bytes. */
static unsigned side_field_bits(int layout) {
    return layout == ORC_LAYOUT_FP16_EXP       ? 11u
           : layout == ORC_LAYOUT_FP8_E4M3_EXP ? 3u
           : layout == ORC_LAYOUT_FP4_E2M1_EXP ? 2u
                                               : 0u;
}

uint64_t orc_side_bytes(uint64_t n, int layout) {
    if (layout == ORC_LAYOUT_BF16_EXP) return n;
    return ((n + 7u) / 8u) * side_field_bits(layout); /* whole groups of 8 elements */
}

static void put_field(uint8_t* side, size_t i, unsigned bits, unsigned v) {
    size_t at = i * bits;
    for (unsigned b = 0; b < bits; ++b, ++at)
        if ((v >> b) & 1u) side[at >> 3] |= (uint8_t)(1u << (at & 7u));
}

static unsigned get_field(const uint8_t* side, size_t i, unsigned bits) {
    size_t at = i * bits;
    unsigned v = 0;
    for (unsigned b = 0; b < bits; ++b, ++at) v |= (unsigned)((side[at >> 3] >> (at & 7u)) & 1u) << b;
    return v;
}

void orc_symbolize(const void* data, size_t n, int layout, uint8_t* sym, uint8_t* side) {
    if (layout == ORC_LAYOUT_BF16_EXP) {
        const uint16_t* w = (const uint16_t*)data;
        for (size_t i = 0; i < n; ++i) {
            sym[i] = (uint8_t)((w[i] >> 7) & 0xFF);
            if (side) side[i] = (uint8_t)(((w[i] >> 8) & 0x80) | (w[i] & 0x7F));
        }
    } else if (layout == ORC_LAYOUT_FP16_EXP) {
        /* s eeeee mmmmmmmmmm: symbol = the 5 exponent bits, side = sign (bit 10) | mantissa */
        const uint16_t* w = (const uint16_t*)data;
        if (side) memset(side, 0, (size_t)orc_side_bytes(n, layout));
        for (size_t i = 0; i < n; ++i) {
            sym[i] = (uint8_t)((w[i] >> 10) & 0x1F);
            if (side) put_field(side, i, 11, ((unsigned)(w[i] >> 15) << 10) | (w[i] & 0x3FFu));
        }
    } else if (layout == ORC_LAYOUT_FP8_E4M3_EXP) {
        /* s eeee mmm: symbol = sign and exponent (5 bits), side = the 3 mantissa bits */
        const uint8_t* w = (const uint8_t*)data;
        if (side) memset(side, 0, (size_t)orc_side_bytes(n, layout));
        for (size_t i = 0; i < n; ++i) {
            sym[i] = (uint8_t)(w[i] >> 3);
            if (side) put_field(side, i, 3, w[i] & 7u);
        }
    } else if (layout == ORC_LAYOUT_FP4_E2M1_EXP) {
        /* two fp4 per byte, s ee m each: symbol = the two sign|exponent fields, side = the two
         * mantissa bits */
        const uint8_t* w = (const uint8_t*)data;
        if (side) memset(side, 0, (size_t)orc_side_bytes(n, layout));
        for (size_t i = 0; i < n; ++i) {
            sym[i] = (uint8_t)(((w[i] >> 1) & 7u) | (((w[i] >> 5) & 7u) << 3));
            if (side) put_field(side, i, 2, (w[i] & 1u) | (((w[i] >> 4) & 1u) << 1));
        }
    } else if (layout == ORC_LAYOUT_BITPLANE8) {
        const uint8_t* w = (const uint8_t*)data;
        for (size_t b0 = 0; b0 < n; b0 += ORC_BITPLANE_BLOCK) {
            const size_t m = n - b0 < ORC_BITPLANE_BLOCK ? n - b0 : ORC_BITPLANE_BLOCK;
            const size_t G = m / 8;
            for (size_t g = 0; g < G; ++g)
                for (unsigned p = 0; p < 8; ++p) {
                    unsigned v = 0;
                    for (unsigned i = 0; i < 8; ++i) v |= ((w[b0 + 8 * g + i] >> p) & 1u) << i;
                    sym[b0 + p * G + g] = (uint8_t)v;
                }
        }
    } else {
        memcpy(sym, data, n);
    }
}

void orc_desymbolize(const uint8_t* sym, const uint8_t* side, size_t n, int layout, void* out) {
    if (layout == ORC_LAYOUT_BF16_EXP) {
        uint16_t* w = (uint16_t*)out;
        for (size_t i = 0; i < n; ++i)
            w[i] = (uint16_t)(((uint16_t)(side[i] & 0x80) << 8) | ((uint16_t)sym[i] << 7) |
                              (side[i] & 0x7F));
    } else if (layout == ORC_LAYOUT_FP16_EXP) {
        uint16_t* w = (uint16_t*)out;
        for (size_t i = 0; i < n; ++i) {
            const unsigned f = get_field(side, i, 11);
            w[i] = (uint16_t)(((f >> 10) << 15) | ((unsigned)(sym[i] & 0x1F) << 10) | (f & 0x3FFu));
        }
    } else if (layout == ORC_LAYOUT_FP8_E4M3_EXP) {
        uint8_t* w = (uint8_t*)out;
        for (size_t i = 0; i < n; ++i)
            w[i] = (uint8_t)(((unsigned)(sym[i] & 0x1F) << 3) | get_field(side, i, 3));
    } else if (layout == ORC_LAYOUT_FP4_E2M1_EXP) {
        uint8_t* w = (uint8_t*)out;
        for (size_t i = 0; i < n; ++i) {
            const unsigned f = get_field(side, i, 2);
            w[i] = (uint8_t)((f & 1u) | ((unsigned)(sym[i] & 7u) << 1) | (((f >> 1) & 1u) << 4) |
                             ((unsigned)((sym[i] >> 3) & 7u) << 5));
        }
    } else if (layout == ORC_LAYOUT_BITPLANE8) {
        uint8_t* w = (uint8_t*)out;
        for (size_t b0 = 0; b0 < n; b0 += ORC_BITPLANE_BLOCK) {
            const size_t m = n - b0 < ORC_BITPLANE_BLOCK ? n - b0 : ORC_BITPLANE_BLOCK;
            const size_t G = m / 8;
            for (size_t g = 0; g < G; ++g)
                for (unsigned i = 0; i < 8; ++i) {
                    unsigned v = 0;
                    for (unsigned p = 0; p < 8; ++p) v |= ((sym[b0 + p * G + g] >> i) & 1u) << p;
                    w[b0 + 8 * g + i] = (uint8_t)v;
                }
        }
    } else {
        memcpy(out, sym, n);
    }
}

/* ======================================================================================= */
/* as written                                                                               */
/* ======================================================================================= */

/* symbol_stat.scala:34-38 -- `freq(sym) := freq(sym) + 1.U`; the Vec index keeps the low
 * log2(N) bits of the symbol, the W-bit add wraps. */
void orc_aw_symbol_stat(const uint8_t* sym, size_t n, int N, int W, uint64_t* freq) {
    const uint64_t m = mask_w(W);
    for (int i = 0; i < N; ++i) freq[i] = 0; /* symbol_stat.scala:27-31 clear on start */
    for (size_t i = 0; i < n; ++i) {
        int s = sym[i] & (N - 1);
        freq[s] = (freq[s] + 1) & m;
    }
}

/* symbol_sort.scala:125-142.  iterCnt runs 0 .. 2N-1; phase = ~iterCnt(0), i.e. 1 on even
 * iterations: pairs (i, i+1) with i odd go first.  Swap only when a < b. */
void orc_aw_symbol_sort(const uint64_t* in, int N, uint64_t* sorted, uint16_t* ids) {
    uint64_t* v = sorted;
    uint16_t* id = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)N);
    uint64_t* nv = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)N);
    uint16_t* nid = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)N);
    for (int i = 0; i < N; ++i) {
        v[i] = in[i];
        id[i] = (uint16_t)i;
    }
    for (int iter = 0; iter < 2 * N; ++iter) {
        int phase = (~iter) & 1;
        /* all comparators of one pass read the old register values (registered update) */
        memcpy(nv, v, sizeof(uint64_t) * (size_t)N);
        memcpy(nid, id, sizeof(uint16_t) * (size_t)N);
        for (int i = 0; i < N - 1; ++i) {
            if ((i % 2) == phase && v[i] < v[i + 1]) {
                nv[i] = v[i + 1];
                nv[i + 1] = v[i];
                nid[i] = id[i + 1];
                nid[i + 1] = id[i];
            }
        }
        memcpy(v, nv, sizeof(uint64_t) * (size_t)N);
        memcpy(id, nid, sizeof(uint16_t) * (size_t)N);
    }
    if (ids) memcpy(ids, id, sizeof(uint16_t) * (size_t)N);
    free(id);
    free(nv);
    free(nid);
}

/* tree_build.scala:47-72 (findTwoMin) and :90-131 (sInit, sMerge).
 * Invalid nodes are masked to maxVal = 2^W - 1; the minimum is taken over all 2N-1 masked
 * weights; the index is the PriorityEncoder (lowest index) over `masked == min && valid`,
 * which yields the last index 2N-2 when nothing matches.  The second search masks the first
 * index as well; the first node is still `valid` there, so it can match again when the second
 * minimum equals maxVal. */
void orc_aw_tree_build(const uint64_t* freqs, int N, int W, uint64_t* weights, uint16_t* lefts,
                       uint16_t* rights, uint16_t* root, uint16_t* heights) {
    const int M = 2 * N - 1;
    const uint64_t m = mask_w(W);
    const uint64_t maxVal = m; /* (1 << W) - 1 */
    const unsigned depth_mask = (1u << log2ceil_i(N + 1)) - 1u; /* depthWidth, :20 */
    uint8_t* valid = (uint8_t*)calloc((size_t)M, 1);
    uint16_t* h = (uint16_t*)calloc((size_t)M, sizeof(uint16_t));
    for (int i = 0; i < M; ++i) { /* sInit, :90-109 */
        weights[i] = i < N ? (freqs[i] & m) : 0;
        lefts[i] = 0;
        rights[i] = 0;
        valid[i] = i < N;
    }
    for (int k = N; k < M; ++k) { /* sMerge, :111-125 */
        uint64_t min1 = maxVal;
        for (int i = 0; i < M; ++i) {
            uint64_t w = valid[i] ? weights[i] : maxVal;
            if (w < min1) min1 = w;
        }
        int i1 = M - 1;
        for (int i = 0; i < M; ++i)
            if (valid[i] && weights[i] == min1) {
                i1 = i;
                break;
            }
        uint64_t min2 = maxVal;
        for (int i = 0; i < M; ++i) {
            uint64_t w = (i == i1 || !valid[i]) ? maxVal : weights[i];
            if (w < min2) min2 = w;
        }
        int i2 = M - 1;
        for (int i = 0; i < M; ++i) {
            uint64_t w = (i == i1 || !valid[i]) ? maxVal : weights[i];
            if (w == min2 && valid[i]) {
                i2 = i;
                break;
            }
        }
        uint64_t wsum = (weights[i1] + weights[i2]) & m; /* :115, W-bit add */
        uint16_t hh = (uint16_t)(((h[i1] >= h[i2] ? h[i1] : h[i2]) + 1u) & depth_mask); /* :120 */
        weights[k] = wsum;
        lefts[k] = (uint16_t)i1;
        rights[k] = (uint16_t)i2;
        h[k] = hh;
        valid[i1] = 0;
        valid[i2] = 0;
        valid[k] = 1; /* last connect wins, :122-124 */
    }
    *root = (uint16_t)(M - 1); /* nodeCnt - 1, :128 */
    if (heights) memcpy(heights, h, sizeof(uint16_t) * (size_t)M);
    free(valid);
    free(h);
}

/* entropy_aware_tree_builder.scala:118-127 */
void orc_aw_code_start(const uint16_t* len, int N, uint16_t count[17], uint16_t code_start[17]) {
    for (int d = 0; d < 17; ++d) {
        unsigned c = 0;
        for (int i = 0; i < N; ++i) c += (len[i] == d);
        count[d] = (uint16_t)c;
    }
    code_start[0] = 0;
    for (int d = 1; d < 17; ++d)
        code_start[d] = (uint16_t)(((unsigned)code_start[d - 1] + count[d - 1]) << 1);
}

/* entropy_aware_tree_builder.scala:133-155: prefixSum(d+1) = prefixSum(d) + count(d),
 * codes(i) = codeStart(len_i) + prefixSum(len_i), indices above 17 read as zero. */
void orc_aw_codes_as_written(const uint16_t* len, int N, uint16_t* code) {
    uint16_t count[17], cs[17], prefix[18];
    orc_aw_code_start(len, N, count, cs);
    prefix[0] = 0;
    for (int d = 0; d <= 16; ++d) prefix[d + 1] = (uint16_t)(prefix[d] + count[d]);
    for (int i = 0; i < N; ++i) {
        unsigned l = len[i] & 31u; /* lvl.pad(5) */
        uint16_t c = l < 17 ? cs[l] : 0;
        uint16_t p = l < 18 ? prefix[l] : 0;
        code[i] = (uint16_t)(c + p);
    }
}

/* shannon_entropy.scala:58-109 with the loop body executed for every index 0..N-1.
 * Chisel widths: prob is W+16 bits; k has log2Ceil(W) bits; the shift amount
 * k + 1 + 16 - 8 is evaluated in max(log2Ceil(W), 5) bits; `1.U << amount` is 32 bits wide
 * for W <= 32; x*k >> 16 and 2x are added in W+17 bits, the subtraction keeps W+17 bits;
 * entropy_acc is a 32-bit register, so only bits 16..47 of prob * approx reach it. */
uint32_t orc_aw_entropy_q16(const uint64_t* freq, int N, int W) {
    const uint64_t m = mask_w(W);
    const int kw = log2ceil_i(W);
    const int sw = kw > 5 ? kw : 5;
    uint64_t total = 0;
    for (int i = 0; i < N; ++i) total = (total + (freq[i] & m)) & m; /* :62 */
    uint32_t acc = 0;
    for (int i = 0; i < N; ++i) {
        uint64_t f = freq[i] & m;
        if (f == 0 || total == 0) continue; /* :70 */
        uint64_t prob = (f << 16) / total;  /* :72 */
        unsigned k = (unsigned)(W - 1);     /* PriorityEncoder default = last index */
        for (int b = W - 1; b >= 0; --b)
            if ((prob >> b) & 1ull) {
                k = (unsigned)(W - 1 - b);
                break;
            }
        unsigned amt = (k + 1u) & ((1u << kw) - 1u); /* :78 */
        amt = (amt + 16u) & ((1u << sw) - 1u);
        amt = (amt - 8u) & ((1u << sw) - 1u);
        uint64_t pow2 = 1ull << amt;
        uint64_t xk = (prob * (uint64_t)k) >> 16; /* :81 */
        uint64_t two_x = prob << 1;               /* :84 */
        uint64_t approx = (xk + two_x - pow2) & ((1ull << (W + 17)) - 1ull); /* :87 */
        uint64_t prod = prob * approx;            /* low 64 bits are exact */
        acc += (uint32_t)(prod >> 16);            /* :90-91 */
    }
    return (uint32_t)(0x10000u - acc); /* :101 */
}

/* adaptive_threshold.scala:26-27 thresholds 0.5 and 6.0 in Q16; :65-79 decision. */
int orc_aw_adaptive_mode(uint32_t e) {
    if (e < 0x8000u) return 1;
    if (e > 0x60000u) return 0;
    return 1;
}

/* adaptive_threshold.scala:41-62: the three register sets, chosen as in :65-79.  Literal
 * values as Scala computes them: (x * 65536).toInt; depth / k in Int arithmetic, kept in a
 * log2Ceil(depth)-bit register; 1 << (wtWidth - 2) as an Int shift (count mod 32), kept in a
 * wtWidth-bit register.  Returns -1 where the Int literal is negative (no such circuit). */
int orc_aw_adaptive_config(uint32_t e, int depth, int W, uint32_t* mode, uint32_t* thr_q16,
                           uint32_t* depth_limit, uint64_t* min_freq) {
    static const double frac[3] = {0.3, 0.6, 0.9};
    static const int div[3] = {4, 2, 8};
    const int sel = e < 0x8000u ? 0 : (e > 0x60000u ? 2 : 1); /* low / medium / high */
    int dbits = 0;
    while ((1 << dbits) < depth) ++dbits;
    const int32_t lit = (int32_t)(1u << ((unsigned)(W - 2) % 32u));
    if (lit < 0) return -1;
    const uint64_t wmask = W >= 64 ? ~0ull : (1ull << W) - 1ull;
    const uint64_t mf[3] = {1ull, 2ull, (uint64_t)lit};
    *mode = sel == 2 ? 0u : 1u;
    *thr_q16 = (uint32_t)(int)(frac[sel] * 65536.0);
    *depth_limit = (uint32_t)(depth / div[sel]) & ((1u << dbits) - 1u);
    *min_freq = mf[sel] & wmask;
    return 0;
}

/* entropy_aware_tree_builder.scala:61-66 */
void orc_aw_filter(const uint64_t* freq, int N, uint64_t thr, uint64_t* out) {
    for (int i = 0; i < N; ++i) out[i] = freq[i] >= thr ? freq[i] : 0;
}

/* top_encoder.scala:131-137 (bypass table) and :148-162 (lookup mux) */
void orc_aw_encode_pairs(const uint8_t* sym, size_t n, const uint16_t* code, const uint8_t* len,
                         int mode, int sym_width, uint8_t* out4) {
    for (size_t i = 0; i < n; ++i) {
        uint16_t c;
        uint8_t l;
        if (mode == 0) {
            c = sym[i];
            l = (uint8_t)sym_width;
        } else {
            c = code[sym[i]];
            l = len[sym[i]];
        }
        out4[4 * i + 0] = (uint8_t)(c & 0xFF);
        out4[4 * i + 1] = (uint8_t)(c >> 8);
        out4[4 * i + 2] = l;
        out4[4 * i + 3] = (uint8_t)(mode & 1);
    }
}

/* tree_q_analyser.scala:21-47.  total is a W-bit sum; each product is W+6 bits and the sum
 * keeps W+6 bits; weighted_length is a 32-bit wire; avg_length_calc keeps 40 bits, the
 * outputs are 16 bits. */
int orc_aw_quality(const uint64_t* freq, const uint8_t* len, int N, int W, uint16_t* avg_len_q8,
                   uint16_t* efficiency, uint16_t* compression_ratio) {
    const uint64_t m = mask_w(W);
    uint64_t total = 0;
    unsigned __int128 acc = 0;
    for (int i = 0; i < N; ++i) {
        total = (total + (freq[i] & m)) & m;
        acc += (unsigned __int128)(freq[i] & m) * (uint64_t)(len[i] & 63);
    }
    if (W + 6 < 128) acc &= (((unsigned __int128)1) << (W + 6)) - 1;
    uint64_t weighted = (uint64_t)acc & 0xFFFFFFFFull; /* :22, :30 */
    if (total == 0) return -1;
    uint64_t avg = (weighted << 8) / total; /* :33 */
    uint64_t theo = ((uint64_t)log2ceil_i(N) << 8) & 0xFFFF; /* :38 */
    if (avg == 0) return -1;
    uint64_t eff = (theo << 8) / avg; /* :41 */
    uint64_t ub = (uint64_t)log2ceil_i(N);
    if (ub == 0) return -1;
    uint64_t ratio = (avg << 8) / (ub << 8); /* :46 */
    *avg_len_q8 = (uint16_t)avg;
    *efficiency = (uint16_t)eff;
    *compression_ratio = (uint16_t)ratio;
    return 0;
}

/* ======================================================================================= */
/* repaired                                                                                 */
/* ======================================================================================= */

/* tree_build.scala:81 returns depthArr.take(numLeaves), which sInit zeroes (:97) and sMerge
 * never writes (:120 writes index nodeCnt >= N only).  The intended output is the depth of
 * each leaf below the root: walk lefts/rights from the root.  Children always have a lower
 * index than their parent, so one descending sweep suffices. */
void orc_rp_leaf_depths(const uint16_t* lefts, const uint16_t* rights, int root, int N,
                        uint16_t* depths) {
    const int M = 2 * N - 1;
    uint16_t* d = (uint16_t*)calloc((size_t)M, sizeof(uint16_t));
    for (int k = root; k >= N; --k) {
        d[lefts[k]] = (uint16_t)(d[k] + 1);
        d[rights[k]] = (uint16_t)(d[k] + 1);
    }
    memcpy(depths, d, sizeof(uint16_t) * (size_t)N);
    free(d);
}

/* tree_adjust.scala keeps a depth histogram (sBuildHist :44-57), moves leaves between levels
 * (sMigrate :59-82, sFillHoles :85-106) and deals the resulting depth multiset to indices in
 * ascending order (sAssign :108-119).  As written the histogram is a last-connect, not a sum
 * (:45-49), K comes from the previous run (:50-54), holesLeft is truncated to 6 bits
 * (:34, :79), the level moves do not preserve the Kraft sum (:71-73, :96-98) and sAssign never
 * terminates for power-of-two N (:24-25, :109).  This keeps the four phases and the
 * assign-by-index rule with arithmetic that provably yields a prefix-code length set:
 *   BuildHist  true histogram, depths above lmax clamped to lmax, kraft = sum hist[d] 2^(lmax-d)
 *   Migrate    while kraft > 2^lmax: deepest level d < lmax with a leaf: move one leaf d -> d+1
 *   FillHoles  holes = 2^lmax - kraft (cf. emptyL, :34): for d = lmax .. 2: while a leaf sits
 *              at d and holes >= 2^(lmax-d): move it d -> d-1
 *   Assign     index p = 0 .. N-1 takes the smallest remaining depth (:108-116)            */
int orc_rp_tree_adjust(const uint16_t* depths_in, int N, int lmax, uint16_t* depths_out) {
    if (lmax < 1 || lmax > 16 || (1l << lmax) < N) return -1;
    uint32_t hist[17];
    memset(hist, 0, sizeof hist);
    for (int i = 0; i < N; ++i) {
        int d = depths_in[i] > lmax ? lmax : depths_in[i];
        hist[d]++;
    }
    uint64_t kraft = 0;
    for (int d = 0; d <= lmax; ++d) kraft += (uint64_t)hist[d] << (lmax - d);
    const uint64_t full = 1ull << lmax;
    int steps = 0;
    while (kraft > full) {
        int d = lmax - 1;
        while (d >= 0 && hist[d] == 0) --d;
        if (d < 0) return -1; /* unreachable: all leaves at lmax means kraft = N <= 2^lmax */
        hist[d]--;
        hist[d + 1]++;
        kraft -= 1ull << (lmax - d - 1);
        ++steps;
    }
    uint64_t holes = full - kraft;
    for (int d = lmax; d >= 2; --d) {
        const uint64_t unit = 1ull << (lmax - d);
        while (hist[d] > 0 && holes >= unit) {
            hist[d]--;
            hist[d - 1]++;
            holes -= unit;
        }
    }
    int d = 0;
    for (int p = 0; p < N; ++p) {
        while (d <= lmax && hist[d] == 0) ++d;
        depths_out[p] = (uint16_t)d;
        hist[d]--;
    }
    return steps;
}

/* The per-symbol offset as written (entropy_aware_tree_builder.scala:148-155) gives every
 * symbol of one length the same code.  The commented-out original (:105-111) walks i upward
 * handing out nextCode(len_i)++: the standard canonical assignment in index order.  count[0]
 * is forced to zero so a zero length cannot shift the code space. */
void orc_rp_canonical_codes(const uint16_t* len, int N, uint16_t* code) {
    uint16_t count[17], cs[17];
    for (int d = 0; d < 17; ++d) count[d] = 0;
    for (int i = 0; i < N; ++i)
        if (len[i] >= 1 && len[i] <= 16) count[len[i]]++;
    cs[0] = 0;
    for (int d = 1; d < 17; ++d) cs[d] = (uint16_t)(((unsigned)cs[d - 1] + count[d - 1]) << 1);
    uint16_t next[17];
    memcpy(next, cs, sizeof next);
    for (int i = 0; i < N; ++i) {
        if (len[i] >= 1 && len[i] <= 16)
            code[i] = next[len[i]]++;
        else
            code[i] = 0;
    }
}

/* EntropyAwareTreeBuilder FSM order (entropy_aware_tree_builder.scala:40): sFilter -> sBuild ->
 * sAdjust -> sGenPrep/sGenExec, with the repaired pieces substituted. */
int orc_rp_codebook(const uint64_t* freq, int N, int W, uint64_t thr, int lmax, uint16_t* code,
                    uint8_t* len) {
    const int M = 2 * N - 1;
    uint64_t* filt = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)N);
    uint64_t* wts = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)M);
    uint16_t* l = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)M);
    uint16_t* r = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)M);
    uint16_t* dep = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)N);
    uint16_t* adj = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)N);
    uint16_t root;
    orc_aw_filter(freq, N, thr, filt);
    orc_aw_tree_build(filt, N, W, wts, l, r, &root, NULL);
    orc_rp_leaf_depths(l, r, root, N, dep);
    int rc = orc_rp_tree_adjust(dep, N, lmax, adj);
    if (rc >= 0) {
        orc_rp_canonical_codes(adj, N, code);
        for (int i = 0; i < N; ++i) len[i] = (uint8_t)adj[i];
        rc = 0;
    }
    free(filt);
    free(wts);
    free(l);
    free(r);
    free(dep);
    free(adj);
    return rc;
}

/* top_encoder.scala:106 feeds the *sorted* counts to the tree builder, so its codes come out
 * indexed by rank, while :155-157 index the table by raw symbol; SymbolSort drops the symbol
 * identities that would connect the two.  Here the identities ride along with the sort and the
 * codebook is scattered back: table[rank2sym[p]] = codebook_over_ranks[p]. */
int orc_rp_system_codebook(const uint64_t* freq_by_symbol, int N, int W, uint64_t thr, int lmax,
                           uint16_t* code, uint8_t* len) {
    uint64_t* sorted = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)N);
    uint16_t* ids = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)N);
    uint16_t* rc_code = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)N);
    uint8_t* rc_len = (uint8_t*)malloc((size_t)N);
    orc_aw_symbol_sort(freq_by_symbol, N, sorted, ids);
    int rc = orc_rp_codebook(sorted, N, W, thr, lmax, rc_code, rc_len);
    if (rc == 0)
        for (int p = 0; p < N; ++p) {
            code[ids[p]] = rc_code[p];
            len[ids[p]] = rc_len[p];
        }
    free(sorted);
    free(ids);
    free(rc_code);
    free(rc_len);
    return rc;
}

uint32_t orc_rp_entropy_q16_u64(const uint64_t* freq, int N) {
    uint64_t total = 0;
    for (int i = 0; i < N; ++i) total += freq[i];
    int shift = 0;
    while ((total >> shift) >> 32) ++shift;
    uint64_t* f = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)N);
    for (int i = 0; i < N; ++i) f[i] = freq[i] >> shift;
    uint32_t e = orc_aw_entropy_q16(f, N, 32);
    free(f);
    return e;
}

/* ---- packed stream ---------------------------------------------------------------------- */

size_t orc_num_chunks(size_t n) { return (n + ORC_CHUNK_SYMS - 1) / ORC_CHUNK_SYMS; }

uint64_t orc_payload_bound_words(size_t n) {
    /* a lane emits at most ceil(256*16/32) = 128 words of code bits plus one word the
     * decoder asks for ahead of need */
    return (uint64_t)orc_num_chunks(n) * ORC_LANES * 129u;
}

#define LANE_SYMS (ORC_GROUP * ORC_ROUNDS) /* 256 */
#define PAIR_STEPS (LANE_SYMS / 2)         /* 128 */

uint64_t orc_rp_encode(const uint8_t* sym, size_t n, const uint16_t* code, const uint8_t* len,
                       uint32_t* payload, uint32_t* chunk_off) {
    const size_t nchunks = orc_num_chunks(n);
    uint64_t pos = 0;
    /* per lane: up to 128 words of bits, pair lengths */
    uint32_t(*words)[129] = malloc(sizeof(uint32_t[129]) * ORC_LANES);
    uint8_t(*plen)[PAIR_STEPS] = malloc(sizeof(uint8_t[PAIR_STEPS]) * ORC_LANES);
    for (size_t c = 0; c < nchunks; ++c) {
        const size_t base = c * (size_t)ORC_CHUNK_SYMS;
        const size_t nvalid = n - base < ORC_CHUNK_SYMS ? n - base : ORC_CHUNK_SYMS;
        unsigned nwords[ORC_LANES];
        if (chunk_off) chunk_off[c] = (uint32_t)pos;
        for (unsigned l = 0; l < ORC_LANES; ++l) {
            uint64_t acc = 0;
            unsigned accbits = 0, nw = 0;
            memset(plen[l], 0, PAIR_STEPS);
            for (unsigned r = 0; r < ORC_ROUNDS; ++r)
                for (unsigned j = 0; j < ORC_GROUP; ++j) {
                    size_t i = ((size_t)r * ORC_LANES + l) * ORC_GROUP + j;
                    if (i >= nvalid) continue;
                    uint8_t s = sym[base + i];
                    unsigned L = len[s];
                    acc = (acc << L) | (uint64_t)(code[s] & ((1u << L) - 1u));
                    accbits += L;
                    plen[l][r * 4 + j / 2] = (uint8_t)(plen[l][r * 4 + j / 2] + L);
                    while (accbits >= 32) {
                        words[l][nw++] = (uint32_t)(acc >> (accbits - 32));
                        accbits -= 32;
                    }
                }
            if (accbits) words[l][nw++] = (uint32_t)(acc << (32 - accbits));
            nwords[l] = nw;
        }
        int nb[ORC_LANES];
        unsigned k[ORC_LANES];
        for (unsigned l = 0; l < ORC_LANES; ++l) nb[l] = 0, k[l] = 0;
        for (unsigned t = 0; t < PAIR_STEPS; ++t) {
            const unsigned r = t / 4, jp = t % 4;
            for (unsigned l = 0; l < ORC_LANES; ++l) {
                size_t i0 = ((size_t)r * ORC_LANES + l) * ORC_GROUP + 2 * jp;
                if (i0 >= nvalid) continue;
                if (nb[l] < 32) {
                    uint32_t w = k[l] < nwords[l] ? words[l][k[l]] : 0u;
                    if (payload) payload[pos] = w;
                    ++pos;
                    ++k[l];
                    nb[l] += 32;
                }
                nb[l] -= plen[l][t];
            }
        }
    }
    if (chunk_off) chunk_off[nchunks] = (uint32_t)pos;
    free(words);
    free(plen);
    return pos;
}

int orc_rp_decode(const uint32_t* payload, const uint32_t* chunk_off, size_t n,
                  const uint16_t* code, const uint8_t* len, uint8_t* sym_out) {
    /* 16-bit prefix table: entry = len << 8 | sym, 0 = no code */
    uint16_t* tab = (uint16_t*)calloc(65536, sizeof(uint16_t));
    for (int s = 0; s < 256; ++s) {
        unsigned L = len[s];
        if (L == 0 || L > 16) continue;
        unsigned first = (unsigned)(code[s] & ((1u << L) - 1u)) << (16 - L);
        for (unsigned e = 0; e < (1u << (16 - L)); ++e) tab[first + e] = (uint16_t)((L << 8) | s);
    }
    const size_t nchunks = orc_num_chunks(n);
    int rc = 0;
    for (size_t c = 0; c < nchunks && rc == 0; ++c) {
        const size_t base = c * (size_t)ORC_CHUNK_SYMS;
        const size_t nvalid = n - base < ORC_CHUNK_SYMS ? n - base : ORC_CHUNK_SYMS;
        uint64_t win[ORC_LANES];
        int nb[ORC_LANES];
        for (unsigned l = 0; l < ORC_LANES; ++l) win[l] = 0, nb[l] = 0;
        uint64_t pos = chunk_off[c];
        const uint64_t end = chunk_off[c + 1];
        for (unsigned t = 0; t < PAIR_STEPS && rc == 0; ++t) {
            const unsigned r = t / 4, jp = t % 4;
            for (unsigned l = 0; l < ORC_LANES; ++l) {
                size_t i0 = ((size_t)r * ORC_LANES + l) * ORC_GROUP + 2 * jp;
                if (i0 >= nvalid) continue;
                if (nb[l] < 32) {
                    if (pos >= end) {
                        rc = -1;
                        break;
                    }
                    win[l] |= (uint64_t)payload[pos++] << (32 - nb[l]);
                    nb[l] += 32;
                }
                for (unsigned q = 0; q < 2; ++q) {
                    if (i0 + q >= nvalid) break;
                    uint16_t e = tab[win[l] >> 48];
                    unsigned L = e >> 8;
                    if (L == 0) {
                        rc = -1;
                        break;
                    }
                    sym_out[base + i0 + q] = (uint8_t)(e & 0xFF);
                    win[l] <<= L;
                    nb[l] -= (int)L;
                }
                if (rc) break;
            }
        }
        if (rc == 0 && pos != end) rc = -1;
    }
    free(tab);
    return rc;
}

/* ---- container, wire format version 1 ------------------------------------------------------- */
static uint64_t round16(uint64_t x) { return (x + 15u) / 16u * 16u; }

static void le32(uint8_t* p, uint32_t v) {
    for (int i = 0; i < 4; ++i) p[i] = (uint8_t)(v >> (8 * i));
}
static void le64(uint8_t* p, uint64_t v) {
    for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(v >> (8 * i));
}

/* offsets of the four sections and the total size */
static void container_offsets(uint64_t n, int layout, int kind, uint64_t payload_words,
                              uint64_t off[4], uint64_t* total) {
    const uint64_t nchunks = orc_num_chunks((size_t)n);
    uint64_t at = 128;
    off[0] = at;
    at = round16(at + (kind == 0 ? 768u : nchunks * 296u));
    off[1] = at;
    at = round16(at + 4u * (nchunks + 1));
    off[2] = 0;
    if (orc_side_bytes(n, layout) != 0 || layout == ORC_LAYOUT_BF16_EXP) {
        off[2] = at;
        at = round16(at + orc_side_bytes(n, layout));
    }
    off[3] = at;
    at += payload_words * 4u < 16u ? 16u : round16(payload_words * 4u);
    *total = at;
}

uint64_t orc_container_bytes(uint64_t n, int layout, int kind, uint64_t payload_words) {
    uint64_t off[4], total;
    container_offsets(n, layout, kind, payload_words, off, &total);
    return total;
}

uint64_t orc_container_pack(uint8_t* out, uint64_t n, int layout, int kind, uint64_t payload_words,
                            uint32_t mode, uint32_t entropy_q16, const uint16_t* code,
                            const uint8_t* len, const uint8_t* rank2sym,
                            const uint8_t* len_by_rank, const uint32_t* modes,
                            const uint32_t* entropies, const uint32_t* chunk_off,
                            const uint8_t* side, const uint32_t* payload) {
    uint64_t off[4], total;
    container_offsets(n, layout, kind, payload_words, off, &total);
    const uint64_t nchunks = orc_num_chunks((size_t)n);
    memset(out, 0, (size_t)total);
    memcpy(out, "HUFHIP1", 8);
    le32(out + 8, 1);
    le32(out + 12, 128);
    le32(out + 16, (uint32_t)layout);
    le32(out + 20, (uint32_t)kind);
    le32(out + 24, ORC_CHUNK_SYMS);
    le32(out + 28, 16);
    le32(out + 32, kind == 0 ? mode : 0xFFFFFFFFu);
    le32(out + 36, kind == 0 ? entropy_q16 : 0u);
    le64(out + 40, n);
    le64(out + 48, nchunks);
    le64(out + 56, payload_words);
    for (int i = 0; i < 4; ++i) le64(out + 64 + 8 * i, off[i]);
    le64(out + 96, total);
    uint8_t* p = out + off[0];
    if (kind == 0) {
        for (int s = 0; s < 256; ++s) {
            p[2 * s] = (uint8_t)(code[s] & 0xFF);
            p[2 * s + 1] = (uint8_t)(code[s] >> 8);
            p[512 + s] = len[s];
        }
    } else {
        for (uint64_t c = 0; c < nchunks; ++c, p += 296) {
            uint32_t per_len[17] = {0};
            memcpy(p, rank2sym + c * 256, 256);
            for (int r = 0; r < 256; ++r) {
                const unsigned l = len_by_rank[c * 256 + r];
                if (l >= 1 && l <= 16) per_len[l]++;
            }
            for (int l = 1; l <= 16; ++l) {
                p[256 + 2 * (l - 1)] = (uint8_t)(per_len[l] & 0xFF);
                p[256 + 2 * (l - 1) + 1] = (uint8_t)(per_len[l] >> 8);
            }
            le32(p + 288, modes[c]);
            le32(p + 292, entropies[c]);
        }
    }
    for (uint64_t c = 0; c <= nchunks; ++c) le32(out + off[1] + 4 * c, chunk_off[c]);
    if (off[2] && n) memcpy(out + off[2], side, (size_t)n);
    for (uint64_t w = 0; w < payload_words; ++w) le32(out + off[3] + 4 * w, payload[w]);
    return total;
}

int orc_rp_best_mode(const uint64_t* freq_by_symbol, const uint8_t* len_by_symbol, int N) {
    unsigned __int128 bits = 0, total = 0;
    for (int s = 0; s < N; ++s) {
        bits += (unsigned __int128)freq_by_symbol[s] * len_by_symbol[s];
        total += freq_by_symbol[s];
    }
    return bits < 8 * total ? 1 : 0;
}

/* ---- per-block codebooks ------------------------------------------------------------------
 * Every chunk of ORC_CHUNK_SYMS symbols is a block of its own, as the reference treats every
 * block it is fed (symbol_stat.scala:26-33 clears the counters on each `start`): SymbolStat,
 * the repaired system codebook (Huffman forced) and the packed stream of that chunk alone.
 * code / len receive nchunks x 256 entries by raw symbol.  chunk_off as in orc_rp_encode. */
uint64_t orc_rp_encode_blocks(const uint8_t* sym, size_t n, uint64_t min_freq_threshold,
                              uint16_t* code, uint8_t* len, uint32_t* payload,
                              uint32_t* chunk_off) {
    const size_t nchunks = orc_num_chunks(n);
    uint64_t pos = 0;
    for (size_t c = 0; c < nchunks; ++c) {
        const size_t base = c * (size_t)ORC_CHUNK_SYMS;
        const size_t m = n - base < ORC_CHUNK_SYMS ? n - base : ORC_CHUNK_SYMS;
        uint64_t freq[256];
        orc_aw_symbol_stat(sym + base, m, 256, 64, freq);
        orc_rp_system_codebook(freq, 256, 64, min_freq_threshold, 16, code + c * 256, len + c * 256);
        uint32_t off2[2];
        const uint64_t words = orc_rp_encode(sym + base, m, code + c * 256, len + c * 256,
                                             payload ? payload + pos : NULL, off2);
        if (chunk_off) chunk_off[c] = (uint32_t)pos;
        pos += words;
    }
    if (chunk_off) chunk_off[nchunks] = (uint32_t)pos;
    return pos;
}

uint64_t orc_rp_slot_blocks(const uint8_t* sym, size_t n, const uint8_t* len, size_t len_stride,
                            const uint32_t* payload, const uint32_t* chunk_off,
                            uint32_t* slot_payload, uint32_t* slot_off, uint32_t* words) {
    const size_t nchunks = (n + ORC_CHUNK_SYMS - 1) / ORC_CHUNK_SYMS;
    uint64_t at = 0;
    for (size_t c = 0; c < nchunks; ++c) {
        const size_t first = c * ORC_CHUNK_SYMS;
        const size_t m = n - first < ORC_CHUNK_SYMS ? n - first : ORC_CHUNK_SYMS;
        uint64_t bits = 0;
        for (size_t i = 0; i < m; ++i) bits += len[c * len_stride + sym[first + i]];
        const uint64_t slot = ORC_SLOT_WORDS(bits);
        const uint32_t w = chunk_off[c + 1] - chunk_off[c];
        if (w > slot) return UINT64_MAX;  /* the bound is broken: never seen, never expected */
        slot_off[c] = (uint32_t)at;
        words[c] = w;
        if (slot_payload) {
            memcpy(slot_payload + at, payload + chunk_off[c], 4u * (size_t)w);
            memset(slot_payload + at + w, 0, 4u * (size_t)(slot - w));
        }
        at += slot;
    }
    slot_off[nchunks] = (uint32_t)at;
    return at;
}

int orc_rp_decode_blocks(const uint32_t* payload, const uint32_t* chunk_off, size_t n,
                         const uint16_t* code, const uint8_t* len, uint8_t* sym_out) {
    const size_t nchunks = orc_num_chunks(n);
    for (size_t c = 0; c < nchunks; ++c) {
        const size_t base = c * (size_t)ORC_CHUNK_SYMS;
        const size_t m = n - base < ORC_CHUNK_SYMS ? n - base : ORC_CHUNK_SYMS;
        const uint32_t off2[2] = {0u, chunk_off[c + 1] - chunk_off[c]};
        if (orc_rp_decode(payload + chunk_off[c], off2, m, code + c * 256, len + c * 256,
                          sym_out + base) != 0)
            return -1;
    }
    return 0;
}

/* ---- paged stream (format version 2, per-block codebooks) ----------------------------------
 * The reference has neither a packed stream nor a decoder (top_encoder.scala:17-21 ends at the
 * {code, length, mode} beat; decoder/top_decoder.scala:6-10 is an empty bundle whose figure,
 * reports/figs/decoder.svg, shows a scheduler feeding parallel sub-decoders).  This is the second
 * stored form of a per-block stream: every page carries its own codebook record and can be found,
 * sized and decoded on its own, by four sub-decoders at once.  PARITY UNPINNED, as orc_rp_encode.
 *
 * A page is ORC_CHUNK_SYMS symbols with one codebook (orc_rp_system_codebook).  Its 256 "strings"
 * are the codes of 64 symbols each: string (q, l), q = 0..3, l = 0..63, holds the symbols of
 * rounds 8q .. 8q+7 of lane l (symbol index ((r * 64 + l) * 8 + j), r the round, j = 0..7; the
 * same ownership as orc_rp_encode), codes MSB first, symbols past the page's end left out.
 * Strings are stored bit-contiguous, without padding, in 128 regions: region r = qf * 64 + l
 * (qf = 0, 1) holds string (qf, l) read forward from the region's first bit and string (qf + 2, l)
 * read BACKWARD from its last bit, so one length per region -- R_r = the bits of both strings --
 * places 256 independent decoders.  Stream bit i of the strings area is bit 31 - (i mod 32) of its
 * word i / 32.
 *
 * Page layout, 32-bit little-endian words:
 *   0       head | mode << 9 | B << 10     head: see orc_rp_page_head; mode: 0 bypass / 1 Huffman in
 *                                          force; B: bits of all codes of the page
 *   1       entropy_q16
 *   2..9    u16[16]: number of ranks with code length 1 .. 16 (lengths never decrease with the
 *           rank, codes are canonical in rank order)
 *   10..    the first `head` symbols in rank order, one byte each, zero padded to a word; the
 *           other ranks are the remaining symbols in ascending order
 *   then    48 words: R_0 .. R_127, 12 bits each, field r at bits [12 r, 12 r + 12) of the table
 *           (bit b of the table = bit b mod 32 of its word b / 32)
 *   then    ceil(B / 32) words of strings
 * page_off[c] = first word of page c, page_off[nchunks] = total.  The size of a page follows from
 * its histogram and codebook alone (B = sum count x length), so every page's place is known before
 * any page is encoded. */
#define PAGE_HDR_WORDS 10u
#define PAGE_GAP_WORDS 48u
#define PAGE_SUBS 4u
#define PAGE_SUB_ROUNDS (ORC_ROUNDS / PAGE_SUBS)

uint32_t orc_rp_page_head(const uint8_t* rank2sym) {
    uint32_t p = 255;
    while (p > 0 && rank2sym[p - 1] < rank2sym[p]) --p;
    return p;
}

uint64_t orc_rp_page_words(uint32_t head, uint64_t bits) {
    return PAGE_HDR_WORDS + (head + 3u) / 4u + PAGE_GAP_WORDS + (bits + 31u) / 32u;
}

uint64_t orc_rp_pages_bound_words(size_t n) {
    return (uint64_t)orc_num_chunks(n) * orc_rp_page_words(255, (uint64_t)ORC_CHUNK_SYMS * 16u);
}

/* the low `nbits` (<= 32) bits of v, MSB first, at stream bit `at` of a zeroed area */
static void put_bits(uint32_t* w, uint64_t at, uint32_t v, unsigned nbits) {
    if (!nbits) return;
    const unsigned off = (unsigned)(at & 31u);
    const uint64_t x = (((uint64_t)v << (64 - nbits)) >> off); /* bit 63 = the word's first bit */
    w[at >> 5] |= (uint32_t)(x >> 32);
    if (off + nbits > 32) w[(at >> 5) + 1] |= (uint32_t)x;
}
/* `nbits` (<= 32) stream bits from bit `at` on, first bit highest */
static uint32_t get_bits(const uint32_t* w, uint64_t at, unsigned nbits) {
    if (!nbits) return 0;
    const unsigned off = (unsigned)(at & 31u);
    uint64_t x = (uint64_t)w[at >> 5] << 32;
    if (off + nbits > 32) x |= w[(at >> 5) + 1];
    return (uint32_t)((x << off) >> (64 - nbits));
}
static uint32_t bit_reverse(uint32_t v, unsigned nbits) {
    uint32_t r = 0;
    for (unsigned b = 0; b < nbits; ++b) r |= ((v >> b) & 1u) << (nbits - 1 - b);
    return r;
}
/* symbol index of symbol (rr, j) of string (q, l) */
static size_t page_symbol(unsigned q, unsigned l, unsigned rr, unsigned j) {
    return (((size_t)(q * PAGE_SUB_ROUNDS + rr)) * ORC_LANES + l) * ORC_GROUP + j;
}

/* region lengths (48 words at gap) and strings (behind them) of one page, into zeroed memory */
static void page_strings(const uint8_t* sym, size_t m, const uint16_t* code, const uint8_t* len,
                         uint32_t* gap) {
    uint32_t* str = gap + PAGE_GAP_WORDS;
    uint64_t at = 0;
    for (unsigned r = 0; r < 128; ++r) {
        const unsigned qf = r >> 6, l = r & 63u;
        unsigned R = 0;
        for (unsigned dir = 0; dir < 2; ++dir)
            for (unsigned rr = 0; rr < PAGE_SUB_ROUNDS; ++rr)
                for (unsigned j = 0; j < ORC_GROUP; ++j) {
                    const size_t i = page_symbol(qf + 2 * dir, l, rr, j);
                    if (i < m) R += len[sym[i]];
                }
        uint64_t fwd = at, bwd = at + R; /* next free bit going up / the bit behind the next one going down */
        for (unsigned rr = 0; rr < PAGE_SUB_ROUNDS; ++rr)
            for (unsigned j = 0; j < ORC_GROUP; ++j) {
                size_t i = page_symbol(qf, l, rr, j);
                if (i < m) {
                    const unsigned L = len[sym[i]];
                    put_bits(str, fwd, code[sym[i]], L);
                    fwd += L;
                }
                i = page_symbol(qf + 2, l, rr, j);
                if (i < m) {
                    const unsigned L = len[sym[i]];
                    bwd -= L;
                    put_bits(str, bwd, bit_reverse(code[sym[i]], L), L);
                }
            }
        for (unsigned b = 0; b < 12; ++b)
            if ((R >> b) & 1u) gap[(12u * r + b) >> 5] |= 1u << ((12u * r + b) & 31u);
        at += R;
    }
}

/* One page from its symbols and its codebook (code / len by symbol; rank2sym / len_by_rank in
 * rank order).  Returns the page's words; out may be NULL to size only. */
uint64_t orc_rp_encode_page(const uint8_t* sym, size_t m, const uint16_t* code, const uint8_t* len,
                            const uint8_t* rank2sym, const uint8_t* len_by_rank, uint32_t mode,
                            uint32_t entropy_q16, uint32_t* out) {
    uint64_t B = 0;
    for (size_t i = 0; i < m; ++i) B += len[sym[i]];
    const uint32_t head = orc_rp_page_head(rank2sym);
    const uint64_t words = orc_rp_page_words(head, B);
    if (!out) return words;
    memset(out, 0, (size_t)words * 4u);
    out[0] = head | ((mode & 1u) << 9) | ((uint32_t)B << 10);
    out[1] = entropy_q16;
    for (int p = 0; p < 256; ++p) {
        const unsigned L = len_by_rank[p];
        if (L >= 1 && L <= 16) out[2 + (L - 1) / 2] += 1u << (16u * ((L - 1) & 1u));
    }
    for (uint32_t p = 0; p < head; ++p) out[PAGE_HDR_WORDS + p / 4] |= (uint32_t)rank2sym[p] << (8u * (p & 3u));
    page_strings(sym, m, code, len, out + PAGE_HDR_WORDS + (head + 3u) / 4u);
    return words;
}

/* Per-block codebooks and pages for a whole stream.  mode_req: 0 bypass, 1 Huffman forced, 2 the
 * entropy decision (orc_aw_adaptive_mode of orc_rp_entropy_q16_u64), 3 best (orc_rp_best_mode).
 * code / len (optional, nchunks x 256 by raw symbol) receive the tables in force.  Returns the
 * total words (payload may be NULL to size only). */
uint64_t orc_rp_encode_pages(const uint8_t* sym, size_t n, uint64_t min_freq_threshold, int mode_req,
                             uint16_t* code_out, uint8_t* len_out, uint32_t* payload,
                             uint32_t* page_off) {
    const size_t nchunks = orc_num_chunks(n);
    uint64_t pos = 0;
    for (size_t c = 0; c < nchunks; ++c) {
        const size_t base = c * (size_t)ORC_CHUNK_SYMS;
        const size_t m = n - base < ORC_CHUNK_SYMS ? n - base : ORC_CHUNK_SYMS;
        uint64_t freq[256], sorted[256];
        uint16_t ids[256], code[256];
        uint8_t len[256], r2s[256], lbr[256];
        orc_aw_symbol_stat(sym + base, m, 256, 64, freq);
        const uint32_t ent = orc_rp_entropy_q16_u64(freq, 256);
        uint32_t mode = mode_req == 2 ? (uint32_t)orc_aw_adaptive_mode(ent) : mode_req == 3 ? 1u : (uint32_t)(mode_req & 1);
        if (mode == 1) {
            orc_rp_system_codebook(freq, 256, 64, min_freq_threshold, 16, code, len);
            if (mode_req == 3 && !orc_rp_best_mode(freq, len, 256)) mode = 0;
        }
        if (mode == 1) {
            orc_aw_symbol_sort(freq, 256, sorted, ids);
            for (int p = 0; p < 256; ++p) {
                r2s[p] = (uint8_t)ids[p];
                lbr[p] = len[ids[p]];
            }
        } else { /* bypass table, top_encoder.scala:131-137: code(i) = i, length = symWidth */
            for (int s = 0; s < 256; ++s) {
                code[s] = (uint16_t)s;
                len[s] = 8;
                r2s[s] = (uint8_t)s;
                lbr[s] = 8;
            }
        }
        if (code_out) memcpy(code_out + c * 256, code, sizeof code);
        if (len_out) memcpy(len_out + c * 256, len, sizeof len);
        if (page_off) page_off[c] = (uint32_t)pos;
        pos += orc_rp_encode_page(sym + base, m, code, len, r2s, lbr, mode, ent,
                                  payload ? payload + pos : NULL);
    }
    if (page_off) page_off[nchunks] = (uint32_t)pos;
    return pos;
}

/* The codebook of a stored page: code / len by raw symbol, mode, entropy.  Returns 0, or -1 when the
 * record is not a canonical order or the page's size does not match `words`. */
int orc_rp_page_codebook(const uint32_t* page, uint64_t words, uint16_t* code, uint8_t* len,
                         uint32_t* mode, uint32_t* entropy_q16) {
    if (words < PAGE_HDR_WORDS + PAGE_GAP_WORDS) return -1;
    const uint32_t head = page[0] & 0x1FFu, B = page[0] >> 10;
    if (head > 255 || (page[0] >> 29) != 0 || orc_rp_page_words(head, B) != words) return -1;
    uint8_t r2s[256], seen[256] = {0};
    for (uint32_t p = 0; p < head; ++p) {
        r2s[p] = (uint8_t)(page[PAGE_HDR_WORDS + p / 4] >> (8u * (p & 3u)));
        if (seen[r2s[p]]) return -1;
        seen[r2s[p]] = 1;
    }
    uint32_t p = head;
    for (int s = 0; s < 256; ++s)
        if (!seen[s]) r2s[p++] = (uint8_t)s;
    uint32_t nlen[17] = {0}, total = 0;
    uint64_t kraft = 0;
    for (unsigned L = 1; L <= 16; ++L) {
        nlen[L] = (page[2 + (L - 1) / 2] >> (16u * ((L - 1) & 1u))) & 0xFFFFu;
        total += nlen[L];
        kraft += (uint64_t)nlen[L] << (16 - L);
    }
    if (total != 256 || kraft > 65536) return -1;
    uint32_t next = 0; /* left-aligned 16-bit code of the next rank */
    p = 0;
    for (unsigned L = 1; L <= 16; ++L)
        for (uint32_t k = 0; k < nlen[L]; ++k, ++p) {
            code[r2s[p]] = (uint16_t)(next >> (16 - L));
            len[r2s[p]] = (uint8_t)L;
            next += 1u << (16 - L);
        }
    if (mode) *mode = (page[0] >> 9) & 1u;
    if (entropy_q16) *entropy_q16 = page[1];
    return 0;
}

/* direct table over the left-aligned 16-bit code space: length << 8 | symbol, 0 = no code */
static void page_code_table(const uint16_t* code, const uint8_t* len, uint16_t* tab) {
    memset(tab, 0, 65536 * sizeof(uint16_t));
    for (int s = 0; s < 256; ++s) {
        const unsigned L = len[s];
        if (L < 1 || L > 16) continue;
        const unsigned first = (unsigned)code[s] << (16 - L);
        for (unsigned e = 0; e < (1u << (16 - L)); ++e) tab[first + e] = (uint16_t)((L << 8) | s);
    }
}

/* the strings of one page of m symbols back into symbols; 0, or -1 when they do not decode */
static int page_decode_strings(const uint32_t* gap, uint32_t B, size_t m, const uint16_t* tab,
                               uint8_t* sym_out) {
    const uint32_t* str = gap + PAGE_GAP_WORDS;
    int rc = 0;
    uint64_t at = 0;
    for (unsigned r = 0; r < 128 && rc == 0; ++r) {
        unsigned R = 0;
        for (unsigned b = 0; b < 12; ++b) R |= ((gap[(12u * r + b) >> 5] >> ((12u * r + b) & 31u)) & 1u) << b;
        if (at + R > B) return -1;
        unsigned used[2] = {0, 0};
        for (unsigned dir = 0; dir < 2 && rc == 0; ++dir) {
            const unsigned q = (r >> 6) + 2 * dir, l = r & 63u;
            unsigned k = 0;
            for (unsigned rr = 0; rr < PAGE_SUB_ROUNDS && rc == 0; ++rr)
                for (unsigned j = 0; j < ORC_GROUP; ++j) {
                    const size_t i = page_symbol(q, l, rr, j);
                    if (i >= m) continue;
                    /* the next 16 bits of the string, zeros past the region */
                    const unsigned nb = R - k < 16 ? R - k : 16;
                    unsigned v = dir ? bit_reverse(get_bits(str, at + R - k - nb, nb), nb)
                                     : get_bits(str, at + k, nb);
                    v <<= 16 - nb;
                    const uint16_t e = tab[v];
                    const unsigned L = e >> 8;
                    if (L == 0 || k + L > R) {
                        rc = -1;
                        break;
                    }
                    sym_out[i] = (uint8_t)(e & 0xFF);
                    k += L;
                }
            used[dir] = k;
        }
        if (rc == 0 && used[0] + used[1] != R) rc = -1;
        at += R;
    }
    if (rc == 0 && at != B) rc = -1;
    return rc;
}

/* Returns 0, or -1 if a page does not decode (bad record, region lengths that do not add up to B,
 * no matching code, a region whose two strings do not meet exactly). */
int orc_rp_decode_pages(const uint32_t* payload, const uint32_t* page_off, size_t n,
                        uint8_t* sym_out) {
    const size_t nchunks = orc_num_chunks(n);
    uint16_t* tab = (uint16_t*)malloc(65536 * sizeof(uint16_t));
    int rc = 0;
    for (size_t c = 0; c < nchunks && rc == 0; ++c) {
        const size_t base = c * (size_t)ORC_CHUNK_SYMS;
        const size_t m = n - base < ORC_CHUNK_SYMS ? n - base : ORC_CHUNK_SYMS;
        const uint32_t* page = payload + page_off[c];
        const uint64_t words = (uint64_t)page_off[c + 1] - page_off[c];
        uint16_t code[256];
        uint8_t len[256];
        if (page_off[c + 1] < page_off[c] || orc_rp_page_codebook(page, words, code, len, NULL, NULL) != 0) {
            rc = -1;
            break;
        }
        page_code_table(code, len, tab);
        const uint32_t head = page[0] & 0x1FFu, B = page[0] >> 10;
        rc = page_decode_strings(page + PAGE_HDR_WORDS + (head + 3u) / 4u, B, m, tab, sym_out + base);
    }
    free(tab);
    return rc;
}

/* ---- paged stream with one codebook for the whole stream ("shared" pages) ---------------------
 * The same pages without a record: word 0 = B << 10 | 1 << 29 (bit 29 says "no record: the codes
 * are the stream's"), then the 48 words of region lengths, then the strings -- a page takes
 * 49 + ceil(B / 32) words.  The codebook travels beside the stream (container codebook section).
 * PARITY UNPINNED, as orc_rp_encode. */
#define PAGE_SHARED_FLAG (1u << 29)

uint64_t orc_rp_shared_page_words(uint64_t bits) { return 1u + PAGE_GAP_WORDS + (bits + 31u) / 32u; }

uint64_t orc_rp_encode_pages_shared(const uint8_t* sym, size_t n, const uint16_t* code,
                                    const uint8_t* len, uint32_t* payload, uint32_t* page_off) {
    const size_t nchunks = orc_num_chunks(n);
    uint64_t pos = 0;
    for (size_t c = 0; c < nchunks; ++c) {
        const size_t base = c * (size_t)ORC_CHUNK_SYMS;
        const size_t m = n - base < ORC_CHUNK_SYMS ? n - base : ORC_CHUNK_SYMS;
        uint64_t B = 0;
        for (size_t i = 0; i < m; ++i) B += len[sym[base + i]];
        const uint64_t words = orc_rp_shared_page_words(B);
        if (page_off) page_off[c] = (uint32_t)pos;
        if (payload) {
            uint32_t* out = payload + pos;
            memset(out, 0, (size_t)words * 4u);
            out[0] = ((uint32_t)B << 10) | PAGE_SHARED_FLAG;
            page_strings(sym + base, m, code, len, out + 1);
        }
        pos += words;
    }
    if (page_off) page_off[nchunks] = (uint32_t)pos;
    return pos;
}

int orc_rp_decode_pages_shared(const uint32_t* payload, const uint32_t* page_off, size_t n,
                               const uint16_t* code, const uint8_t* len, uint8_t* sym_out) {
    const size_t nchunks = orc_num_chunks(n);
    uint16_t* tab = (uint16_t*)malloc(65536 * sizeof(uint16_t));
    int rc = 0;
    page_code_table(code, len, tab);
    for (size_t c = 0; c < nchunks && rc == 0; ++c) {
        const size_t base = c * (size_t)ORC_CHUNK_SYMS;
        const size_t m = n - base < ORC_CHUNK_SYMS ? n - base : ORC_CHUNK_SYMS;
        const uint32_t* page = payload + page_off[c];
        const uint64_t words = (uint64_t)page_off[c + 1] - page_off[c];
        const uint32_t B = (page[0] >> 10) & 0x7FFFFu;
        if (page_off[c + 1] < page_off[c] || words < 1u + PAGE_GAP_WORDS ||
            (page[0] & 0x3FFu) != 0 || (page[0] >> 29) != 1u || orc_rp_shared_page_words(B) != words) {
            rc = -1;
            break;
        }
        rc = page_decode_strings(page + 1, B, m, tab, sym_out + base);
    }
    free(tab);
    return rc;
}

/* ---- container, wire format version 2 ------------------------------------------------------
 * Version 1 plus a placement field and, for slotted streams, the words array: header bytes 104
 * (u32 placement: 1 slotted, 2 paged), 108 (zero), 112 (u64 offset of the words array, 0 unless
 * slotted).  Sections: codebook (as version 1; absent, offset 0, for a paged stream of codebook
 * kind 1, whose pages carry the records), chunk offsets, chunk words (slotted only, u32[nchunks]), side bytes,
 * payload.  A dense stream (placement 0) is always a version-1 container. */
static void container2_offsets(uint64_t n, int layout, int kind, int placement, uint64_t payload_words,
                               uint64_t off[5], uint64_t* total) {
    const uint64_t nchunks = orc_num_chunks((size_t)n);
    uint64_t at = 128;
    off[0] = 0;
    if (placement != 2 || kind == 0) { /* (shared pages: the stream's one codebook) */
        off[0] = at;
        at = round16(at + (kind == 0 ? 768u : nchunks * 296u));
    }
    off[1] = at;
    at = round16(at + 4u * (nchunks + 1));
    off[4] = 0;
    if (placement == 1) {
        off[4] = at;
        at = round16(at + 4u * nchunks);
    }
    off[2] = 0;
    if (orc_side_bytes(n, layout) != 0 || layout == ORC_LAYOUT_BF16_EXP) {
        off[2] = at;
        at = round16(at + orc_side_bytes(n, layout));
    }
    off[3] = at;
    at += payload_words * 4u < 16u ? 16u : round16(payload_words * 4u);
    *total = at;
}

uint64_t orc_container2_bytes(uint64_t n, int layout, int kind, int placement, uint64_t payload_words) {
    uint64_t off[5], total;
    container2_offsets(n, layout, kind, placement, payload_words, off, &total);
    return total;
}

uint64_t orc_container2_pack(uint8_t* out, uint64_t n, int layout, int kind, int placement,
                             uint64_t payload_words, uint32_t mode, uint32_t entropy_q16,
                             const uint16_t* code, const uint8_t* len, const uint8_t* rank2sym,
                             const uint8_t* len_by_rank, const uint32_t* modes,
                             const uint32_t* entropies, const uint32_t* chunk_off,
                             const uint32_t* chunk_words, const uint8_t* side,
                             const uint32_t* payload) {
    if (placement == 0)
        return orc_container_pack(out, n, layout, kind, payload_words, mode, entropy_q16, code, len,
                                  rank2sym, len_by_rank, modes, entropies, chunk_off, side, payload);
    uint64_t off[5], total;
    container2_offsets(n, layout, kind, placement, payload_words, off, &total);
    const uint64_t nchunks = orc_num_chunks((size_t)n);
    memset(out, 0, (size_t)total);
    memcpy(out, "HUFHIP1", 8);
    le32(out + 8, 2);
    le32(out + 12, 128);
    le32(out + 16, (uint32_t)layout);
    le32(out + 20, (uint32_t)kind);
    le32(out + 24, ORC_CHUNK_SYMS);
    le32(out + 28, 16);
    le32(out + 32, kind == 0 ? mode : 0xFFFFFFFFu);
    le32(out + 36, kind == 0 ? entropy_q16 : 0u);
    le64(out + 40, n);
    le64(out + 48, nchunks);
    le64(out + 56, payload_words);
    for (int i = 0; i < 4; ++i) le64(out + 64 + 8 * i, off[i]);
    le64(out + 96, total);
    le32(out + 104, (uint32_t)placement);
    le64(out + 112, off[4]);
    if (off[0]) {
        uint8_t* p = out + off[0];
        if (kind == 0) {
            for (int s = 0; s < 256; ++s) {
                p[2 * s] = (uint8_t)(code[s] & 0xFF);
                p[2 * s + 1] = (uint8_t)(code[s] >> 8);
                p[512 + s] = len[s];
            }
        } else {
            for (uint64_t c = 0; c < nchunks; ++c, p += 296) {
                uint32_t per_len[17] = {0};
                memcpy(p, rank2sym + c * 256, 256);
                for (int r = 0; r < 256; ++r) {
                    const unsigned l = len_by_rank[c * 256 + r];
                    if (l >= 1 && l <= 16) per_len[l]++;
                }
                for (int l = 1; l <= 16; ++l) {
                    p[256 + 2 * (l - 1)] = (uint8_t)(per_len[l] & 0xFF);
                    p[256 + 2 * (l - 1) + 1] = (uint8_t)(per_len[l] >> 8);
                }
                le32(p + 288, modes[c]);
                le32(p + 292, entropies[c]);
            }
        }
    }
    for (uint64_t c = 0; c <= nchunks; ++c) le32(out + off[1] + 4 * c, chunk_off[c]);
    if (off[4])
        for (uint64_t c = 0; c < nchunks; ++c) le32(out + off[4] + 4 * c, chunk_words[c]);
    if (off[2] && n) memcpy(out + off[2], side, (size_t)orc_side_bytes(n, layout));
    for (uint64_t w = 0; w < payload_words; ++w) le32(out + off[3] + 4 * w, payload[w]);
    return total;
}
