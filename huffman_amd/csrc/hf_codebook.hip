// Table stages of the encoder pipeline, one wavefront per codebook:
//   SymbolSort            symbol_sort.scala:82-148
//   ShannonEntropy        shannon_entropy.scala:11-118, adaptive_threshold.scala:11-87
//   HuffmanTreeBuilder    tree_build.scala:17-140
//   tree_adjust           tree_adjust.scala:6-127
//   EntropyAwareTreeBuilder canonical codes  entropy_aware_tree_builder.scala:61-66, 116-157
// plus the decode-table build for the device codebook.
//
// Everything lives in one wavefront's registers and a ~11 KiB LDS block, and the kernels run one
// wavefront per workgroup, so a barrier costs nothing; N (alphabet size) is a run-time value, 2 <= N <= 256.
#include <stddef.h>

#include "hf_common.h"

namespace {

typedef unsigned long long u64;

struct CbLds {
    u64 w[512];        // node weights: leaves 0..N-1 in index order, internals N..2N-2
    u64 lq_w[256];     // leaf queue, ascending (weight, index)
    uint16_t lq_i[256];
    uint16_t left[512];
    uint16_t right[512];
    uint16_t parent[512];
    uint16_t depth[256];  // true leaf depths
    uint16_t adj[256];    // limited lengths
    uint16_t code[256];
    uint16_t rank2sym[256];
    uint32_t hist[20];
    uint32_t cum[20];
    uint32_t misc[8];
};

// Bitonic network over 256 distinct keys, 4 per lane (element e = lane * 4 + i), result descending.
// Strides 1 and 2 stay in a lane's registers.  Across lanes nothing goes through the LDS crossbar
// and nothing needs an address register: the partner's key comes by a data-parallel-primitive move
// for lane distances 1, 2 (quad permutes), 4 (half-row mirror, then quad reverse: l ^ 7 ^ 3) and
// 8 (row rotate by 8 = l ^ 8), and by the row / half-wave swap instructions of gfx950 for 16 and
// 32.  Which lanes keep the larger key of a pair is a property of the lane number alone, so for
// every stage it is a 64-bit constant: it sits in a scalar register pair and selects between the
// pair's maximum and minimum.
constexpr unsigned long long sort_keep_mask(const int kk, const int lj) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (((l & lj) == 0) == (((l * 4) & kk) == 0)) m |= 1ull << l;
    return m;
}
constexpr unsigned long long sort_asc_mask(const int kk, const int i) {
    unsigned long long m = 0;  // lanes whose element i sits in an ascending run of this phase
    for (int l = 0; l < 64; ++l)
        if (((l * 4 + i) & kk) != 0) m |= 1ull << l;
    return m;
}

// the value lane (l ^ LJ) holds, LJ = 1, 2, 4, 8
template <int LJ>
__device__ __forceinline__ uint32_t lane_xor(const uint32_t v) {
    static_assert(LJ == 1 || LJ == 2 || LJ == 4 || LJ == 8, "within a row of 16 lanes");
    if constexpr (LJ == 1) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);  // quad_perm 1,0,3,2
    } else if constexpr (LJ == 2) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  // quad_perm 2,3,0,1
    } else if constexpr (LJ == 4) {
        const int m = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
        return (uint32_t)__builtin_amdgcn_update_dpp(0, m, 0x1B, 0xF, 0xF, true);       // quad_perm 3,2,1,0
    } else {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);  // row_ror:8
    }
}

// one compare-exchange of the keys a lane holds with those of lane ^ LJ (phase KK), 32-bit keys
template <int KK, int LJ>
__device__ __forceinline__ void sort_cross_u32(uint32_t (&k)[4]) {
    constexpr unsigned long long keep = sort_keep_mask(KK, LJ);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t a, b;  // LJ < 16: own key, partner's key.  LJ >= 16: the two keys of the pair in
                        // the same order in both lanes
        if constexpr (LJ >= 16) {
            const auto sw = LJ == 16 ? __builtin_amdgcn_permlane16_swap(k[i], k[i], false, false)
                                     : __builtin_amdgcn_permlane32_swap(k[i], k[i], false, false);
            a = sw[0], b = sw[1];
        } else {
            a = k[i], b = lane_xor<LJ>(k[i]);
        }
        // the larger key where `keep` says so, the smaller elsewhere: maximum, minimum and a
        // select on the constant -- no compare result travels through a scalar register.  (Reading
        // the partner's key through the permutation of the max / min instructions themselves, one
        // instruction less per key, measured slower: 2430 against 2094 cycles per sort.)
        const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
        uint32_t r;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(lo), "v"(hi), "s"(keep));
        k[i] = r;
    }
}
// ... and with 64-bit keys
template <int KK, int LJ>
__device__ __forceinline__ void sort_cross_u64(u64 (&k)[4], const int lane) {
    constexpr unsigned long long keep = sort_keep_mask(KK, LJ);
    const bool keeps_larger = (keep >> lane) & 1ull;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t lo = (uint32_t)k[i], hi = (uint32_t)(k[i] >> 32);
        u64 a, b;
        if constexpr (LJ >= 16) {
            const auto sl = LJ == 16 ? __builtin_amdgcn_permlane16_swap(lo, lo, false, false)
                                     : __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
            const auto sh = LJ == 16 ? __builtin_amdgcn_permlane16_swap(hi, hi, false, false)
                                     : __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
            a = ((u64)sh[0] << 32) | sl[0];
            b = ((u64)sh[1] << 32) | sl[1];
        } else {
            a = k[i];
            b = ((u64)lane_xor<LJ>(hi) << 32) | lane_xor<LJ>(lo);
        }
        k[i] = ((a > b) != keeps_larger) ? b : a;
    }
}

// strides 1 and 2: both keys in this lane's registers
template <int KK, int J>
__device__ __forceinline__ void sort_local_u32(uint32_t (&k)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if ((i & J) == 0) {
            // descending run: the larger key first; ascending run: the smaller
            const unsigned long long asc = sort_asc_mask(KK, i);
            const uint32_t a = k[i], b = k[i | J];
            const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
            uint32_t na, nb;
            asm("v_cndmask_b32_e64 %0, %2, %3, %4\n\t"
                "v_cndmask_b32_e64 %1, %3, %2, %4"
                : "=&v"(na), "=&v"(nb)
                : "v"(hi), "v"(lo), "s"(asc));
            k[i] = na;
            k[i | J] = nb;
        }
    }
}
template <int KK, int J>
__device__ __forceinline__ void sort_local_u64(u64 (&k)[4], const int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if ((i & J) == 0) {
            const bool desc = (((lane * 4 + i) & KK) == 0);
            const u64 a = k[i], b = k[i | J];
            const bool sw = desc ? (a < b) : (a > b);
            k[i] = sw ? b : a;
            k[i | J] = sw ? a : b;
        }
    }
}

// phase KK of the network: strides KK / 2 .. 1
template <int KK, int J>
__device__ __forceinline__ void sort_steps_u32(uint32_t (&k)[4]) {
    if constexpr (J >= 4) sort_cross_u32<KK, J / 4>(k);
    else sort_local_u32<KK, J>(k);
    if constexpr (J > 1) sort_steps_u32<KK, J / 2>(k);
}
template <int KK>
__device__ __forceinline__ void sort_phases_u32(uint32_t (&k)[4]) {
    if constexpr (KK > 2) sort_phases_u32<KK / 2>(k);
    sort_steps_u32<KK, KK / 2>(k);
}
template <int KK, int J>
__device__ __forceinline__ void sort_steps_u64(u64 (&k)[4], const int lane) {
    if constexpr (J >= 4) sort_cross_u64<KK, J / 4>(k, lane);
    else sort_local_u64<KK, J>(k, lane);
    if constexpr (J > 1) sort_steps_u64<KK, J / 2>(k, lane);
}
template <int KK>
__device__ __forceinline__ void sort_phases_u64(u64 (&k)[4], const int lane) {
    if constexpr (KK > 2) sort_phases_u64<KK / 2>(k, lane);
    sort_steps_u64<KK, KK / 2>(k, lane);
}

__device__ __forceinline__ void sort256_desc_u32(uint32_t (&k)[4], const int lane) {
    (void)lane;
    sort_phases_u32<256>(k);
}
__device__ __forceinline__ void sort256_desc(u64 (&k)[4], const int lane) {
    sort_phases_u64<256>(k, lane);
}

// SymbolSort with identities: total order (count desc, index asc), which is the order the
// RTL's stable odd-even transposition (swap only when a < b, symbol_sort.scala:131-135)
// produces for the values.  Counts must be below 2^56.
__device__ void stage_sort(const u64* __restrict__ freq, const int N, u64* sorted,
                           uint16_t* rank2sym, const int lane) {
    u64 k[4];
    bool small = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane * 4 + i;
        const u64 f = e < N ? freq[e] : 0ull;
        small = small && f < (1ull << 24);
        k[i] = (f << 8) | (u64)(255 - e);
    }
    if (__ballot(!small) == 0ull) {
        // counts of a block of up to 16 Mi symbols: the keys fit 32 bits, half the moves
        uint32_t k32[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) k32[i] = (uint32_t)k[i];
        sort256_desc_u32(k32, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) k[i] = k32[i];
    } else {
        sort256_desc(k, lane);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = lane * 4 + i;
        if (r < N) {
            sorted[r] = k[i] >> 8;
            rank2sym[r] = (uint16_t)(255 - (int)(k[i] & 0xFF));
        }
    }
}

// ShannonEntropy.sCalculateEntropy (shannon_entropy.scala:67-93) at W = 32, every index
// visited once; the 32-bit accumulator makes the sum order-free.  Totals above 32 bits are
// brought into range by shifting every count right first.
__device__ uint32_t stage_entropy(const u64* __restrict__ freq, const int N, const int lane) {
    u64 f[4];
    u64 tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i;
        f[i] = e < N ? freq[e] : 0ull;
        tot += f[i];
    }
    tot = hf_wave_sum_u64(tot);
    const int bits = 64 - __clzll(tot | 1ull);
    const int sh = bits > 32 ? bits - 32 : 0;
    u64 t2 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[i] = (f[i] >> sh) & 0xFFFFFFFFull;
        t2 += f[i];
    }
    const u64 total = hf_wave_sum_u64(t2) & 0xFFFFFFFFull;  // shannon_entropy.scala:62
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (f[i] != 0 && total != 0) {
            const u64 prob = (f[i] << 16) / total;                   // :72
            const uint32_t lo = (uint32_t)prob;
            const uint32_t k = lo ? (uint32_t)__clz(lo) : 31u;       // :52-54
            const u64 pow2 = 1ull << ((k + 9u) & 31u);               // :78, 5-bit shift amount
            const u64 xk = (prob * (u64)k) >> 16;                    // :81
            const u64 approx = (xk + (prob << 1) - pow2) & ((1ull << 49) - 1ull);  // :84-87
            acc += (uint32_t)((prob * approx) >> 16);                // :90-91
        }
    }
    acc = hf_wave_sum_u32(acc);
    return 0x10000u - acc;  // :101
}

// adaptive_threshold.scala:26-27, :65-79
__device__ __forceinline__ uint32_t entropy_mode(const uint32_t e) {
    return (e < 0x8000u) ? 1u : ((e > 0x60000u) ? 0u : 1u);
}

// HuffmanTreeBuilder.  The RTL picks, every cycle, the lowest-index node of minimum weight
// and then the lowest-index remaining node of minimum weight (tree_build.scala:47-72) and
// appends their parent (:113-125).  With leaf indices below internal indices and internal
// weights created in non-decreasing order, that is a two-queue merge which prefers the leaf
// queue on ties, leaves ordered by (weight asc, index asc).  Requires sum(weights) < 2^56.
__device__ void stage_tree(CbLds& L, const int N, const int lane) {
    {
        u64 k[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = lane * 4 + i;
            k[i] = e < N ? ~((L.w[e] << 8) | (u64)e) : 0ull;
        }
        sort256_desc(k, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = lane * 4 + i;
            if (q < N) {
                const u64 v = ~k[i];
                L.lq_w[q] = v >> 8;
                L.lq_i[q] = (uint16_t)(v & 0xFF);
            }
        }
    }
    for (int e = lane; e < 2 * N - 1; e += 64) {
        L.left[e] = 0;
        L.right[e] = 0;
        L.parent[e] = 0;
        if (e >= N) L.w[e] = 0;
    }
    __syncthreads();
    if (lane == 0) {
        int li = 0, ii = N;
        u64 lw = L.lq_w[0], iw = 0;
        for (int k = N; k < 2 * N - 1; ++k) {
            int idx[2];
            u64 wt[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const bool leaf = (li < N) && (ii >= k || lw <= iw);
                if (leaf) {
                    idx[p] = L.lq_i[li];
                    wt[p] = lw;
                    ++li;
                    lw = li < N ? L.lq_w[li] : 0;
                } else {
                    idx[p] = ii;
                    wt[p] = iw;
                    ++ii;
                    iw = ii < k ? L.w[ii] : 0;
                }
            }
            const u64 ws = wt[0] + wt[1];
            L.w[k] = ws;
            L.left[k] = (uint16_t)idx[0];
            L.right[k] = (uint16_t)idx[1];
            L.parent[idx[0]] = (uint16_t)k;
            L.parent[idx[1]] = (uint16_t)k;
            if (ii == k) iw = ws;
        }
    }
    __syncthreads();
    // true leaf depth = number of parent links up to the root (the RTL's depths port
    // returns the never-written leaf entries of depthArr, tree_build.scala:81)
    const int root = 2 * N - 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i;
        if (e < N) {
            int p = e, d = 0;
            while (p != root) {
                p = L.parent[p];
                ++d;
            }
            L.depth[e] = (uint16_t)d;
        }
    }
    __syncthreads();
}

// tree_adjust (tree_adjust.scala:44-119) with Kraft-correct moves works on the level counts alone:
// BuildHist (depths above lmax clamp to lmax) gives the number of leaves per level, Migrate and
// FillHoles move leaves between levels, Assign hands the levels out by index.  The level counts
// live one per lane (lane d holds level d) and the loop control in scalar registers, so no step
// waits for a memory access.
//
// Migrate: the deepest level below lmax gives a leaf to the level under it until the Kraft sum
// fits.  FillHoles: leaves move up while the freed code space allows, deepest level first.
// Returns the number of Migrate steps.
__device__ __forceinline__ uint32_t adjust_levels(uint32_t& h, const int lmax, const int lane) {
    uint32_t kraft = hf_wave_sum_u32(lane <= lmax ? h << (lmax - lane) : 0u);
    kraft = (uint32_t)__builtin_amdgcn_readfirstlane((int)kraft);
    const uint32_t full = 1u << lmax;
    uint32_t steps = 0;
    while (kraft > full) {
        // the deepest level below lmax that still has a leaf (level 0 if none has)
        const u64 m = __ballot(h != 0 && lane >= 1 && lane < lmax);
        const int d = m ? 63 - __builtin_clzll(m) : 0;
        // A leaf that leaves level d is, step after step, the deepest leaf below lmax again and
        // walks down to lmax: lmax - d steps that take 2^(lmax-d) - 1 off the Kraft sum, provided
        // the sum was at least that much too large.  While it is, and level d has leaves, the
        // walks repeat unchanged, so a safe number of them (excess >> (lmax - d), never more than
        // the exact quotient) is taken at once; what is left goes step by step.
        const uint32_t hd = (uint32_t)__builtin_amdgcn_readlane((int)h, d);
        const uint32_t walks = (kraft - full) >> (lmax - d);
        const uint32_t n = hd < walks ? hd : walks;
        if (n != 0) {
            if (lane == d) h -= n;
            if (lane == lmax) h += n;
            kraft -= n * ((1u << (lmax - d)) - 1u);
            steps += n * (uint32_t)(lmax - d);
        } else {
            if (lane == d) h -= 1;
            if (lane == d + 1) h += 1;
            kraft -= 1u << (lmax - d - 1);
            ++steps;
        }
    }
    uint32_t holes = full - kraft;
    for (int d = lmax; d >= 2 && holes != 0; --d) {
        const uint32_t can = holes >> (lmax - d);  // the freed space in units of a level-d leaf
        const uint32_t hd = (uint32_t)__builtin_amdgcn_readlane((int)h, d);
        const uint32_t mv = hd < can ? hd : can;
        if (lane == d) h -= mv;
        if (lane == d - 1) h += mv;
        holes -= mv << (lmax - d);
    }
    return steps;
}

// Assign: the depth of index p is the number of levels whose cumulative count is <= p.  Depths
// therefore never decrease with p, and the canonical code of p follows in the same pass:
// count/codeStart as entropy_aware_tree_builder.scala:118-127 (count[0] taken as 0),
// code(p) = codeStart[len_p] + #{j < p : len_j == len_p} -- the assignment the RTL's
// commented-out loop intends (:105-111) -- where the second term is p minus the number of leaves
// on the levels above len_p.  `p` are the four indices this lane handles.
__device__ __forceinline__ void assign_codes(const uint32_t h, const int lmax, const int lane,
                                             const uint32_t (&p)[4], uint32_t (&dep)[4],
                                             uint32_t (&code)[4]) {
    // cumulative counts, lane d = number of leaves at levels <= d
    uint32_t cum = h;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const uint32_t u = __shfl_up(cum, o);
        if (lane >= o) cum += u;
    }
    uint32_t above[4] = {0, 0, 0, 0}, start[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) dep[i] = 0;
    uint32_t cs = 0;  // codeStart[q] on entry to iteration q
    for (int q = 0; q <= lmax; ++q) {
        const uint32_t cq = (uint32_t)__builtin_amdgcn_readlane((int)cum, q);
        const uint32_t hq = (uint32_t)__builtin_amdgcn_readlane((int)h, q);
        const uint32_t cs_next = q == 0 ? 0u : ((cs + hq) << 1) & 0xFFFFu;  // codeStart[q + 1]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = cq <= p[i];
            dep[i] += in;
            above[i] = in ? cq : above[i];
            start[i] = in ? cs_next : start[i];
        }
        cs = cs_next;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) code[i] = dep[i] ? (start[i] + p[i] - above[i]) & 0xFFFFu : 0u;
}

// The stage over an LDS block: L.depth[N] -> L.adj[N] (limited lengths) and L.code[N].
template <class LDS>
__device__ uint32_t stage_adjust(LDS& L, const int N, const int lmax, const int lane) {
    if (lane < 20) L.hist[lane] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i;
        if (e < N) {
            const int d = L.depth[e] > lmax ? lmax : L.depth[e];
            atomicAdd(&L.hist[d], 1u);
        }
    }
    __syncthreads();
    uint32_t h = lane <= lmax ? L.hist[lane] : 0u;
    const uint32_t steps = adjust_levels(h, lmax, lane);
    const uint32_t p[4] = {(uint32_t)lane, (uint32_t)lane + 64u, (uint32_t)lane + 128u,
                           (uint32_t)lane + 192u};
    uint32_t dep[4], code[4];
    assign_codes(h, lmax, lane, p, dep, code);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if ((int)p[i] < N) {
            L.adj[p[i]] = (uint16_t)dep[i];
            L.code[p[i]] = (uint16_t)code[i];
        }
    }
    __syncthreads();
    return steps;
}

// ---- module-level stages behind the C ABI's host-table calls ---------------------------------
__global__ __launch_bounds__(64) void hf_k_stage(hf_stage_io* io, const int N, const int op,
                                                 const int lmax, const u64 thr) {
    __shared__ CbLds L;
    const int lane = hf_lane();
    if (op == HF_OP_SORT) {
        stage_sort(reinterpret_cast<const u64*>(io->freq), N, reinterpret_cast<u64*>(io->sorted),
                   io->rank2sym, lane);
    } else if (op == HF_OP_ENTROPY) {
        const uint32_t e = stage_entropy(reinterpret_cast<const u64*>(io->freq), N, lane);
        if (lane == 0) {
            io->entropy_q16 = e;
            io->mode = entropy_mode(e);
        }
    } else if (op == HF_OP_TREE || op == HF_OP_CODEBOOK) {
        for (int e = lane; e < N; e += 64) {
            const u64 f = io->freq[e];
            // EntropyAwareTreeBuilder.sFilter, entropy_aware_tree_builder.scala:61-66
            L.w[e] = (op == HF_OP_CODEBOOK && f < thr) ? 0ull : f;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        stage_tree(L, N, lane);
        if (op == HF_OP_TREE) {
            for (int e = lane; e < 2 * N - 1; e += 64) {
                io->weights[e] = L.w[e];
                io->lefts[e] = L.left[e];
                io->rights[e] = L.right[e];
            }
            for (int e = lane; e < N; e += 64) io->depths[e] = L.depth[e];
            if (lane == 0) io->root = (uint32_t)(2 * N - 2);
        } else {
            const uint32_t steps = stage_adjust(L, N, lmax, lane);
            for (int e = lane; e < N; e += 64) {
                io->code[e] = L.code[e];
                io->len[e] = (uint8_t)L.adj[e];
            }
            if (lane == 0) io->migrate_steps = steps;
        }
    } else if (op == HF_OP_ADJUST) {
        for (int e = lane; e < N; e += 64) L.depth[e] = io->depths[e];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        const uint32_t steps = stage_adjust(L, N, lmax, lane);
        for (int e = lane; e < N; e += 64) io->adj[e] = L.adj[e];
        if (lane == 0) io->migrate_steps = steps;
    }
    if (lane == 0) io->status = 0;
}

// HuffmanTreeBuilder for the system kernel (N = 256): the same two-queue merge as stage_tree, but
// only what the next stage needs is kept.  That stage works on the number of leaves per level,
// which follows from the parent links of the internal nodes alone (they leave the FIFO in creation
// order, so a tree level is a contiguous range of them; see hf_k_chunks_back).  Weights are all
// that matters for it: the leaf queue is the filtered counts in rank order read backwards (no
// second sort), and no child link, leaf parent or leaf depth is written.  Returns the leaves per
// level, level d in lane d, depths above lmax counted at lmax.
//
// Synchronisation inside one wavefront: its LDS operations complete in order, so a fence on both
// sides of a wave barrier is all a store by one lane needs to be seen by the others.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ u64 uniform_u64(const u64 v) {
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    return ((u64)hi << 32) | (u64)lo;
}

// The same merge, whole runs of it at a time.  The two-queue merge takes its two nodes from the
// leaf queue while the second of them is no heavier than the oldest internal node waiting (ties go
// to the leaves), and that node does not change while only leaves are taken: a run of leaf-leaf
// merges is as long as the number of leaf pairs whose second weight stays within that bound, which
// one ballot tells, and the run's parents are made by as many lanes at once.  Likewise a run of
// merges that take two internal nodes each lasts while the second of a pair is lighter than the
// leaf at the head of its queue (and exists already).  Everything else -- a leaf with an internal
// node -- is a single step.  The sequence of merges, hence every weight and parent link, is the
// serial loop's; a histogram with many equal or zero counts needs a few dozen steps instead of
// 255, a Fibonacci-like one as many as before.  One wavefront; uses only wave-level
// synchronisation, so it may run while other wavefronts of the workgroup wait at a barrier.
__device__ __forceinline__ uint32_t system_tree_levels_runs(CbLds& L, const u64* sorted,
                                                            const u64 thr, const int lmax,
                                                            const int lane) {
    constexpr int N = 256;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = lane + 64 * i;
        const u64 f = sorted[N - 1 - q];
        L.lq_w[q] = f < thr ? 0ull : f;  // sFilter, entropy_aware_tree_builder.scala:61-66
    }
    wave_sync();
    // the very first merge finds no internal node waiting and takes two leaves; from then on there
    // is always one (every merge makes one)
    if (lane == 0) L.w[N] = L.lq_w[0] + L.lq_w[1];
    wave_sync();
    int li = 2, ii = N, k = N + 1;  // next leaf, next internal node to take, next node to make
    while (k < 2 * N - 1) {
        // Everything a step can need is read in one go (one LDS round trip per step): the two
        // leaves and the two internal nodes pair `lane` of a run would take.  Lane 0's are the
        // heads of the two queues.
        const int lidx = li + 2 * lane, iidx = ii + 2 * lane;
        const u64 la = L.lq_w[lidx < N ? lidx : N - 1];
        const u64 lb = L.lq_w[lidx + 1 < N ? lidx + 1 : N - 1];
        const u64 ia = L.w[iidx < k ? iidx : N];
        const u64 ib = L.w[iidx + 1 < k ? iidx + 1 : N];
        const bool have_leaf = li < N;
        const u64 lw = uniform_u64(la);  // (clamped to the last leaf when there is none)
        const u64 H = uniform_u64(ia);
        {   // leaf-leaf run: pair j takes leaves li + 2j and li + 2j + 1
            const bool ok = lidx + 1 < N && lb <= H;
            const u64 m = __ballot(ok);
            const int t = m == ~0ull ? 64 : (int)__builtin_ctzll(~m);
            if (t > 0) {
                if (lane < t) L.w[k + lane] = la + lb;
                li += 2 * t;
                k += t;
                wave_sync();
                continue;
            }
        }
        if (have_leaf && lw <= H) {
            // a leaf, then the internal node (the next leaf is heavier than it, or there is none)
            if (lane == 0) {
                L.parent[ii] = (uint16_t)k;
                L.w[k] = lw + H;
            }
            ++li, ++ii, ++k;
            wave_sync();
            continue;
        }
        {   // internal-internal run over the nodes that exist now: pair j takes ii + 2j, ii + 2j + 1
            const bool ok = iidx + 1 < k && (!have_leaf || ib < lw);
            const u64 m = __ballot(ok);
            const int t = m == ~0ull ? 64 : (int)__builtin_ctzll(~m);
            if (t > 0) {
                if (lane < t) {
                    L.w[k + lane] = ia + ib;
                    L.parent[iidx] = (uint16_t)(k + lane);
                    L.parent[iidx + 1] = (uint16_t)(k + lane);
                }
                ii += 2 * t;
                k += t;
                wave_sync();
                continue;
            }
        }
        // the internal node, then a leaf (no second internal node lighter than it)
        if (lane == 0) {
            L.parent[ii] = (uint16_t)k;
            L.w[k] = H + lw;
        }
        ++ii, ++li, ++k;
        wave_sync();
    }
    uint32_t par[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) par[i] = lane + 64 * i < N - 2 ? L.parent[N + lane + 64 * i] : 0u;
    uint32_t lo = 2 * N - 2, hi = 2 * N - 2, up = 1, h = 0, placed = 0;  // the root alone
    for (uint32_t d = 1;; ++d) {
        if (d == (uint32_t)lmax) {
            if (lane == lmax) h += (uint32_t)N - placed;
            break;
        }
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            cnt += (uint32_t)__popcll(__ballot(lane + 64 * i < N - 2 && par[i] >= lo && par[i] <= hi));
        const uint32_t lv = 2u * up - cnt;
        if ((uint32_t)lane == d) h += lv;
        placed += lv;
        if (cnt == 0) break;
        hi = lo - 1u;
        lo -= cnt;
        up = cnt;
    }
    return h;
}

// ---- decode tables of a device codebook --------------------------------------------------------
// Built in LDS by one workgroup of 256 threads from the encoder table and written out once.
// First level: 2^12 entries indexed by the next 12 bits.  Codes longer than 12 bits share their
// 12-bit prefix with at most 15 other codes; every distinct prefix gets a 16-entry second-level
// table indexed by the following 4 bits, and its first-level entry carries that table's number
// with length 0.  Works for any prefix-free code, canonical or not.  The multi-symbol table
// follows from the first level: the entry for 12 bits that start with a code of len0 bits looks
// the remaining 12 - len0 bits up again (left-aligned, zero-filled), and resolves a second symbol
// when that entry's code fits into them.
struct TabLds {
    uint32_t enc[256];
    uint16_t lut[HF_LUT_SIZE];
    uint16_t sub[256 * 16];
    uint16_t subidx[HF_LUT_SIZE];
    uint32_t lut2[HF_LUT_SIZE];
    uint32_t cnt[4];     // second-level tables named by each wavefront's threads
    uint32_t max_len, hits;
};
static_assert(offsetof(hf_cb_dev, lut) % 16 == 0 && offsetof(hf_cb_dev, sub) % 16 == 0 &&
              offsetof(hf_cb_dev, lut2) % 16 == 0, "tables are written 16 bytes at a time");

// T.enc holds the encoder table (code | len << 16 by symbol); every thread of the 256 calls.
__device__ void build_tables(TabLds& T, hf_cb_dev* cb, const int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    if (tid == 0) T.max_len = T.hits = 0;
    for (int i = tid; i < HF_LUT_SIZE; i += 256) {
        T.lut[i] = 0;
        T.sub[i] = 0;
        T.subidx[i] = 0xFFFFu;
    }
    __syncthreads();
    const uint32_t my = T.enc[tid];
    const uint32_t my_len = my >> 16;
    const uint32_t my_code = my & 0xFFFFu;
    {
        uint32_t mx = my_len;  // the longest code: one atomic per wavefront
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const uint32_t v = __shfl_xor(mx, o);
            mx = v > mx ? v : mx;
        }
        if (lane == 0) atomicMax(&T.max_len, mx);
    }
    // First level.  A code of 6..12 bits covers at most 64 entries and its thread writes them, 16
    // bytes at a time from 8 entries on (a span starts at a multiple of its size).  The span of a
    // shorter code (128..2048 entries) is written by the thread's whole wavefront, eight entries
    // per lane and store, the wavefront's short codes one after the other (what a span needs
    // travels by readlane, nothing is read back from LDS).
    const bool fits = my_len >= 1 && my_len <= HF_LUT_BITS;
    const uint32_t span = fits ? 1u << (HF_LUT_BITS - my_len) : 0u;
    const uint32_t first = fits ? (my_code << (HF_LUT_BITS - my_len)) & (HF_LUT_SIZE - 1) : 0u;
    const uint32_t val = ((uint32_t)tid << 8) | my_len;
    const uint32_t val2 = val | (val << 16);
    if (span >= 8 && span <= 64) {
        for (uint32_t k = 0; k < span; k += 8)
            *reinterpret_cast<uint4*>(&T.lut[first + k]) = make_uint4(val2, val2, val2, val2);
    } else if (span == 4) {
        *reinterpret_cast<uint2*>(&T.lut[first]) = make_uint2(val2, val2);
    } else if (span == 2) {
        *reinterpret_cast<uint32_t*>(&T.lut[first]) = val2;
    } else if (span == 1) {
        T.lut[first] = (uint16_t)val;
    }
    for (u64 wide = __ballot(span > 64); wide != 0; wide &= wide - 1) {
        const int src = (int)__builtin_ctzll(wide);
        const uint32_t w_first = (uint32_t)__builtin_amdgcn_readlane((int)first, src);
        const uint32_t w_span = (uint32_t)__builtin_amdgcn_readlane((int)span, src);
        const uint32_t w_val2 = (uint32_t)__builtin_amdgcn_readlane((int)val2, src);
        for (uint32_t k = 8u * (uint32_t)lane; k < w_span; k += 512u)
            *reinterpret_cast<uint4*>(&T.lut[w_first + k]) = make_uint4(w_val2, w_val2, w_val2, w_val2);
    }
    const bool is_long = my_len > HF_LUT_BITS && my_len <= HF_MAX_CODE_LEN;
    const uint32_t my_prefix = is_long ? (my_code >> (my_len - HF_LUT_BITS)) & (HF_LUT_SIZE - 1) : 0;
    if (is_long) T.subidx[my_prefix] = 0;  // mark (every writer stores the same value)
    __syncthreads();
    // number the marked prefixes in ascending order: thread t owns entries 16t .. 16t+15, read as
    // two 16-byte pieces
    uint32_t own[8];
    {
        const uint4 p0 = reinterpret_cast<const uint4*>(T.subidx)[tid * 2];
        const uint4 p1 = reinterpret_cast<const uint4*>(T.subidx)[tid * 2 + 1];
        own[0] = p0.x, own[1] = p0.y, own[2] = p0.z, own[3] = p0.w;
        own[4] = p1.x, own[5] = p1.y, own[6] = p1.z, own[7] = p1.w;
    }
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) c += ((own[k] & 0xFFFFu) == 0) + ((own[k] >> 16) == 0);
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) T.cnt[wv] = incl;
    __syncthreads();
    uint32_t base = incl - c;
    for (int w = 0; w < wv; ++w) base += T.cnt[w];
    const uint32_t n_sub = T.cnt[0] + T.cnt[1] + T.cnt[2] + T.cnt[3];
    if (c != 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = tid * 16 + k;
            if (((own[k >> 1] >> (16 * (k & 1))) & 0xFFFFu) == 0) {
                T.subidx[i] = (uint16_t)base;
                T.lut[i] = (uint16_t)(base << 8);  // length 0 = escape to table `base`
                ++base;
            }
        }
    }
    __syncthreads();
    if (is_long) {
        const uint32_t rem = my_len - HF_LUT_BITS;  // 1 .. 4 bits beyond the prefix
        const uint32_t f2 = (my_code & ((1u << rem) - 1u)) << (4 - rem);
        const uint32_t span2 = 1u << (4 - rem);
        const uint32_t t = T.subidx[my_prefix];
        for (uint32_t k = 0; k < span2; ++k) T.sub[t * 16 + f2 + k] = (uint16_t)val;
    }
    __syncthreads();
    // The multi-symbol table.  The second lookup is made for every entry (an escape entry has
    // length 0 and looks itself up), so that the sixteen entries of a thread are sixteen
    // independent pairs of reads.
    uint32_t hits = 0;
    uint32_t e0[16], e1[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) e0[j] = T.lut[tid + 256 * j];
#pragma unroll
    for (int j = 0; j < 16; ++j)
        e1[j] = T.lut[((uint32_t)(tid + 256 * j) << (e0[j] & 0xFFu)) & (HF_LUT_SIZE - 1)];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t len0 = e0[j] & 0xFFu, sym0 = e0[j] >> 8;
        const uint32_t len1 = e1[j] & 0xFFu;
        const bool two = len0 != 0 && len1 != 0 && len0 + len1 <= HF_LUT_BITS;
        // escape: the second-level table's number, lengths zero
        const uint32_t v = len0 == 0 ? sym0 << 8
                           : two     ? (len0 + len1) | (sym0 << 8) | ((e1[j] >> 8) << 16) | ((len0 | 0x80u) << 24)
                                     : len0 | (sym0 << 8) | (len0 << 24);
        hits += two;
        T.lut2[tid + 256 * j] = v;
    }
    hits = hf_wave_sum_u32(hits);
    if (lane == 0) atomicAdd(&T.hits, hits);
    __syncthreads();
    for (int i = tid; i < HF_LUT_SIZE * 2 / 16; i += 256) {
        reinterpret_cast<uint4*>(cb->lut)[i] = reinterpret_cast<const uint4*>(T.lut)[i];
        reinterpret_cast<uint4*>(cb->sub)[i] = reinterpret_cast<const uint4*>(T.sub)[i];
    }
    for (int i = tid; i < HF_LUT_SIZE * 4 / 16; i += 256)
        reinterpret_cast<uint4*>(cb->lut2)[i] = reinterpret_cast<const uint4*>(T.lut2)[i];
    if (tid == 0) {
        cb->n_sub = n_sub;
        cb->max_len = T.max_len;
        cb->pair_hits = T.hits;
    }
}

// Tables for a codebook whose encoder table is already in place (hf_codebook_upload).
__global__ __launch_bounds__(256) void hf_k_tables(hf_cb_dev* cb) {
    __shared__ TabLds T;
    T.enc[threadIdx.x] = cb->enc[threadIdx.x];
    __syncthreads();
    build_tables(T, cb, (int)threadIdx.x);
}

// ---- front half of EntropyAwareHuffmanSystem, device to device --------------------------------
// stat counts (by raw symbol) -> sort -> entropy decision -> filter -> tree -> adjust(16) ->
// canonical codes over ranks -> scatter through rank2sym (the RTL indexes the rank-ordered
// table by raw symbol, top_encoder.scala:106 vs :155-157) -> encoder table -> decode tables.
// One workgroup of four wavefronts: the first builds the codebook (a chain of wave-wide steps)
// while the others wait at the barrier, then all four build the decode tables from the encoder
// table left in LDS -- one launch, and the tables never make a trip through global memory.
//
// FROM_DATA: the counts are not given but taken from one block of at most HF_BLOCK_MAX_SYMS symbols
// first (SymbolStat in front of the rest, as in top_encoder.scala:29-41): counted in LDS by all
// four wavefronts and left in `freq_out`, from where the codebook is built as from given counts.
template <bool FROM_DATA>
__global__ __launch_bounds__(256) void hf_k_system(const u64* freq_in, const int mode_req,
                                                   const u64 thr, hf_cb_dev* cb, const void* data,
                                                   const uint32_t n, const int layout,
                                                   u64* freq_out) {
    __shared__ CbLds L;
    __shared__ u64 sorted[256];
    __shared__ TabLds T;
    const int N = 256;
    const u64* freq256 = freq_in;
    if (FROM_DATA) {
        __shared__ uint32_t block_counts[256];
        block_counts[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t* words = reinterpret_cast<const uint32_t*>(data);  // (16-byte aligned)
        if (layout == HF_LAYOUT_BF16_EXP) {  // symbol = bits 7..14 of a 16-bit element
            for (uint32_t i = threadIdx.x; i < n / 2; i += 256) {
                const uint32_t w = words[i];
                atomicAdd(&block_counts[(w >> 7) & 0xFFu], 1u);
                atomicAdd(&block_counts[(w >> 23) & 0xFFu], 1u);
            }
            if ((n & 1u) && threadIdx.x == 0) {
                const uint32_t e = reinterpret_cast<const uint16_t*>(data)[n - 1];
                atomicAdd(&block_counts[(e >> 7) & 0xFFu], 1u);
            }
        } else {
            for (uint32_t i = threadIdx.x; i < n / 4; i += 256) {
                const uint32_t w = words[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(&block_counts[(w >> (8 * k)) & 0xFFu], 1u);
            }
            if (threadIdx.x < (n & 3u))
                atomicAdd(&block_counts[reinterpret_cast<const uint8_t*>(data)[(n & ~3u) + threadIdx.x]],
                          1u);
        }
        __syncthreads();
        freq_out[threadIdx.x] = block_counts[threadIdx.x];
        __syncthreads();  // (the workgroup's global stores are visible to its later loads)
        freq256 = freq_out;
    }
    if (threadIdx.x < 64) {  // (wave-uniform)
        const int lane = hf_lane();
        stage_sort(freq256, N, sorted, L.rank2sym, lane);
        const uint32_t ent = stage_entropy(freq256, N, lane);
        const uint32_t emode = entropy_mode(ent);
        // HF_MODE_BEST builds the Huffman tables first and keeps them only if they pay
        uint32_t mode = mode_req == HF_MODE_BEST   ? (uint32_t)HF_MODE_HUFFMAN
                        : mode_req == HF_MODE_AUTO ? emode
                                                   : (uint32_t)(mode_req & 1);
        wave_sync();
        u64 bits = 0, cnt = 0;
        if (mode == HF_MODE_HUFFMAN) {
            uint32_t h = system_tree_levels_runs(L, sorted, thr, HF_MAX_CODE_LEN, lane);
            adjust_levels(h, HF_MAX_CODE_LEN, lane);
            {
                const uint32_t pidx[4] = {(uint32_t)lane, (uint32_t)lane + 64u,
                                          (uint32_t)lane + 128u, (uint32_t)lane + 192u};
                uint32_t dep[4], code[4];
                assign_codes(h, HF_MAX_CODE_LEN, lane, pidx, dep, code);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    L.adj[pidx[i]] = (uint16_t)dep[i];
                    L.code[pidx[i]] = (uint16_t)code[i];
                }
            }
            wave_sync();
            for (int p = lane; p < N; p += 64) {
                const u64 f = sorted[p];
                bits += f * (u64)L.adj[p];
                cnt += f;
            }
            bits = hf_wave_sum_u64(bits);
            cnt = hf_wave_sum_u64(cnt);
            if (mode_req == HF_MODE_BEST && bits >= 8ull * cnt) mode = HF_MODE_BYPASS;
        }
        if (mode == HF_MODE_BYPASS) {
            // bypass table, top_encoder.scala:131-137: code(i) = i, length = symWidth
            for (int s = lane; s < 256; s += 64) {
                T.enc[s] = (uint32_t)s | (8u << 16);
                cb->enc[s] = (uint32_t)s | (8u << 16);
                cb->code[s] = (uint16_t)s;
                cb->len[s] = 8;
            }
            bits = 8, cnt = 1;
        } else {
            for (int p = lane; p < N; p += 64) {
                const int s = L.rank2sym[p];
                const uint32_t e = (uint32_t)L.code[p] | ((uint32_t)L.adj[p] << 16);
                T.enc[s] = e;
                cb->enc[s] = e;
                cb->code[s] = L.code[p];
                cb->len[s] = (uint8_t)L.adj[p];
            }
        }
        if (lane == 0) {
            cb->mode = mode;
            cb->entropy_q16 = ent;
            cb->entropy_mode = emode;
            // expected payload of a full chunk: 16384 symbols x mean code length, in 32-bit
            // words, plus the two look-ahead words each of the 64 lanes may leave behind
            cb->chunk_words_hint =
                cnt ? (uint32_t)(bits * (HF_CHUNK_SYMS / 32) / cnt) + 128u : 0u;
        }
    }
    __syncthreads();
    build_tables(T, cb, (int)threadIdx.x);
}

// ---- per-block codebooks ----------------------------------------------------------------------
// A batch of per-block codebooks goes through three kernels, each parallel the way its phase is:
//   front  one wavefront per codebook: SymbolSort, ShannonEntropy, mode;
//   merge  one LANE per codebook: the two-queue merge of HuffmanTreeBuilder, which is serial in
//          nature -- run by one lane of a wavefront it would cost the whole wavefront's issue
//          slots per codebook, run by 64 lanes on 64 codebooks it costs a 64th of that;
//   back   one wavefront per codebook: leaves per level, tree_adjust, canonical codes, the record.
// Between the kernels a codebook lives in a scratch block in global memory (counts in rank
// order, rank -> symbol, parent links of the internal nodes, a few words of state).
constexpr uint32_t kChunkCbTodo = 0xFFFFFFFFu;  // mode word of a record left to the general kernel

struct ChunkScratch {
    uint8_t* par;     // [batch][256] parent of internal node N + i, as its index minus N
    uint16_t* f;      // [batch][256] counts in rank order
    uint8_t* r2s;     // [batch][256] symbol of rank p
    uint32_t* meta;   // [batch][4]   entropy, mode, total count, 1 if the merge has to run
};

ChunkScratch carve_scratch(void* base, const uint64_t batch) {
    uint8_t* p = reinterpret_cast<uint8_t*>(base);
    ChunkScratch S;
    S.par = p;
    p += batch * 256;
    S.f = reinterpret_cast<uint16_t*>(p);
    p += batch * 512;
    S.r2s = p;
    p += batch * 256;
    S.meta = reinterpret_cast<uint32_t*>(p);
    return S;
}

// front: counts of one block -> counts and symbols in rank order, entropy, mode.  32-bit keys
// (counts of a real block sum to 16384; sums of 2^16 and more are left to the general kernel).
__global__ __launch_bounds__(64) void hf_k_chunks_front(const uint32_t* __restrict__ freq_chunks,
                                                        const uint32_t nbatch, const int mode_req,
                                                        const ChunkScratch S) {
    const int lane = hf_lane();
    const uint32_t c = blockIdx.x;
    if (c >= nbatch) return;
    uint32_t cnt[4];
    {
        const uint4 v = reinterpret_cast<const uint4*>(freq_chunks + (size_t)c * 256)[lane];
        cnt[0] = v.x, cnt[1] = v.y, cnt[2] = v.z, cnt[3] = v.w;
    }
    u64 tot64 = (u64)cnt[0] + cnt[1] + cnt[2] + cnt[3];
    tot64 = hf_wave_sum_u64(tot64);
    if (tot64 >= (1ull << 16)) {
        if (lane == 0) {
            S.meta[c * 4 + 1] = kChunkCbTodo;
            S.meta[c * 4 + 3] = 0;
        }
        return;
    }
    uint32_t k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) k[i] = (cnt[i] << 8) | (uint32_t)(255 - (lane * 4 + i));
    sort256_desc_u32(k, lane);  // (count desc, symbol asc); element e = lane * 4 + i
    uint32_t f[4];
    uint32_t r2s4 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[i] = k[i] >> 8;
        r2s4 |= (255u - (k[i] & 0xFFu)) << (8 * i);
    }
    reinterpret_cast<uint32_t*>(S.r2s + (size_t)c * 256)[lane] = r2s4;
    reinterpret_cast<uint2*>(S.f + (size_t)c * 256)[lane] =
        make_uint2(f[0] | (f[1] << 16), f[2] | (f[3] << 16));
    // ShannonEntropy (shannon_entropy.scala:67-101); the sum is order-free
    const uint32_t tot = (uint32_t)tot64;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (f[i] != 0 && tot != 0) {
            const uint32_t prob32 = (f[i] << 16) / tot;                  // :72 (f < 2^16)
            const u64 prob = prob32;
            const uint32_t kk = prob32 ? (uint32_t)__clz(prob32) : 31u;  // :52-54
            const u64 pow2 = 1ull << ((kk + 9u) & 31u);                  // :78
            const u64 xk = (prob * (u64)kk) >> 16;                       // :81
            const u64 approx = (xk + (prob << 1) - pow2) & ((1ull << 49) - 1ull);  // :84-87
            acc += (uint32_t)((prob * approx) >> 16);                    // :90-91
        }
    }
    acc = hf_wave_sum_u32(acc);
    const uint32_t ent = 0x10000u - acc;
    // HF_MODE_BEST builds the Huffman table first and keeps it only if it pays
    const uint32_t mode = mode_req == HF_MODE_BEST   ? (uint32_t)HF_MODE_HUFFMAN
                          : mode_req == HF_MODE_AUTO ? entropy_mode(ent)
                                                     : (uint32_t)(mode_req & 1);
    if (lane == 0) {
        S.meta[c * 4 + 0] = ent;
        S.meta[c * 4 + 1] = mode;
        S.meta[c * 4 + 2] = tot;
        S.meta[c * 4 + 3] = mode == HF_MODE_HUFFMAN;
    }
}

// merge: lane = codebook.  HuffmanTreeBuilder as a two-queue merge (see stage_tree; the leaf queue
// is preferred on ties).  Only weights matter here: what the next stage needs of the tree is the
// number of leaves per level, and that follows from the parent links of the internal nodes alone
// (see the back kernel).  The leaf queue is therefore just the counts in rank order read backwards
// (sFilter applied: entropy_aware_tree_builder.scala:61-66), whatever order ties would take.
// Both queues live in LDS, one column per lane: the 64 rank-ordered count arrays are loaded with
// coalesced reads and transposed on the way in (rows padded to 66 entries so that the transposing
// writes spread over 8 banks), the queue of internal nodes is a FIFO written at node kn, the same
// row in every lane, and read at each lane's own head; it reuses the rows of the leaves already
// taken, so a wavefront needs 33 KiB and four of them fit a CU.  A node that leaves the FIFO
// leaves its cell to the index of its parent, and the parent links go out coalesced at the end.  The next two candidates of either queue
// sit in registers, fetched one pick before they can become a head.
constexpr int kLqStride = 66;
__global__ __launch_bounds__(64) void hf_k_chunks_merge(const uint32_t nbatch, const u64 thr,
                                                        const ChunkScratch S) {
    constexpr int N = 256;
    // Row q holds leaf weight q of the ascending queue (rank N - 1 - q) until the lane has taken
    // it, afterwards the weight of internal node N + q: node N + q is made when at least q leaves
    // are gone (every merge takes two nodes and at most q internal ones exist to take), and the
    // leaves q and q + 1 already sit in registers by then.
    __shared__ uint16_t lqw[N][kLqStride];
    const int lane = hf_lane();
    const uint32_t c0 = blockIdx.x * 64u;
    {
        const uint32_t nhere = nbatch - c0 < 64u ? nbatch - c0 : 64u;
        for (uint32_t j0 = 0; j0 < nhere; j0 += 8) {
            uint2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                v[k] = j0 + k < nhere
                           ? reinterpret_cast<const uint2*>(S.f + (size_t)(c0 + j0 + k) * 256)[lane]
                           : make_uint2(0u, 0u);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t w4[4] = {v[k].x & 0xFFFFu, v[k].x >> 16, v[k].y & 0xFFFFu, v[k].y >> 16};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    lqw[N - 1 - (lane * 4 + i)][j0 + k] = (u64)w4[i] < thr ? (uint16_t)0 : (uint16_t)w4[i];
            }
        }
    }
    __syncthreads();
    const uint32_t c = c0 + lane;
    const bool run = c < nbatch && S.meta[c * 4 + 3] != 0;
    int li = 0, ii = N;
    uint32_t l0 = lqw[0][lane], l1 = lqw[1][lane];  // the next two leaf weights
    uint32_t iw0 = 0, iw1 = 0;                      // w[ii], w[ii + 1] while those nodes exist
    for (int kn = N; run && kn < 2 * N - 1; ++kn) {
        uint32_t wt[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const bool leaf = (li < N) && (ii >= kn || l0 <= iw0);
            if (leaf) {
                wt[p] = l0;
                ++li;
                l0 = l1;
                l1 = li + 1 < N ? (uint32_t)lqw[li + 1][lane] : 0u;
            } else {
                // the node's weight is in a register by now: its cell takes the parent's index
                lqw[ii - N][lane] = (uint16_t)(kn - N);
                wt[p] = iw0;
                ++ii;
                iw0 = iw1;
                iw1 = ii + 1 < kn ? (uint32_t)lqw[ii + 1 - N][lane] : 0u;
            }
        }
        const uint32_t ws = wt[0] + wt[1];
        lqw[kn - N][lane] = (uint16_t)ws;
        if (ii == kn) iw0 = ws;
        else if (ii + 1 == kn) iw1 = ws;
    }
    __syncthreads();
    // parent links out, transposed back: lane l writes the parents of nodes 4l .. 4l+3 of one
    // codebook after the other (row N - 2 is the root's weight, row N - 1 unused: never read)
    const uint32_t nhere = nbatch - c0 < 64u ? nbatch - c0 : 64u;
    for (uint32_t j = 0; j < nhere; ++j) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) v |= ((uint32_t)lqw[lane * 4 + i][j] & 0xFFu) << (8 * i);
        reinterpret_cast<uint32_t*>(S.par + (size_t)(c0 + j) * 256)[lane] = v;
    }
}

// back: parent links -> leaves per level -> tree_adjust -> canonical codes -> the record.
//
// The internal nodes leave the merge's FIFO in the order they were made, so the parent index never
// decreases with the node index, and the internal nodes of one level are a contiguous index range
// directly below the range of the level above (the root alone on level 0).  Level d + 1 is the set
// of nodes whose parent lies in level d -- one ballot per four nodes -- and a level of I[d]
// internal nodes under I[d-1] parents leaves 2 I[d-1] - I[d] places to leaves.  tree_adjust needs
// no more than these counts (clamped to the length limit), so no leaf depth is ever computed.
__global__ __launch_bounds__(64) void hf_k_chunks_back(const uint32_t nbatch, const int mode_req,
                                                       const ChunkScratch S,
                                                       hf_chunk_cb* __restrict__ cbs) {
    constexpr int N = 256;
    constexpr int lmax = HF_MAX_CODE_LEN;
    const int lane = hf_lane();
    const uint32_t c = blockIdx.x;
    if (c >= nbatch) return;
    hf_chunk_cb* cb = cbs + c;
    uint32_t mode = S.meta[c * 4 + 1];
    if (mode == kChunkCbTodo) {
        if (lane == 0) cb->mode = kChunkCbTodo;
        return;
    }
    const uint32_t ent = S.meta[c * 4 + 0];
    const uint32_t p[4] = {(uint32_t)lane * 4u, (uint32_t)lane * 4u + 1u, (uint32_t)lane * 4u + 2u,
                           (uint32_t)lane * 4u + 3u};
    uint32_t dep[4] = {8, 8, 8, 8}, code[4] = {0, 0, 0, 0};
    if (mode == HF_MODE_HUFFMAN) {
        // lane l holds the parents of internal nodes 4l .. 4l+3 (node N-2 is the root, N-1 no node)
        const uint32_t par4 = reinterpret_cast<const uint32_t*>(S.par + (size_t)c * 256)[lane];
        uint32_t lo = N - 2, hi = N - 2, up = 1;  // the level above: index range and size
        uint32_t h = 0;                           // leaves on level `lane`
        uint32_t placed = 0;                      // leaves on the levels seen so far
        for (uint32_t d = 1;; ++d) {
            if (d == (uint32_t)lmax) {  // everything deeper is clamped to lmax anyway
                if (lane == lmax) h += (uint32_t)N - placed;
                break;
            }
            uint32_t cnt = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t pk = (par4 >> (8 * i)) & 0xFFu;
                cnt += (uint32_t)__popcll(__ballot(p[i] < (uint32_t)(N - 2) && pk >= lo && pk <= hi));
            }
            const uint32_t lv = 2u * up - cnt;
            if ((uint32_t)lane == d) h += lv;
            placed += lv;
            if (cnt == 0) break;
            hi = lo - 1u;
            lo -= cnt;
            up = cnt;
        }
        adjust_levels(h, lmax, lane);
        assign_codes(h, lmax, lane, p, dep, code);
        if (mode_req == HF_MODE_BEST) {
            // keep the Huffman table only if it saves bits on this block's own counts
            const uint2 fv = reinterpret_cast<const uint2*>(S.f + (size_t)c * 256)[lane];
            uint32_t bits = (fv.x & 0xFFFFu) * dep[0] + (fv.x >> 16) * dep[1] +
                            (fv.y & 0xFFFFu) * dep[2] + (fv.y >> 16) * dep[3];
            bits = hf_wave_sum_u32(bits);
            if (bits >= 8u * S.meta[c * 4 + 2]) mode = HF_MODE_BYPASS;
        }
    }
    if (mode != HF_MODE_HUFFMAN) {
        // bypass table, top_encoder.scala:131-137: code(i) = i, length = symWidth
        for (int s = lane; s < 256; s += 64) {
            cb->enc[s] = (uint32_t)s | (8u << 16);
            cb->rank2sym[s] = (uint8_t)s;
            cb->len_by_rank[s] = 8;
        }
    } else {
        const uint32_t r2s4 = reinterpret_cast<const uint32_t*>(S.r2s + (size_t)c * 256)[lane];
        uint32_t len4 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t sym = (r2s4 >> (8 * i)) & 0xFFu, len = dep[i];
            cb->enc[sym] = code[i] | (len << 16);
            len4 |= len << (8 * i);
        }
        reinterpret_cast<uint32_t*>(cb->rank2sym)[lane] = r2s4;
        reinterpret_cast<uint32_t*>(cb->len_by_rank)[lane] = len4;
    }
    if (lane == 0) {
        cb->mode = mode;
        cb->entropy_q16 = ent;
    }
}

// General stages (64-bit weights) for the records the front kernel marked.  A small grid of
// wavefronts scans the marks 64 records at a time.
__global__ __launch_bounds__(64) void hf_k_system_chunks_rest(
    const uint32_t* __restrict__ freq_chunks, const uint32_t nchunks, const int mode_req,
    const u64 thr, hf_chunk_cb* __restrict__ cbs) {
    __shared__ CbLds L;
    __shared__ u64 sorted[256];
    __shared__ u64 fq[256];
    const int lane = hf_lane();
    const int N = 256;
    for (uint32_t base = blockIdx.x * 64u; base < nchunks; base += gridDim.x * 64u) {
        const uint32_t mine = base + lane;
        u64 todo = __ballot(mine < nchunks && cbs[mine].mode == kChunkCbTodo);
        while (todo) {
            const uint32_t c = base + (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            hf_chunk_cb* cb = cbs + c;
            __syncthreads();  // the previous record's LDS reads are done
            for (int e = lane; e < N; e += 64) fq[e] = freq_chunks[(size_t)c * 256 + e];
            __syncthreads();
            stage_sort(fq, N, sorted, L.rank2sym, lane);
            const uint32_t ent = stage_entropy(fq, N, lane);
            const uint32_t emode = entropy_mode(ent);
            uint32_t mode = mode_req == HF_MODE_BEST   ? (uint32_t)HF_MODE_HUFFMAN
                            : mode_req == HF_MODE_AUTO ? emode
                                                       : (uint32_t)(mode_req & 1);
            __syncthreads();
            if (mode == HF_MODE_HUFFMAN) {
                for (int e = lane; e < N; e += 64) {
                    const u64 f = sorted[e];
                    L.w[e] = f < thr ? 0ull : f;
                }
                __syncthreads();
                stage_tree(L, N, lane);
                stage_adjust(L, N, HF_MAX_CODE_LEN, lane);
                if (mode_req == HF_MODE_BEST) {
                    u64 bits = 0, cnt = 0;
                    for (int p = lane; p < N; p += 64) {
                        bits += sorted[p] * (u64)L.adj[p];
                        cnt += sorted[p];
                    }
                    bits = hf_wave_sum_u64(bits);
                    cnt = hf_wave_sum_u64(cnt);
                    if (bits >= 8ull * cnt) mode = HF_MODE_BYPASS;
                }
            }
            if (mode == HF_MODE_BYPASS) {
                for (int s = lane; s < 256; s += 64) {
                    cb->enc[s] = (uint32_t)s | (8u << 16);
                    cb->rank2sym[s] = (uint8_t)s;
                    cb->len_by_rank[s] = 8;
                }
            } else {
                for (int p = lane; p < N; p += 64) {
                    const int s = L.rank2sym[p];
                    cb->enc[s] = (uint32_t)L.code[p] | ((uint32_t)L.adj[p] << 16);
                    cb->rank2sym[p] = (uint8_t)s;
                    cb->len_by_rank[p] = (uint8_t)L.adj[p];
                }
            }
            if (lane == 0) {
                cb->mode = mode;
                cb->entropy_q16 = ent;
            }
        }
    }
}

// encoder lookup mux, one beat per symbol (top_encoder.scala:148-162)
template <int LAYOUT>
__global__ __launch_bounds__(256) void hf_k_pairs(const void* __restrict__ data, const uint64_t n,
                                                  const hf_cb_dev* __restrict__ cb,
                                                  hf_code_beat* __restrict__ out) {
    __shared__ uint32_t enc[256];
    enc[threadIdx.x] = cb->enc[threadIdx.x];
    const uint32_t mode = cb->mode;
    __syncthreads();
    // a beat is {code u16, length u8, mode u8} = the encoder table entry with the mode on top;
    // a lane takes 4 consecutive symbols and writes their 4 beats as one 16-byte store, so both
    // the loads and the stores of a wavefront are contiguous; two such groups per iteration
    static_assert(sizeof(hf_code_beat) == 4, "beat layout");
    const uint32_t mtop = mode << 24;
    const uint64_t ngroups = n / 8 * 2;  // groups of 4 symbols (the last n % 8 symbols go below)
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint32_t* o32 = reinterpret_cast<uint32_t*>(out);
    auto load4 = [&](const uint64_t g, uint32_t (&sy)[4]) {
        if (LAYOUT == HF_LAYOUT_BF16_EXP) {
            const hf_u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const hf_u32x2*>(data) + g);
            sy[0] = (v.x >> 7) & 0xFFu, sy[1] = (v.x >> 23) & 0xFFu;
            sy[2] = (v.y >> 7) & 0xFFu, sy[3] = (v.y >> 23) & 0xFFu;
        } else {
            const uint32_t v = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(data) + g);
            sy[0] = v & 0xFFu, sy[1] = (v >> 8) & 0xFFu, sy[2] = (v >> 16) & 0xFFu, sy[3] = v >> 24;
        }
    };
    auto store4 = [&](const uint64_t g, const uint32_t (&sy)[4]) {
        __builtin_nontemporal_store(hf_u32x4{enc[sy[0]] | mtop, enc[sy[1]] | mtop, enc[sy[2]] | mtop,
                                             enc[sy[3]] | mtop},
                                    reinterpret_cast<hf_u32x4*>(o32) + g);
    };
    uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (; g + stride < ngroups; g += 2 * stride) {
        uint32_t a[4], b[4];
        load4(g, a);
        load4(g + stride, b);
        store4(g, a);
        store4(g + stride, b);
    }
    if (g < ngroups) {
        uint32_t a[4];
        load4(g, a);
        store4(g, a);
    }
    // the last n % 8 symbols
    if (blockIdx.x == 0 && threadIdx.x < (uint32_t)(n - ngroups * 4)) {
        const uint64_t i = ngroups * 4 + threadIdx.x;
        uint32_t sym;
        if (LAYOUT == HF_LAYOUT_BF16_EXP)
            sym = (reinterpret_cast<const uint16_t*>(data)[i] >> 7) & 0xFFu;
        else
            sym = reinterpret_cast<const uint8_t*>(data)[i];
        o32[i] = enc[sym] | mtop;
    }
}

}  // namespace

void hf_launch_stage(hf_stage_io* io, int N, int op, int lmax, unsigned long long thr,
                     hipStream_t s) {
    hipLaunchKernelGGL(hf_k_stage, dim3(1), dim3(64), 0, s, io, N, op, lmax, thr);
}

void hf_launch_system(const unsigned long long* d_freq256, int mode, unsigned long long thr,
                      hf_cb_dev* cb, hipStream_t s) {
    hipLaunchKernelGGL(hf_k_system<false>, dim3(1), dim3(256), 0, s, d_freq256, mode, thr, cb,
                       (const void*)nullptr, 0u, 0, (unsigned long long*)nullptr);
}

void hf_launch_system_block(const void* d_data, uint32_t n, int layout, int mode,
                            unsigned long long thr, unsigned long long* d_freq256, hf_cb_dev* cb,
                            hipStream_t s) {
    hipLaunchKernelGGL(hf_k_system<true>, dim3(1), dim3(256), 0, s,
                       (const unsigned long long*)nullptr, mode, thr, cb, d_data, n, layout,
                       d_freq256);
}

static_assert(HF_CB_SCRATCH_PER_CHUNK == 256 + 512 + 256 + 16, "scratch layout (carve_scratch)");
size_t hf_chunk_cb_scratch_bytes(void) { return (size_t)HF_CB_BATCH * HF_CB_SCRATCH_PER_CHUNK; }

void hf_launch_system_chunks(const uint32_t* d_freq_chunks, uint64_t nchunks, int mode,
                             unsigned long long thr, hf_chunk_cb* cbs, void* d_scratch,
                             uint32_t batch, hipStream_t s) {
    if (nchunks == 0 || batch == 0) return;
    const ChunkScratch S = carve_scratch(d_scratch, batch);
    // the scratch block holds `batch` codebooks; larger jobs go through it batch by batch
    for (uint64_t c0 = 0; c0 < nchunks; c0 += batch) {
        const uint32_t nb = (uint32_t)(nchunks - c0 < batch ? nchunks - c0 : batch);
        hipLaunchKernelGGL(hf_k_chunks_front, dim3(nb), dim3(64), 0, s, d_freq_chunks + c0 * 256, nb,
                           mode, S);
        hipLaunchKernelGGL(hf_k_chunks_merge, dim3((nb + 63) / 64), dim3(64), 0, s, nb, thr, S);
        hipLaunchKernelGGL(hf_k_chunks_back, dim3(nb), dim3(64), 0, s, nb, mode, S, cbs + c0);
    }
    uint64_t rest = (nchunks + 63) / 64;
    if (rest > 1024) rest = 1024;
    hipLaunchKernelGGL(hf_k_system_chunks_rest, dim3((unsigned)rest), dim3(64), 0, s, d_freq_chunks,
                       (uint32_t)nchunks, mode, thr, cbs);
}

void hf_launch_tables(hf_cb_dev* cb, hipStream_t s) {
    hipLaunchKernelGGL(hf_k_tables, dim3(1), dim3(256), 0, s, cb);
}

void hf_launch_pairs(const void* d_data, uint64_t n, int layout, const hf_cb_dev* cb,
                     hf_code_beat* out, hipStream_t s) {
    if (n == 0) return;
    uint64_t grid = (n / 8 + 255) / 256;  // two 4-symbol groups per thread and pass
    if (grid > 8192) grid = 8192;
    if (grid == 0) grid = 1;
    if (layout == HF_LAYOUT_BF16_EXP)
        hipLaunchKernelGGL(hf_k_pairs<HF_LAYOUT_BF16_EXP>, dim3((unsigned)grid), dim3(256), 0, s,
                           d_data, n, cb, out);
    else
        hipLaunchKernelGGL(hf_k_pairs<HF_LAYOUT_U8>, dim3((unsigned)grid), dim3(256), 0, s, d_data,
                           n, cb, out);
}
