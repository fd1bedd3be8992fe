// Packed-stream decoder.  The reference only names this block (ParrallelHuffmanDecoder,
// decoder/top_decoder.scala:6-10, an empty IO bundle; reports/figs/decoder.svg sketches a
// scheduler feeding parallel sub-decoders and a combiner), so the design is this library's.
//
// One wavefront per chunk of 16384 symbols, 64 lanes decoding 64 interleaved bit strings in
// lock step (layout: hf_encode.hip).  Before each pair of symbols the lanes holding fewer
// than 32 buffered bits take the next payload words, lanes in ascending order, so the
// wavefront consumes the payload strictly front to back.  It keeps a 128-word window of the
// payload in two registers per lane (lane j holds words base+j and base+64+j) with the next
// 64 words already in flight, and lanes fetch their word with a wave shuffle: every global
// access is a contiguous 256-byte read.  A 12-bit first-level table in LDS resolves almost
// every symbol with one lookup; longer codes fall back to a scan of the few long entries.
//
// HBM traffic per bf16 element: ~L/8 B payload + 1 B side read, 2 B written.
#include "hf_common.h"

namespace {

typedef unsigned long long u64;
constexpr int kWaves = 4;

struct DecTables {
    uint16_t lut[HF_LUT_SIZE];
    uint32_t long_tab[256];  // code | len << 16 | sym << 24 for codes longer than HF_LUT_BITS
    uint32_t n_long;
};

__device__ __forceinline__ uint32_t decode_one(u64& win, int& nb, const DecTables& T,
                                               uint32_t* err) {
    const uint32_t top = (uint32_t)(win >> 32);
    const uint32_t e = T.lut[top >> (32 - HF_LUT_BITS)];
    uint32_t len = e & 0xFFu;
    uint32_t sym = e >> 8;
    if (len == 0) {
        const uint32_t v16 = top >> 16;
        const uint32_t nl = T.n_long;
        len = 1;  // keep moving on a stream no code matches
        sym = 0;
        bool found = false;
        for (uint32_t i = 0; i < nl; ++i) {
            const uint32_t t = T.long_tab[i];
            const uint32_t L = (t >> 16) & 0x1Fu;
            if ((v16 >> (16 - L)) == (t & 0xFFFFu)) {
                len = L;
                sym = t >> 24;
                found = true;
                break;
            }
        }
        if (!found) atomicOr(err, HF_KERR_CORRUPT);
    }
    win <<= len;
    nb -= (int)len;
    return sym;
}

template <int LAYOUT, bool FULL>
__device__ void decode_chunk(const uint32_t* __restrict__ P, const uint32_t ndw,
                             const uint8_t* __restrict__ side_chunk, void* __restrict__ out_chunk,
                             const uint32_t nvalid, const DecTables& T, uint32_t* err,
                             const int lane) {
    uint32_t cur = (uint32_t)lane < ndw ? P[lane] : 0u;
    uint32_t nxt = 64u + lane < ndw ? P[64 + lane] : 0u;
    uint32_t nn = 128u + lane < ndw ? P[128 + lane] : 0u;
    uint32_t wbase = 0;  // payload index of cur's lane 0
    uint32_t pos = 0;    // next unread payload word (wave-uniform)
    u64 win = 0;         // buffered bits, MSB aligned
    int nb = 0;

    const uint32_t rounds = FULL ? HF_ROUNDS : (nvalid + 511u) / 512u;
    uint2 sd = make_uint2(0, 0), sd_next = make_uint2(0, 0);
    if (LAYOUT == HF_LAYOUT_BF16_EXP) {
        if (FULL || ((uint32_t)lane * 8 + 8 <= nvalid))
            sd = reinterpret_cast<const uint2*>(side_chunk)[lane];
        else
            for (int i = 0; i < 8; ++i)
                if ((uint32_t)lane * 8 + i < nvalid) {
                    const uint32_t b = side_chunk[lane * 8 + i];
                    if (i < 4) sd.x |= b << (8 * i); else sd.y |= b << (8 * (i - 4));
                }
    }

    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t g = r * 64u + lane;
        if (LAYOUT == HF_LAYOUT_BF16_EXP && r + 1 < rounds) {
            const uint32_t g1 = g + 64u;
            sd_next = make_uint2(0, 0);
            if (FULL || (g1 * 8 + 8 <= nvalid))
                sd_next = reinterpret_cast<const uint2*>(side_chunk)[g1];
            else
                for (int i = 0; i < 8; ++i)
                    if (g1 * 8 + i < nvalid) {
                        const uint32_t b = side_chunk[(size_t)g1 * 8 + i];
                        if (i < 4) sd_next.x |= b << (8 * i); else sd_next.y |= b << (8 * (i - 4));
                    }
        }
        uint32_t s[8];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            bool active = true, valid_b = true;
            if (!FULL) {
                const uint32_t i0 = g * 8 + 2 * jp;
                active = i0 < nvalid;
                valid_b = i0 + 1 < nvalid;
            }
            const bool need = active && nb < 32;
            const u64 m = __ballot(need);
            if (m != 0) {  // wave-uniform
                const uint32_t idx = pos - wbase + hf_mbcnt(m);  // 0 .. 127
                const uint32_t v0 = __shfl(cur, (int)(idx & 63u));
                const uint32_t v1 = __shfl(nxt, (int)(idx & 63u));
                const uint32_t w = idx < 64u ? v0 : v1;
                if (need) {
                    win |= (u64)w << (32 - nb);
                    nb += 32;
                }
                pos += (uint32_t)__popcll(m);
                if (pos - wbase >= 64u) {
                    cur = nxt;
                    nxt = nn;
                    wbase += 64u;
                    const uint32_t k = wbase + 128u + lane;
                    nn = k < ndw ? P[k] : 0u;
                }
            }
            s[2 * jp] = active ? decode_one(win, nb, T, err) : 0u;
            s[2 * jp + 1] = valid_b ? decode_one(win, nb, T, err) : 0u;
        }
        if (LAYOUT == HF_LAYOUT_BF16_EXP) {
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t b0 = ((i < 2 ? sd.x : sd.y) >> (16 * (i & 1))) & 0xFFu;
                const uint32_t b1 = ((i < 2 ? sd.x : sd.y) >> (16 * (i & 1) + 8)) & 0xFFu;
                const uint32_t w0 = ((b0 & 0x80u) << 8) | (s[2 * i] << 7) | (b0 & 0x7Fu);
                const uint32_t w1 = ((b1 & 0x80u) << 8) | (s[2 * i + 1] << 7) | (b1 & 0x7Fu);
                o[i] = w0 | (w1 << 16);
            }
            if (FULL || (g * 8 + 8 <= nvalid)) {
                reinterpret_cast<uint4*>(out_chunk)[g] = make_uint4(o[0], o[1], o[2], o[3]);
            } else {
                uint16_t* q = reinterpret_cast<uint16_t*>(out_chunk) + (size_t)g * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (g * 8 + i < nvalid) q[i] = (uint16_t)(o[i >> 1] >> (16 * (i & 1)));
            }
            sd = sd_next;
        } else {
            const uint32_t o0 = s[0] | (s[1] << 8) | (s[2] << 16) | (s[3] << 24);
            const uint32_t o1 = s[4] | (s[5] << 8) | (s[6] << 16) | (s[7] << 24);
            if (FULL || (g * 8 + 8 <= nvalid)) {
                reinterpret_cast<uint2*>(out_chunk)[g] = make_uint2(o0, o1);
            } else {
                uint8_t* q = reinterpret_cast<uint8_t*>(out_chunk) + (size_t)g * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (g * 8 + i < nvalid) q[i] = (uint8_t)((i < 4 ? o0 : o1) >> (8 * (i & 3)));
            }
        }
    }
    // a well-formed chunk is consumed exactly
    if (pos != ndw && lane == 0) atomicOr(err, HF_KERR_CORRUPT);
}

template <int LAYOUT>
__global__ __launch_bounds__(kWaves * 64) void hf_k_decode(
    const uint32_t* __restrict__ payload, const uint64_t payload_words,
    const uint32_t* __restrict__ chunk_off, const uint8_t* __restrict__ side, const uint64_t n,
    const uint32_t nchunks,
    const hf_cb_dev* __restrict__ cb, void* __restrict__ out, uint32_t* err) {
    __shared__ DecTables T;
    {
        const uint4* src = reinterpret_cast<const uint4*>(cb->lut);
        uint4* dst = reinterpret_cast<uint4*>(T.lut);
        for (int i = threadIdx.x; i < (int)(HF_LUT_SIZE * 2 / 16); i += kWaves * 64) dst[i] = src[i];
        const uint32_t nl = cb->n_long;
        if (threadIdx.x == 0) T.n_long = nl;
        for (uint32_t i = threadIdx.x; i < nl; i += kWaves * 64) {
            const uint32_t s = cb->long_syms[i];
            T.long_tab[i] = cb->enc[s] | (s << 24);
        }
    }
    __syncthreads();
    const int lane = hf_lane();
    const uint32_t wave = blockIdx.x * kWaves + (threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * kWaves;
    const size_t elem_bytes = LAYOUT == HF_LAYOUT_BF16_EXP ? 2 : 1;
    for (uint32_t c = wave; c < nchunks; c += nwaves) {
        const uint32_t off0 = __builtin_amdgcn_readfirstlane(chunk_off[c]);
        const uint32_t off1 = __builtin_amdgcn_readfirstlane(chunk_off[c + 1]);
        // offsets that run backwards or past the payload buffer are never followed
        const bool sane = off1 >= off0 && (uint64_t)off1 <= payload_words;
        const uint32_t ndw = sane ? off1 - off0 : 0u;
        if (!sane && lane == 0) atomicOr(err, HF_KERR_CORRUPT);
        const uint64_t first = (uint64_t)c * HF_CHUNK_SYMS;
        const uint64_t left = n - first;
        const uint32_t nvalid = left < HF_CHUNK_SYMS ? (uint32_t)left : HF_CHUNK_SYMS;
        const uint8_t* side_chunk = side ? side + first : nullptr;
        void* out_chunk = reinterpret_cast<uint8_t*>(out) + first * elem_bytes;
        if (nvalid == HF_CHUNK_SYMS)
            decode_chunk<LAYOUT, true>(payload + off0, ndw, side_chunk, out_chunk, nvalid, T, err,
                                       lane);
        else
            decode_chunk<LAYOUT, false>(payload + off0, ndw, side_chunk, out_chunk, nvalid, T, err,
                                        lane);
    }
}

}  // namespace

void hf_launch_decode(const uint32_t* d_payload, uint64_t payload_words,
                      const uint32_t* d_chunk_off, const uint8_t* d_side, uint64_t n, int layout,
                      const hf_cb_dev* cb, void* d_out, uint32_t* d_err, int num_cu,
                      hipStream_t s) {
    const uint64_t nchunks = (n + HF_CHUNK_SYMS - 1) / HF_CHUNK_SYMS;
    if (nchunks == 0) return;
    uint64_t grid = (nchunks + kWaves - 1) / kWaves;
    const uint64_t cap = (uint64_t)num_cu * 8;
    if (grid > cap) grid = cap;
    if (layout == HF_LAYOUT_BF16_EXP)
        hipLaunchKernelGGL(hf_k_decode<HF_LAYOUT_BF16_EXP>, dim3((unsigned)grid),
                           dim3(kWaves * 64), 0, s, d_payload, payload_words, d_chunk_off,
                           d_side, n, (uint32_t)nchunks, cb, d_out, d_err);
    else
        hipLaunchKernelGGL(hf_k_decode<HF_LAYOUT_U8>, dim3((unsigned)grid), dim3(kWaves * 64), 0, s,
                           d_payload, payload_words, d_chunk_off, (const uint8_t*)nullptr, n,
                           (uint32_t)nchunks, cb, d_out, d_err);
}
