// Container (wire format v1, include/huffhip.h): conversion of per-chunk codebook records
// between their device form (hf_chunk_cb, 1552 bytes: encoder table + canonical order) and their
// stored form (296 bytes: canonical order only).  The reference has no stored form at all
// (top_encoder.scala:17-21 ends at the {code, length, mode} beat), so the format is this
// library's; everything else of a container is plain copies done by the host API.
#include "hf_common.h"

namespace {

// One wavefront per record.  Lane l owns ranks 4l .. 4l+3.
__global__ __launch_bounds__(64) void hf_k_cb_to_wire(const hf_chunk_cb* __restrict__ cbs,
                                                      const uint32_t nchunks,
                                                      hf_wire_chunk_cb* __restrict__ out) {
    const int lane = hf_lane();
    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const hf_chunk_cb* cb = cbs + c;
        hf_wire_chunk_cb* w = out + c;
        const uint32_t lens4 = reinterpret_cast<const uint32_t*>(cb->len_by_rank)[lane];
        reinterpret_cast<uint32_t*>(w->rank2sym)[lane] =
            reinterpret_cast<const uint32_t*>(cb->rank2sym)[lane];
        uint32_t mine = 0;
        for (uint32_t d = 1; d <= 16; ++d) {
            uint32_t cnt = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                cnt += (uint32_t)__popcll(__ballot(((lens4 >> (8 * i)) & 0xFFu) == d));
            if ((uint32_t)lane == d - 1) mine = cnt;
        }
        if (lane < 16) w->nlen[lane] = (uint16_t)mine;
        if (lane == 0) {
            w->mode = cb->mode;
            w->entropy_q16 = cb->entropy_q16;
        }
    }
}

// Stored record -> device record.  Lengths by rank follow from the counts per length (they never
// decrease with rank); the codes of ranks 0, 1, 2, ... tile the left-aligned 16-bit code space
// front to back, rank p covering 2^(16 - len_p) values, so a prefix sum of those widths gives
// every code -- the same canonical assignment the codebook kernel makes.  A record whose counts
// do not describe 256 ranks of a prefix code is reported and replaced by the bypass table.
__global__ __launch_bounds__(64) void hf_k_wire_to_cb(const hf_wire_chunk_cb* __restrict__ in,
                                                      const uint32_t nchunks,
                                                      hf_chunk_cb* __restrict__ cbs,
                                                      uint32_t* err) {
    const int lane = hf_lane();
    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const hf_wire_chunk_cb* w = in + c;
        hf_chunk_cb* cb = cbs + c;
        uint32_t cum[16];
        uint32_t total = 0, kraft = 0;  // kraft in units of 2^-16
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const uint32_t cnt = w->nlen[d];
            total += cnt;
            kraft += cnt << (15 - d);
            cum[d] = total;
        }
        const uint32_t mode = w->mode;
        const bool bad = total != 256u || kraft > 65536u || mode > 1u;
        const uint32_t syms4 = reinterpret_cast<const uint32_t*>(w->rank2sym)[lane];
        uint32_t len[4], start[4];
        uint32_t lane_total = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t p = (uint32_t)lane * 4u + i;
            uint32_t l = 1;
#pragma unroll
            for (int d = 0; d < 16; ++d) l += cum[d] <= p;
            len[i] = bad ? 8u : (l > 16u ? 16u : l);
            start[i] = lane_total;
            lane_total += 1u << (16 - len[i]);
        }
        uint32_t incl = lane_total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        const uint32_t lane_base = incl - lane_total;
        uint32_t r2s4 = 0, len4 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t p = (uint32_t)lane * 4u + i;
            const uint32_t sym = bad ? p : ((syms4 >> (8 * i)) & 0xFFu);
            const uint32_t code = ((lane_base + start[i]) >> (16 - len[i])) & 0xFFFFu;
            cb->enc[sym] = code | (len[i] << 16);
            r2s4 |= sym << (8 * i);
            len4 |= len[i] << (8 * i);
        }
        reinterpret_cast<uint32_t*>(cb->rank2sym)[lane] = r2s4;
        reinterpret_cast<uint32_t*>(cb->len_by_rank)[lane] = len4;
        if (lane == 0) {
            cb->mode = bad ? (uint32_t)HF_MODE_BYPASS : mode;
            cb->entropy_q16 = w->entropy_q16;
            cb->pad[0] = cb->pad[1] = 0;
            if (bad) atomicOr(err, HF_KERR_CORRUPT);
        }
    }
}

}  // namespace

void hf_launch_cb_to_wire(const hf_chunk_cb* cbs, uint64_t nchunks, hf_wire_chunk_cb* out,
                          hipStream_t s) {
    if (nchunks == 0) return;
    const uint64_t grid = nchunks < 65536 ? nchunks : 65536;
    hipLaunchKernelGGL(hf_k_cb_to_wire, dim3((unsigned)grid), dim3(64), 0, s, cbs, (uint32_t)nchunks,
                       out);
}

void hf_launch_wire_to_cb(const hf_wire_chunk_cb* in, uint64_t nchunks, hf_chunk_cb* cbs,
                          uint32_t* d_err, hipStream_t s) {
    if (nchunks == 0) return;
    const uint64_t grid = nchunks < 65536 ? nchunks : 65536;
    hipLaunchKernelGGL(hf_k_wire_to_cb, dim3((unsigned)grid), dim3(64), 0, s, in, (uint32_t)nchunks,
                       cbs, d_err);
}
