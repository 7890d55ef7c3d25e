// k_anchor — stage A1 of the `matchy match` hot path on gfx950: the streaming pass over the log.
//
// One wavefront owns one 16 KiB segment at a time (grid-stride). Per 1 KiB block:
//   * every lane loads 16 contiguous log bytes (one coalesced global_load_dwordx4 per lane = 1 KiB per wave),
//   * bytes become class bytes through a 256-entry LDS table and are staged in LDS (one ds_write_b128 per lane),
//   * 16 rows of 64 bytes: each lane reads the class bytes of positions j-4..j (two ds_read_b32 + v_alignbyte) and
//     tests the anchor patterns with and/compare pairs; the compare results ARE the wave's 64-bit lane masks, so
//     compaction needs only v_mbcnt on them (wavefront ballot + prefix-sum) and ring bookkeeping stays scalar,
//   * anchors go to per-type LDS rings (IPv4, domain, rare) and are flushed 64 at a time with ONE atomic and one
//     coalesced store per flush.
// Tokens long enough to be hashes / crypto addresses are rare: a block is checked for them with three SWAR
// operations per lane and only flagged blocks run the exact per-row token-length logic.
//
// Anchor rules (exact-coverage arguments in DESIGN.md §Anchors; differential-tested against oracle/):
//   IPv4    '.' at j preceded by 1-3 digits preceded by a boundary / buffer start  (ext:1120-1179, 813-869)
//   domain  byte that can start a PSL last label at j, '.' at j-1, label byte at j-2 (ext:537-628)
//   IPv6    "::" ending at j, no third ':' before it                                (ext:1044-1116)
//   e-mail  '@' at j                                                                 (ext:1182-1196)
//   token   boundary at j closing a token of length 26..62, 64, 90..110 or 128       (ext:1212-1409)
#include "device_common.h"

namespace mxy {

constexpr int AW = 4;                   // waves per workgroup
constexpr uint32_t BLK_BYTES = 1024;    // bytes per wave iteration
constexpr uint32_t CS_PREFIX = 16;      // class bytes kept in front of the block (the last 4 are used)
constexpr uint32_t QCAP = 128;          // ring entries per wave and type

__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Move `n` (<= 64) ring entries to the global list through the wave's chunk writer.
template <class T, uint32_t CHUNK>
__device__ __forceinline__ void flush_ring(const T* ring, uint32_t& head, uint32_t n, ChunkWriter<T, CHUNK>& cw, T* out, uint32_t cap,
                                           uint32_t* counter, const T& sentinel) {
    const uint32_t lane = lane_id();
    __builtin_amdgcn_wave_barrier();
    T v = sentinel;
    if (lane < n) v = ring[(head + lane) & (QCAP - 1)];
    cw.append(lane < n, v, out, cap, counter, sentinel);
    head += n;
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(AW * 64) void k_anchor(TokParams p, DevDb db) {
    __shared__ uint8_t ctab[256];
    __shared__ __attribute__((aligned(16))) uint32_t cstage[AW][(CS_PREFIX + BLK_BYTES) / 4];
    __shared__ uint32_t q_v4[AW][QCAP];
    __shared__ uint32_t q_dom[AW][QCAP];
    __shared__ uint2 q_misc[AW][QCAP];

    ctab[threadIdx.x] = (uint8_t)(class_of(threadIdx.x) | (((db.tld_first[threadIdx.x >> 5] >> (threadIdx.x & 31)) & 1) ? C_TLD1 : 0));
    __syncthreads();

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * AW + wave, nw = gridDim.x * AW;
    uint32_t* cs32 = cstage[wave];
    uint32_t* rv4 = q_v4[wave];
    uint32_t* rdom = q_dom[wave];
    uint2* rmisc = q_misc[wave];
    const uint32_t len = p.len;
    const bool en_v4 = (p.flags & EX_IPV4) != 0, en_dom = (p.flags & EX_DOMAINS) != 0;
    const bool en_v6 = (p.flags & EX_IPV6) != 0, en_at = (p.flags & EX_EMAILS) != 0;
    const bool en_tok = (p.flags & (EX_HASHES | EX_BITCOIN | EX_ETHEREUM | EX_MONERO)) != 0;
    const bool en_rare_row = en_v6 || en_at;

    constexpr uint32_t M_DOM = (C_TLD1 << 24) | (C_DOT << 16) | (C_LD << 8);
    constexpr uint32_t M_V6A = (C_COLON << 24) | (C_COLON << 16) | (C_COLON << 8), M_V6B = (C_COLON << 24) | (C_COLON << 16);
    constexpr uint32_t M_P1 = (C_DOT << 24) | (C_DIG << 16) | (C_B << 8);
    constexpr uint32_t M_P2 = (C_DOT << 24) | (C_DIG << 16) | (C_DIG << 8) | C_B;
    constexpr uint32_t M_P3 = (C_DOT << 24) | (C_DIG << 16) | (C_DIG << 8) | C_DIG;

    const uint32_t sh = lane & 3;
    const bool sh3 = sh == 3;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t nl_count = 0;                       // per-lane '\n' count, reduced once at the end
    uint32_t v4h = 0, v4t = 0, dh = 0, dt = 0, mh = 0, mt = 0;  // ring heads / tails (wave-uniform)
    ChunkWriter<uint32_t, ANCHOR_CHUNK> cw_v4, cw_dom;
    ChunkWriter<uint2, RARE_CHUNK> cw_misc;  // rare anchors are sparse: small chunks keep the list dense
    const uint32_t S32 = 0xFFFFFFFFu;
    const uint2 S64 = make_uint2(0xFFFFFFFFu, 0xFFu);
    uint2* rare_out = reinterpret_cast<uint2*>(p.rare);

    for (uint32_t seg = gw; seg < p.n_segs; seg += nw) {
        const uint32_t seg_start = seg * SEG_BYTES;
        // positions 0..len are scanned: position `len` (padding, class "boundary") closes a trailing token
        const uint32_t seg_end = min(seg_start + SEG_BYTES, len + 1);
        if (lane == 0) {
            uint32_t pre = C_B * 0x01010101u;  // before the buffer: boundary
            if (seg_start) {
                pre = 0;
                for (uint32_t k = 0; k < 4; ++k) pre |= (uint32_t)ctab[p.log[seg_start - 4 + k]] << (8 * k);
            }
            cs32[CS_PREFIX / 4 - 1] = pre;
        }
        bool prev_flagged = true;   // unknown for the block before the segment: take the exact path for the first block
        bool prev_slow = false;
        uint32_t prev_h1_last = 0;
        int32_t lastB = -1;

        for (uint32_t blk = seg_start; blk < seg_end; blk += BLK_BYTES) {
            // ---- stage 1 KiB: coalesced 16 B per lane, bytes -> class bytes
            const uint32_t pos0 = blk + lane * 16;
            uint32_t wv[4];
            if (pos0 + 16 <= len) {
                uint4 v = *reinterpret_cast<const uint4*>(p.log + pos0);
                wv[0] = v.x; wv[1] = v.y; wv[2] = v.z; wv[3] = v.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t x = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        uint32_t q = pos0 + k * 4 + b;
                        x |= (q < len ? (uint32_t)p.log[q] : (uint32_t)' ') << (8 * b);
                    }
                    wv[k] = x;
                }
            }
            uint32_t cv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t x = wv[k];
                cv[k] = (uint32_t)ctab[x & 0xFF] | ((uint32_t)ctab[(x >> 8) & 0xFF] << 8) | ((uint32_t)ctab[(x >> 16) & 0xFF] << 16) |
                        ((uint32_t)ctab[x >> 24] << 24);
                nl_count += __popc(cv[k] & (C_NL * 0x01010101u));
            }
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<uint4*>(&cs32[CS_PREFIX / 4 + lane * 4]) = make_uint4(cv[0], cv[1], cv[2], cv[3]);
            __builtin_amdgcn_wave_barrier();

            // ---- long-token precheck: a token of >= 26 bytes contains two consecutive boundary-free aligned 8-byte chunks
            bool slow = false;
            if (en_tok) {
                const uint64_t H0 = __ballot(((cv[0] | cv[1]) & (C_B * 0x01010101u)) == 0);
                const uint64_t H1 = __ballot(((cv[2] | cv[3]) & (C_B * 0x01010101u)) == 0);
                const bool flagged = ((H0 & H1) | (H1 & (H0 >> 1))) != 0 || (prev_h1_last && (H0 & 1));
                slow = flagged || prev_flagged;
                prev_flagged = flagged;
                prev_h1_last = (uint32_t)(H1 >> 63);
                if (slow && !prev_slow) {
                    // entering the exact path: find the last boundary within 256 bytes before this block
                    if (blk == 0) lastB = -1;
                    else {
                        lastB = (int32_t)blk - 257;  // "far": a token reaching back this far is longer than 128
                        const uint32_t back = min(blk, 256u);
                        for (uint32_t base = blk - back; base < blk; base += 64) {
                            uint32_t c = ctab[p.log[base + lane]];
                            uint64_t bm = __ballot(c & C_B);
                            if (bm) lastB = (int32_t)(base + 63 - __clzll((unsigned long long)bm));
                        }
                    }
                }
                prev_slow = slow;
            }

#pragma unroll 4
            for (uint32_t r = 0; r < 16; ++r) {
                const uint32_t row_base = blk + r * 64;
                const uint32_t j = row_base + lane;
                const uint32_t di = (CS_PREFIX + r * 64 + lane) >> 2;
                const uint32_t d1 = cs32[di], d0 = cs32[di - 1];
                // hist = classes of j-4..j-1 (byte 0 = j-4), y = classes of j-3..j (byte 3 = j)
                const uint32_t hist = __builtin_amdgcn_alignbyte(d1, d0, sh);
                const uint32_t ya = __builtin_amdgcn_alignbyte(d1, d0, sh + 1);
                const uint32_t y = sh3 ? d1 : ya;

                if (en_dom) {
                    const bool dom = (y & M_DOM) == M_DOM;
                    const uint64_t m = __ballot(dom);
                    if (m) {
                        if (dom) rdom[(dt + mbcnt64(m)) & (QCAP - 1)] = j;
                        dt += (uint32_t)__popcll(m);
                        if (dt - dh >= 64) flush_ring(rdom, dh, 64u, cw_dom, p.dom_list, p.dom_cap, &p.counters->n_dom, S32);
                    }
                }
                if (en_v4) {
                    const bool v4 = ((y & M_P1) == M_P1) || ((y & M_P2) == M_P2) || (((y & M_P3) == M_P3) && (hist & C_B));
                    const uint64_t m = __ballot(v4);
                    if (m) {
                        if (v4) rv4[(v4t + mbcnt64(m)) & (QCAP - 1)] = j;
                        v4t += (uint32_t)__popcll(m);
                        if (v4t - v4h >= 64) flush_ring(rv4, v4h, 64u, cw_v4, p.v4_list, p.v4_cap, &p.counters->n_v4, S32);
                    }
                }
                if (en_rare_row) {
                    const bool v6 = en_v6 && (y & M_V6A) == M_V6B;
                    const bool at = en_at && (y & (C_AT << 24)) != 0;
                    const uint64_t m = __ballot(v6 || at);
                    if (m) {
                        if (v6 || at) rmisc[(mt + mbcnt64(m)) & (QCAP - 1)] = make_uint2(j, v6 ? (uint32_t)RARE_V6 : (uint32_t)RARE_AT);
                        mt += (uint32_t)__popcll(m);
                        if (mt - mh >= 64) flush_ring(rmisc, mh, 64u, cw_misc, rare_out, p.rare_cap, &p.counters->n_rare, S64);
                    }
                }
                if (slow) {
                    // exact token ends: boundary at j, non-boundary at j-1, length from the last boundary before j
                    const bool b0 = (y & (C_B << 24)) != 0;
                    const uint64_t bmask = __ballot(b0);
                    const uint64_t mlt = bmask & lt_mask;
                    const int32_t lb = mlt ? (int32_t)(row_base + 63 - __clzll((unsigned long long)mlt)) : lastB;
                    const uint32_t tl = (uint32_t)((int32_t)j - 1 - lb);
                    const bool tok = b0 && !(y & (C_B << 16)) && ((tl >= 26 && tl <= 62) || tl == 64 || (tl >= 90 && tl <= 110) || tl == 128);
                    if (bmask) lastB = (int32_t)(row_base + 63 - __clzll((unsigned long long)bmask));
                    const uint64_t m = __ballot(tok);
                    if (m) {
                        if (tok) rmisc[(mt + mbcnt64(m)) & (QCAP - 1)] = make_uint2(j - tl, (uint32_t)RARE_TOK | (tl << 8));
                        mt += (uint32_t)__popcll(m);
                        if (mt - mh >= 64) flush_ring(rmisc, mh, 64u, cw_misc, rare_out, p.rare_cap, &p.counters->n_rare, S64);
                    }
                }
            }
            // keep the last 4 class bytes as the next block's prefix
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) cs32[CS_PREFIX / 4 - 1] = cs32[(CS_PREFIX + BLK_BYTES) / 4 - 1];
            __builtin_amdgcn_wave_barrier();
        }
    }
    // drain what is left in the rings
    if (dt != dh) flush_ring(rdom, dh, dt - dh, cw_dom, p.dom_list, p.dom_cap, &p.counters->n_dom, S32);
    if (v4t != v4h) flush_ring(rv4, v4h, v4t - v4h, cw_v4, p.v4_list, p.v4_cap, &p.counters->n_v4, S32);
    if (mt != mh) flush_ring(rmisc, mh, mt - mh, cw_misc, rare_out, p.rare_cap, &p.counters->n_rare, S64);
    cw_dom.pad_rest(p.dom_list, p.dom_cap, S32);
    cw_v4.pad_rest(p.v4_list, p.v4_cap, S32);
    cw_misc.pad_rest(rare_out, p.rare_cap, S64);
    // line count: wave reduction of the per-lane counts, one atomic per wave
    unsigned long long lines = nl_count;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lines += __shfl_down(lines, off);
    if (lane == 0 && lines) atomicAdd(&p.counters->lines, lines);
}

void launch_anchor(const TokParams& p, const DevDb& db, int grid, hipStream_t stream) {
    hipLaunchKernelGGL(k_anchor, dim3(grid), dim3(AW * 64), 0, stream, p, db);
}

}  // namespace mxy
