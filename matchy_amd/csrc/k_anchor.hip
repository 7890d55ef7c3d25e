// k_anchor — stage A1 of the `matchy match` hot path on gfx950: the streaming pass over the log.
//
// One wavefront owns one 16 KiB segment at a time (grid-stride). Per 1 KiB block:
//   * every lane loads 16 contiguous log bytes (one coalesced global_load_dwordx4 per lane = 1 KiB per wave),
//   * bytes become class bytes through a 256-entry LDS table and are staged in LDS (one ds_write_b128 per lane),
//   * 4 super-rows of 256 bytes: each lane owns one dword of class bytes (4 positions); with the previous dword and
//     three v_alignbyte it has the classes of positions j-4..j for its 4 positions and evaluates every anchor
//     pattern for all 4 at once (SWAR: shifts + ands, result in bit 0 of each byte); lanes with a hit are compacted
//     with ballot + v_mbcnt (wavefront ballot / prefix-sum) into per-type LDS rings,
//   * rings are moved to the global anchor lists 64 entries at a time through wave-private chunks
//     (one atomic per 1024 anchors, coalesced 256-byte stores).
// Tokens long enough to be hashes / crypto addresses (>= 26 bytes) are found without per-byte work: the ballot of
// "my dword has no boundary byte" gives one bit per dword, five set bits in a row below a token end are necessary
// for such a token, and only then is the exact length computed.
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

// (cur << k) with the top k bits of prev shifted in at the bottom: bit i = "dword i-k", across the super-row edge
__device__ __forceinline__ uint64_t shl_carry(uint64_t cur, uint64_t prev, int k) { return (cur << k) | (prev >> (64 - k)); }

__global__ __launch_bounds__(AW * 64) void k_anchor(TokParams p, DevDb db) {
    __shared__ uint8_t ctab[256];
    __shared__ __attribute__((aligned(16))) uint32_t cstage[AW][(CS_PREFIX + BLK_BYTES) / 4];
    __shared__ uint32_t q_v4[AW][QCAP];
    __shared__ uint32_t q_dom[AW][QCAP];
    __shared__ uint2 q_misc[AW][QCAP];
    __shared__ uint2 q_tok[AW][QCAP];

    ctab[threadIdx.x] = (uint8_t)(class_of(threadIdx.x) | (((db.tld_first[threadIdx.x >> 5] >> (threadIdx.x & 31)) & 1) ? C_TLD1 : 0));
    __syncthreads();

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * AW + wave, nw = gridDim.x * AW;
    uint32_t* cs32 = cstage[wave];
    uint32_t* rv4 = q_v4[wave];
    uint32_t* rdom = q_dom[wave];
    uint2* rmisc = q_misc[wave];
    uint2* rtok = q_tok[wave];
    const uint32_t len = p.len;
    const bool en_v4 = (p.flags & EX_IPV4) != 0, en_dom = (p.flags & EX_DOMAINS) != 0;
    const bool en_v6 = (p.flags & EX_IPV6) != 0, en_at = (p.flags & EX_EMAILS) != 0;
    const bool en_tok = (p.flags & (EX_HASHES | EX_BITCOIN | EX_ETHEREUM | EX_MONERO)) != 0;
    const bool en_rare_row = en_v6 || en_at;
    static_assert(C_B == 1 && C_DIG == 2 && C_DOT == 4 && C_COLON == 8 && C_AT == 16 && C_LD == 32 && C_TLD1 == 128, "SWAR shifts below");
    constexpr uint32_t LSB = 0x01010101u;

    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t nl_count = 0;                       // per-lane '\n' count, reduced once at the end
    uint32_t v4h = 0, v4t = 0, dh = 0, dt = 0, mh = 0, mt = 0, kh = 0, kt = 0;  // ring heads / tails (wave-uniform)
    ChunkWriter<uint32_t, ANCHOR_CHUNK> cw_v4, cw_dom;
    ChunkWriter<uint2, RARE_CHUNK> cw_misc, cw_tok;  // rare anchors are sparse: small chunks keep the lists dense
    const uint32_t S32 = 0xFFFFFFFFu;
    const uint2 S64 = make_uint2(0xFFFFFFFFu, 0xFFu);
    uint2* rare_out = reinterpret_cast<uint2*>(p.rare);
    uint2* tok_out = reinterpret_cast<uint2*>(p.tok);

    for (uint32_t seg = gw; seg < p.n_segs; seg += nw) {
        const uint32_t seg_start = seg * SEG_BYTES;
        // positions 0..len are scanned: position `len` (padding, class "boundary") closes a trailing token
        const uint32_t seg_end = min(seg_start + SEG_BYTES, len + 1);
        if (lane == 0) {
            uint32_t pre = C_B * LSB;  // before the buffer: boundary
            if (seg_start) {
                pre = 0;
                for (uint32_t k = 0; k < 4; ++k) pre |= (uint32_t)ctab[p.log[seg_start - 4 + k]] << (8 * k);
            }
            cs32[CS_PREFIX / 4 - 1] = pre;
        }
        // token state carried along the segment: position of the last boundary byte seen, and the boundary-free bits
        // of the previous super-row's dwords. At a segment start nothing is known about the dwords before it, so they
        // are taken as boundary-free (more exact checks, never fewer) and lastB comes from a 256-byte look-back.
        int32_t lastB = -1;
        uint64_t Zprev = ~0ull;
        if (en_tok && seg_start) {
            lastB = (int32_t)seg_start - 257;  // "far": a token reaching back this far is longer than 128 bytes
            for (uint32_t base = seg_start - 256; base < seg_start; base += 64) {
                const uint32_t c = ctab[p.log[base + lane]];
                const uint64_t bm = __ballot(c & C_B);
                if (bm) lastB = (int32_t)(base + 63 - __clzll((unsigned long long)bm));
            }
        }

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
                nl_count += __popc(cv[k] & (C_NL * LSB));
            }
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<uint4*>(&cs32[CS_PREFIX / 4 + lane * 4]) = make_uint4(cv[0], cv[1], cv[2], cv[3]);
            __builtin_amdgcn_wave_barrier();

            // ---- 4 super-rows of 256 bytes: every lane owns one dword of class bytes (4 positions) and evaluates the
            // anchor patterns for its 4 positions at once (SWAR on bit 0 of each byte)
#pragma unroll
            for (uint32_t sr = 0; sr < 4; ++sr) {
                const uint32_t d = CS_PREFIX / 4 + sr * 64 + lane;
                const uint32_t A = cs32[d], P = cs32[d - 1];
                const uint32_t A1 = __builtin_amdgcn_alignbyte(A, P, 3);  // classes of positions pos-1 .. pos+2
                const uint32_t A2 = __builtin_amdgcn_alignbyte(A, P, 2);  // pos-2 .. pos+1
                const uint32_t A3 = __builtin_amdgcn_alignbyte(A, P, 1);  // pos-3 .. pos
                const uint32_t sr_base = blk + sr * 256;
                const uint32_t pos = sr_base + lane * 4;
                if (en_dom) {
                    // TLD1 at j (bit 7), '.' at j-1 (bit 2), label byte at j-2 (bit 5)
                    uint32_t f = (A >> 7) & (A1 >> 2) & (A2 >> 5) & LSB;
                    for (;;) {
                        const uint64_t m = __ballot(f != 0);
                        if (!m) break;
                        if (f) {
                            rdom[(dt + mbcnt64(m)) & (QCAP - 1)] = pos + ((uint32_t)(__ffs((int)f) - 1) >> 3);
                            f &= f - 1;
                        }
                        dt += (uint32_t)__popcll(m);
                        if (dt - dh >= 64) flush_ring(rdom, dh, 64u, cw_dom, p.dom_list, p.dom_cap, &p.counters->n_dom, S32);
                    }
                }
                if (en_v4) {
                    // '.' at j (bit 2), digit at j-1 (bit 1), then boundary | digit,boundary | digit,digit,boundary
                    uint32_t f = (A >> 2) & (A1 >> 1) & (A2 | ((A2 >> 1) & (A3 | ((A3 >> 1) & P)))) & LSB;
                    for (;;) {
                        const uint64_t m = __ballot(f != 0);
                        if (!m) break;
                        if (f) {
                            rv4[(v4t + mbcnt64(m)) & (QCAP - 1)] = pos + ((uint32_t)(__ffs((int)f) - 1) >> 3);
                            f &= f - 1;
                        }
                        v4t += (uint32_t)__popcll(m);
                        if (v4t - v4h >= 64) flush_ring(rv4, v4h, 64u, cw_v4, p.v4_list, p.v4_cap, &p.counters->n_v4, S32);
                    }
                }
                if (en_rare_row) {
                    // "::" ending at j without a third ':' (bit 3 -> flag bit 0); '@' at j (bit 4 -> flag bit 1)
                    uint32_t f = 0;
                    if (en_v6) f |= (A >> 3) & (A1 >> 3) & ~(A2 >> 3) & LSB;
                    if (en_at) f |= (A >> 3) & (LSB << 1);
                    for (;;) {
                        const uint64_t m = __ballot(f != 0);
                        if (!m) break;
                        if (f) {
                            const uint32_t bit = (uint32_t)(__ffs((int)f) - 1);
                            rmisc[(mt + mbcnt64(m)) & (QCAP - 1)] = make_uint2(pos + (bit >> 3), (bit & 7) ? (uint32_t)RARE_AT : (uint32_t)RARE_V6);
                            f &= f - 1;
                        }
                        mt += (uint32_t)__popcll(m);
                        if (mt - mh >= 64) flush_ring(rmisc, mh, 64u, cw_misc, rare_out, p.rare_cap, &p.counters->n_rare, S64);
                    }
                }
                if (en_tok) {
                    // Z: one bit per dword of this super-row, set when the dword holds no boundary byte.
                    const uint32_t bl = A & LSB;
                    const uint64_t Z = __ballot(bl == 0);
                    // A token of >= 26 bytes that ends in dword i leaves dwords i-1..i-5 boundary-free (necessary).
                    const uint64_t C5 = shl_carry(Z, Zprev, 1) & shl_carry(Z, Zprev, 2) & shl_carry(Z, Zprev, 3) & shl_carry(Z, Zprev, 4) &
                                        shl_carry(Z, Zprev, 5);
                    // Only the lowest boundary byte of a dword can close a long token; it must follow a non-boundary byte.
                    const uint32_t low = bl & (0u - bl);
                    const bool cand = (low & ~A1) != 0 && ((C5 >> lane) & 1);
                    const uint64_t cm = __ballot(cand);
                    if (cm) {
                        // exact length: last boundary before my dword = highest boundary byte of the nearest lower dword
                        // that has one (this super-row), else the carried lastB
                        const uint64_t mlt = ~Z & lt_mask;
                        const uint32_t e = mlt ? (uint32_t)(63 - __clzll((unsigned long long)mlt)) : 0u;
                        const uint32_t Ae = (uint32_t)__shfl((int)bl, (int)e);
                        const int32_t lb = mlt ? (int32_t)(sr_base + e * 4 + ((31u - (uint32_t)__clz((int)Ae)) >> 3)) : lastB;
                        const uint32_t j = pos + ((uint32_t)(__ffs((int)low) - 1) >> 3);
                        const uint32_t tl = (uint32_t)((int32_t)j - 1 - lb);
                        const bool tok = cand && ((tl >= 26 && tl <= 62) || tl == 64 || (tl >= 90 && tl <= 110) || tl == 128);
                        const uint64_t m = __ballot(tok);
                        if (m) {
                            if (tok) rtok[(kt + mbcnt64(m)) & (QCAP - 1)] = make_uint2(j - tl, (uint32_t)RARE_TOK | (tl << 8));
                            kt += (uint32_t)__popcll(m);
                            if (kt - kh >= 64) flush_ring(rtok, kh, 64u, cw_tok, tok_out, p.tok_cap, &p.counters->n_tok, S64);
                        }
                    }
                    // carry: last boundary byte of this super-row, and its Z bits
                    const uint64_t nz = ~Z;
                    if (nz) {
                        const uint32_t e = (uint32_t)(63 - __clzll((unsigned long long)nz));
                        const uint32_t Ae = (uint32_t)__builtin_amdgcn_readlane((int)bl, (int)e);
                        lastB = (int32_t)(sr_base + e * 4 + ((31u - (uint32_t)__clz((int)Ae)) >> 3));
                    }
                    Zprev = Z;
                }
            }
            // keep the last 4 class bytes as the next block's prefix
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) cs32[CS_PREFIX / 4 - 1] = cs32[(CS_PREFIX + BLK_BYTES) / 4 - 1];
            __builtin_amdgcn_wave_barrier();
        }
    }
    // drain what is left in the rings, then mark the unused tail of every open chunk
    if (dt != dh) flush_ring(rdom, dh, dt - dh, cw_dom, p.dom_list, p.dom_cap, &p.counters->n_dom, S32);
    if (v4t != v4h) flush_ring(rv4, v4h, v4t - v4h, cw_v4, p.v4_list, p.v4_cap, &p.counters->n_v4, S32);
    if (mt != mh) flush_ring(rmisc, mh, mt - mh, cw_misc, rare_out, p.rare_cap, &p.counters->n_rare, S64);
    if (kt != kh) flush_ring(rtok, kh, kt - kh, cw_tok, tok_out, p.tok_cap, &p.counters->n_tok, S64);
    cw_dom.pad_rest(p.dom_list, p.dom_cap, S32);
    cw_v4.pad_rest(p.v4_list, p.v4_cap, S32);
    cw_misc.pad_rest(rare_out, p.rare_cap, S64);
    cw_tok.pad_rest(tok_out, p.tok_cap, S64);
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
