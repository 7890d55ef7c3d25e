// k_anchor — stage A1 of the `matchy match` hot path on gfx950: the streaming pass over the log.
//
// One wavefront owns one segment of the log at a time (grid-stride; 8-64 KiB, TokParams::seg_bytes). Per 1 KiB block:
//   * every lane loads 16 contiguous log bytes (one coalesced global_load_dwordx4 per lane = 1 KiB per wave; the next
//     block's load is issued before the current block is processed),
//   * the raw bytes go to a per-wave circular LDS window (8 or 4 KiB) so that anchors can be processed later without
//     touching HBM again; bytes become class bytes through a 256-entry LDS table,
//   * each lane keeps the class bytes of its 16 positions in 4 registers, gets the neighbouring dwords with two DPP wave
//     shifts and evaluates every anchor pattern for all 16 positions at once (SWAR: v_alignbyte + shifts + ands, result
//     in bit 0 of each byte); lanes with a hit are compacted with ballot + v_mbcnt (wavefront ballot / prefix-sum)
//     into per-type LDS rings, one compaction loop per block and type,
//   * when a ring holds 64 anchors the whole wave processes them, one anchor per lane, from the LDS window:
//     IPv4 anchors are fully validated (dotted-quad rules), checked against the database's /24 bitmap and leave as
//     candidates; domain anchors pass a SWAR prefilter (a later dot owns the run; the run ends at a boundary; the last
//     label is some public suffix's last label — Bloom filter) and the survivors leave with 32 bytes of context for
//     k_validate_dom. Anchors whose look-ahead lies in the block that is not staged yet go back into the ring.
//     Lists are written through wave-private chunks (one atomic per chunk, coalesced stores).
// Tokens long enough to be hashes / crypto addresses (>= 26 bytes) are found without per-byte work: four ballots of
// "my dword has no boundary byte" give one bit per dword, five set bits in a row below a token end are necessary
// for such a token (scalar shift / and on the masks), and only then is the exact length computed.
//
// Anchor rules (exact-coverage arguments in DESIGN.md §Anchors; differential-tested against oracle/):
//   IPv4    '.' at j preceded by 1-3 digits preceded by a boundary / buffer start, followed by 1-3 digits and '.'
//                                                                                    (ext:1120-1179, 813-869)
//   domain  byte that can start a PSL last label at j, '.' at j-1, label byte at j-2 (ext:537-628)
//   IPv6    "::" ending at j, no third ':' before it                                (ext:1044-1116)
//   e-mail  '@' at j                                                                 (ext:1182-1196)
//   token   boundary at j closing a token of length 26..62, 64, 90..110 or 128       (ext:1212-1409)
#include "device_common.h"

namespace mxy {

constexpr int AW = 4;                   // waves per workgroup
constexpr uint32_t BLK_BYTES = 1024;    // bytes per wave iteration
constexpr uint32_t QCAP = 128;          // ring entries per wave and type
// Raw-byte window per wave (circular, block granular): the kernel is built for two sizes. 8 KiB lets the anchor rings fill
// up before their oldest entry leaves the window (fuller drains: fewer instructions) but leaves room for 12 waves per CU;
// 4 KiB keeps 16 — better when most IPv4 candidates survive the /24 filter and the kernel writes a candidate per line
// (latency-sensitive). The host picks (TokParams::small_window).
constexpr uint32_t RAW_BYTES_MAX = 8192;
static_assert(SEG_ALIGN % RAW_BYTES_MAX == 0 && 4096 % BLK_BYTES == 0, "window wraps on block edges inside a segment");

__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// (cur << k) with the top k bits of prev shifted in at the bottom: bit i = "dword i-k", across the block edge
__device__ __forceinline__ uint64_t shl_carry(uint64_t cur, uint64_t prev, int k) { return (cur << k) | (prev >> (64 - k)); }

// n_dw + 1 dwords of the circular window starting at absolute byte position `a`, shifted so that byte `a` is byte 0
template <int N>
__device__ __forceinline__ void raw_read(const uint32_t* raw32, uint32_t raw_mask_dw, uint32_t a, uint32_t (&out)[N]) {
    const uint32_t i0 = a >> 2, sh = a & 3;
    uint32_t w[N + 1];
#pragma unroll
    for (int i = 0; i <= N; ++i) w[i] = raw32[(i0 + i) & raw_mask_dw];
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
}

struct WaveCtx {
    const TokParams* p;
    const uint32_t* raw32;
    uint32_t raw_mask_dw;      // window size in dwords - 1
    const uint32_t* bloom;
    const uint32_t* bm24;
    uint32_t res_lo, res_hi;   // absolute byte range currently held by the raw window
};

// IPv4 anchors are processed in two steps so that the one global load of the pipeline (the /24 occupancy bitmap of
// the database, DevDb::ip_bm24) has a whole ring period to arrive: drain_v4 validates up to 64 anchors, one per lane,
// from the LDS window and issues the bitmap load; commit_v4 (called by the next drain, or at the end) appends the
// candidates whose /24 holds database entries. Valid candidates that cannot hit are only counted.
struct PendingV4 {
    Candidate c{0, 0, 0, 0};
    uint32_t word = 0;
    bool ok = false;       // lane holds a validated candidate
    uint32_t n_valid = 0;  // per-lane count of validated candidates (reduced into ScanCounters::cand_true at the end)
};

__device__ __forceinline__ void commit_v4(PendingV4& pd, const WaveCtx& cx, BufferedWriter<Candidate>& cw) {
    const TokParams& p = *cx.p;
    const bool emit = pd.ok && ((pd.word >> ((pd.c.v4 >> 8) & 31)) & 1);
    cw.append(emit, pd.c, p.cands, p.cand_cap, &p.counters->n_cand);
    pd.ok = false;
}

__device__ __forceinline__ void drain_v4(uint32_t* ring, uint32_t& head, uint32_t& tail, uint32_t n, bool final, const WaveCtx& cx,
                                         PendingV4& pd, BufferedWriter<Candidate>& cw) {
    const uint32_t lane = lane_id();
    const TokParams& p = *cx.p;
    commit_v4(pd, cx, cw);
    __builtin_amdgcn_wave_barrier();
    const bool have = lane < n;
    uint32_t dot = 0;
    if (have) dot = ring[(head + lane) & (QCAP - 1)];
    head += n;
    // an anchor in the last bytes of the newest block has its look-ahead in the block that is not staged yet: it goes
    // back into the ring and is validated by a later drain (without this nearly every drain would drag one lane
    // through the global-memory fallback below)
    const bool later = have && !final && dot + 16 > cx.res_hi;
    const uint64_t rm = __ballot(later);
    if (rm) {
        __builtin_amdgcn_wave_barrier();
        if (later) ring[(tail + mbcnt64(rm)) & (QCAP - 1)] = dot;
        tail += (uint32_t)__popcll(rm);
    }
    if (have && !later) {
        uint32_t s = 0, e = 0, a = 0;
        bool ok;
        if (dot >= cx.res_lo + 4 && dot + 16 <= cx.res_hi) {
            uint32_t w[5];
            raw_read<5>(cx.raw32, cx.raw_mask_dw, dot - 4, w);
            ok = d_ipv4_from_window(make_uint4(w[0], w[1], w[2], w[3]), w[4], dot, s, e, a);
        } else if (dot >= 4 && dot + 16 <= p.len) {
            ok = val_ipv4_fast(p.log, dot, s, e, a);          // window not resident (segment edges): HBM
        } else {
            ok = val_ipv4(LogView{p.log, p.len}, dot, s, e, a);  // buffer edge
        }
        if (ok) {
            pd.c.start = s; pd.c.len_type = (e - s) | ((uint32_t)IT_IPV4 << 24); pd.c.v4 = a;
            pd.ok = true;
            pd.n_valid += 1;
            pd.word = p.filter_v4 ? cx.bm24[a >> 13] : 0xFFFFFFFFu;
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// Prefilter up to 64 domain anchors (first byte of the last label), one per lane; survivors go to the domain list.
// Mirrors the first steps of val_domain (validate_kernels.hip): a '.' inside the label means a later dot owns the run, the
// run must end at a boundary, and the label must be the last label of some public suffix. Undecidable cases (label
// longer than 8 bytes, bytes not resident) are kept.
__device__ __forceinline__ void drain_dom(uint32_t* ring, uint32_t& head, uint32_t& tail, uint32_t n, bool final, const WaveCtx& cx,
                                          DomWriter& dw) {
    const uint32_t lane = lane_id();
    const TokParams& p = *cx.p;
    __builtin_amdgcn_wave_barrier();
    uint32_t j = 0xFFFFFFFFu;
    bool keep = false, have_ctx = false;
    uint32_t ctx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ctx[k] = 0;
    const bool have = lane < n;
    if (have) j = ring[(head + lane) & (QCAP - 1)];
    head += n;
    // label window not staged yet (anchor in the last bytes of the newest block): back into the ring for a later drain
    const bool later = have && !final && j + 8 > cx.res_hi;
    const uint64_t rm = __ballot(later);
    if (rm) {
        __builtin_amdgcn_wave_barrier();
        if (later) ring[(tail + mbcnt64(rm)) & (QCAP - 1)] = j;
        tail += (uint32_t)__popcll(rm);
    }
    if (have && !later) {
        keep = true;
        if (j >= cx.res_lo + 24 && j + 8 <= cx.res_hi) {
            raw_read<8>(cx.raw32, cx.raw_mask_dw, j - 24, ctx);   // log[j-24, j+8): the last label starts at byte 24
            have_ctx = true;
            // the label's 8-byte window as SWAR masks (no per-byte loop, no divergent control flow)
            constexpr uint64_t H = 0x8080808080808080ull;
            const uint64_t w64 = (uint64_t)ctx[6] | ((uint64_t)ctx[7] << 32);
            const ByteMasks mw = domain_masks(w64);
            const uint64_t ndc = ~mw.dc & H;
            if (ndc) {   // the label ends inside the window: ll = its length (>= 1: byte j can start a TLD)
                const uint32_t ll = (uint32_t)(__ffsll((long long)ndc) - 1) >> 3;
                const uint64_t below = (1ull << (8 * ll)) - 1ull;
                const uint32_t stop = (uint32_t)(w64 >> (8 * ll)) & 0xFF;
                const uint32_t bit = tld_hash8((uint32_t)(w64 & below), (uint32_t)((w64 & below) >> 32)) & (TLD_BLOOM_BITS - 1);
                // a '.' inside the label: a later dot owns the run; the run must end at a boundary; the label must be
                // some public suffix's last label
                keep = (mw.dot & below) == 0 && d_is_boundary(stop) && ((cx.bloom[bit >> 5] >> (bit & 31)) & 1);
            } else if (mw.dot) {
                keep = false;         // 8 domain chars with a dot among them: a later dot owns the run
            } else if (j + 24 <= cx.res_hi) {
                // a label of 8+ bytes: usually not the last one ("www.examplesite.com" at 'e'). Look 16 bytes further: a
                // dot before the first non-domain byte settles it; otherwise it stays undecided (long last label)
                uint32_t more[4];
                raw_read<4>(cx.raw32, cx.raw_mask_dw, j + 8, more);
                const ByteMasks ma = domain_masks((uint64_t)more[0] | ((uint64_t)more[1] << 32));
                const ByteMasks mb = domain_masks((uint64_t)more[2] | ((uint64_t)more[3] << 32));
                const uint64_t na = ~ma.dc & H, nb = ~mb.dc & H;
                const uint64_t below_a = na ? ((1ull << ((uint32_t)(__ffsll((long long)na) - 1) & ~7u)) - 1ull) : ~0ull;
                const uint64_t below_b = na ? 0ull : (nb ? ((1ull << ((uint32_t)(__ffsll((long long)nb) - 1) & ~7u)) - 1ull) : ~0ull);
                keep = ((ma.dot & below_a) | (mb.dot & below_b)) == 0;
            }
        }
    }
    const uint32_t slot = dw.reserve(keep, p.dom_list, p.dom_cap, &p.counters->n_dom);
    if (slot != 0xFFFFFFFFu) {
        p.dom_list[dom_plane_index(slot, 0)] = have_ctx ? j : (j | 0x80000000u);
#pragma unroll
        for (int k = 0; k < 8; ++k) p.dom_list[dom_plane_index(slot, 1 + k)] = ctx[k];
    }
    __builtin_amdgcn_wave_barrier();
}

// value of lane-1 / lane+1 (DPP wave shift: no LDS traffic); lane 0 / lane 63 get `edge`
__device__ __forceinline__ uint32_t from_prev_lane(uint32_t v, uint32_t edge) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t from_next_lane(uint32_t v, uint32_t edge) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}
// index (0..3) of the highest non-zero byte of a word whose bytes are 0 or 1
__device__ __forceinline__ uint32_t top_byte(uint32_t x) { return (31u - (uint32_t)__clz((int)x)) >> 3; }

template <uint32_t RAW_BYTES>
__global__ __launch_bounds__(AW * 64) void k_anchor(TokParams p, DevDb db) {
    constexpr uint32_t RAW_DW = RAW_BYTES / 4;
    __shared__ uint8_t ctab[256];
    __shared__ uint32_t bloom[TLD_BLOOM_WORDS];
    __shared__ __attribute__((aligned(16))) uint32_t rawst[AW][RAW_DW];
    __shared__ uint32_t q_v4[AW][QCAP];
    __shared__ uint32_t q_dom[AW][QCAP];
    __shared__ uint2 wb_misc[AW][64], wb_tok[AW][64];   // BufferedWriter staging
    __shared__ Candidate wb_cand[AW][64];

    ctab[threadIdx.x] = (uint8_t)(class_of(threadIdx.x) | (((db.tld_first[threadIdx.x >> 5] >> (threadIdx.x & 31)) & 1) ? C_TLD1 : 0));
    for (uint32_t i = threadIdx.x; i < TLD_BLOOM_WORDS; i += AW * 64) bloom[i] = db.tld_bloom[i];
    __syncthreads();

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * AW + wave, nw = gridDim.x * AW;
    uint32_t* raw32 = rawst[wave];
    uint32_t* rv4 = q_v4[wave];
    uint32_t* rdom = q_dom[wave];
    const uint32_t len = p.len;
    const bool en_v4 = (p.flags & EX_IPV4) != 0, en_dom = (p.flags & EX_DOMAINS) != 0;
    const bool en_v6 = (p.flags & EX_IPV6) != 0, en_at = (p.flags & EX_EMAILS) != 0;
    const bool en_tok = (p.flags & (EX_HASHES | EX_BITCOIN | EX_ETHEREUM | EX_MONERO)) != 0;
    const bool en_rare_row = en_v6 || en_at;
    static_assert(C_B == 1 && C_DIG == 2 && C_DOT == 4 && C_COLON == 8 && C_AT == 16 && C_LD == 32 && C_TLD1 == 128, "SWAR shifts below");
    constexpr uint32_t LSB = 0x01010101u;

    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t nl_count = 0;                       // per-lane '\n' count, reduced once at the end
    uint32_t v4h = 0, v4t = 0, dh = 0, dt = 0;   // ring heads / tails (wave-uniform)
    uint32_t v4_old = 0, dom_old = 0;            // block start of the oldest ring entry (valid while the ring is non-empty)
    BufferedWriter<Candidate> cw_cand(wb_cand[wave]);   // IPv4 candidates that pass the /24 bitmap are sparse
    DomWriter cw_dom;
    BufferedWriter<uint2> cw_misc(wb_misc[wave]), cw_tok(wb_tok[wave]);   // rare anchors and long tokens are sparse: dense lists
    const uint2 S64 = make_uint2(0xFFFFFFFFu, 0xFFu);
    uint2* rare_out = reinterpret_cast<uint2*>(p.rare);
    uint2* tok_out = reinterpret_cast<uint2*>(p.tok);
    WaveCtx cx{&p, raw32, RAW_DW - 1, bloom, db.ip_bm24, 0u, 0u};
    PendingV4 pend;

    // 16 bytes of block `b` for this lane; positions >= len read as ' ' (a boundary, like the end of the buffer)
    auto load_block = [&](uint32_t b, uint32_t (&w)[4]) {
        const uint32_t pos0 = b + lane * 16;
        if (pos0 + 16 <= len) {
            // streamed once: a non-temporal load keeps the log from pushing the database's /24 bitmap (the one table
            // this kernel reads at random) out of L2
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p.log + pos0));
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t x = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const uint32_t q = pos0 + k * 4 + bb;
                    x |= (q < len ? (uint32_t)p.log[q] : (uint32_t)' ') << (8 * bb);
                }
                w[k] = x;
            }
        }
    };

    for (uint32_t seg = gw; seg < p.n_segs; seg += nw) {
        const uint32_t seg_start = seg * p.seg_bytes;
        // positions 0..len are scanned: position `len` (padding, class "boundary") closes a trailing token
        const uint32_t seg_end = min(seg_start + p.seg_bytes, len + 1);
        // classes of the 4 bytes in front of the segment (before the buffer: boundary), wave-uniform
        uint32_t carryP = C_B * LSB;
        if (seg_start) {
            carryP = 0;
            for (uint32_t k = 0; k < 4; ++k) carryP |= (uint32_t)ctab[p.log[seg_start - 4 + k]] << (8 * k);
            carryP = __builtin_amdgcn_readfirstlane(carryP);
        }
        // token state carried along the segment: position of the last boundary byte seen, and the boundary-free bits
        // of the previous block's dwords (Zp[k] bit L = dword k of lane L). At a segment start nothing is known about
        // the dwords before it, so they are taken as boundary-free (more exact checks, never fewer) and lastB comes
        // from a 256-byte look-back.
        int32_t lastB = -1;
        uint64_t Zp[4] = {~0ull, ~0ull, ~0ull, ~0ull};
        if (en_tok && seg_start) {
            lastB = (int32_t)seg_start - 257;  // "far": a token reaching back this far is longer than 128 bytes
            for (uint32_t base = seg_start - 256; base < seg_start; base += 64) {
                const uint32_t c = ctab[p.log[base + lane]];
                const uint64_t bm = __ballot(c & C_B);
                if (bm) lastB = (int32_t)(base + 63 - __clzll((unsigned long long)bm));
            }
        }

        uint32_t nx[4];
        load_block(seg_start, nx);
        for (uint32_t blk = seg_start; blk < seg_end; blk += BLK_BYTES) {
            // ---- anchors whose look-back bytes are about to be overwritten in the raw window leave first
            if (blk >= seg_start + RAW_BYTES - BLK_BYTES) {
                const uint32_t lim = blk - (RAW_BYTES - BLK_BYTES);   // entries of blocks <= lim expire
                if (v4t != v4h && v4_old <= lim) { drain_v4(rv4, v4h, v4t, v4t - v4h, false, cx, pend, cw_cand); v4_old = blk - BLK_BYTES; }
                if (dt != dh && dom_old <= lim) { drain_dom(rdom, dh, dt, dt - dh, false, cx, cw_dom); dom_old = blk - BLK_BYTES; }
            }
            // ---- this lane's 16 bytes: raw bytes into the window, bytes -> class bytes; prefetch the next block
            uint32_t wv[4] = {nx[0], nx[1], nx[2], nx[3]};
            if (blk + BLK_BYTES < seg_end) load_block(blk + BLK_BYTES, nx);
            // X[0] = classes of the 4 bytes before this lane's bytes, X[1..4] = its own 16, X[5] = the 4 after
            uint32_t X[6];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t x = wv[k];
                X[k + 1] = (uint32_t)ctab[x & 0xFF] | ((uint32_t)ctab[(x >> 8) & 0xFF] << 8) | ((uint32_t)ctab[(x >> 16) & 0xFF] << 16) |
                           ((uint32_t)ctab[x >> 24] << 24);
                nl_count += __popc(X[k + 1] & (C_NL * LSB));
            }
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<uint4*>(&raw32[((blk & (RAW_BYTES - 1)) >> 2) + lane * 4]) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
            __builtin_amdgcn_wave_barrier();
            cx.res_hi = blk + BLK_BYTES;
            cx.res_lo = cx.res_hi - seg_start > RAW_BYTES ? cx.res_hi - RAW_BYTES : seg_start;
            X[0] = from_prev_lane(X[4], carryP);
            // the bytes after lane 63's are in the next block: "anything" (all class bits) keeps the look-ahead test of
            // the IPv4 anchor conservative there
            X[5] = from_next_lane(X[1], 0xFFFFFFFFu);
            carryP = (uint32_t)__builtin_amdgcn_readlane((int)X[4], 63);
            const uint32_t pos0 = blk + lane * 16;

            // ---- every anchor pattern for the lane's 16 positions (SWAR on bit 0 of each byte, 4 dwords); the flags of
            // dword k go to bit k of each byte: bit (8 b + k) <-> position 4 k + b
            uint32_t Fd = 0, F4 = 0, Fm = 0, bl[4], A1s[4];
            // byte-shifted views shared by neighbouring dwords: S1[k] = classes of positions (4k-3 .. 4k) seen from dword
            // k-1/k, i.e. v_alignbyte(X[k+1], X[k], 1); the look-ahead views of dword k are the look-back views of k+1
            uint32_t S1[5], S2[5], S3[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                S1[k] = __builtin_amdgcn_alignbyte(X[k + 1], X[k], 1);
                S2[k] = __builtin_amdgcn_alignbyte(X[k + 1], X[k], 2);
                S3[k] = __builtin_amdgcn_alignbyte(X[k + 1], X[k], 3);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t A = X[k + 1], P = X[k], N = X[k + 2];
                const uint32_t A1 = S3[k];  // classes of positions pos-1 .. pos+2
                const uint32_t A2 = S2[k];  // pos-2 .. pos+1
                const uint32_t A3 = S1[k];  // pos-3 .. pos
                bl[k] = A & LSB;
                A1s[k] = A1;
                // domain: TLD1 at j (bit 7), '.' at j-1 (bit 2), label byte at j-2 (bit 5)
                if (en_dom) Fd |= ((A >> 7) & (A1 >> 2) & (A2 >> 5) & LSB) << k;
                if (en_v4) {
                    // '.' at j (bit 2), digit at j-1 (bit 1), then boundary | digit,boundary | digit,digit,boundary
                    uint32_t f = (A >> 2) & (A1 >> 1) & (A2 | ((A2 >> 1) & (A3 | ((A3 >> 1) & P)))) & LSB;
                    // ... and followed by a second octet and a second dot: digit, then '.' | digit,'.' | digit,digit,'.'
                    // (necessary for a dotted quad; drops "HTTP/1.1", "Mozilla/5.0", "Safari/537.36" style anchors)
                    const uint32_t F1 = S1[k + 1], F2 = S2[k + 1], F3 = S3[k + 1];  // pos+1 .., pos+2 .., pos+3 ..
                    f &= (F1 >> 1) & ((F2 >> 2) | ((F2 >> 1) & ((F3 >> 2) | ((F3 >> 1) & (N >> 2)))));
                    F4 |= f << k;
                }
                // "::" ending at j without a third ':' (bit 3) -> flag bits 0..3; '@' at j (bit 4) -> flag bits 4..7
                if (en_v6) Fm |= ((A >> 3) & (A1 >> 3) & ~(A2 >> 3) & LSB) << k;
                if (en_at) Fm |= ((A >> 4) & LSB) << (4 + k);
            }
            if (en_dom) {
                for (;;) {
                    const uint64_t m = __ballot(Fd != 0);
                    if (!m) break;
                    if (Fd) {
                        const uint32_t bit = (uint32_t)(__ffs((int)Fd) - 1);
                        rdom[(dt + mbcnt64(m)) & (QCAP - 1)] = pos0 + ((bit & 7) << 2) + (bit >> 3);
                        Fd &= Fd - 1;
                    }
                    if (dt == dh) dom_old = blk;
                    dt += (uint32_t)__popcll(m);
                    if (dt - dh >= 64) { drain_dom(rdom, dh, dt, 64u, false, cx, cw_dom); dom_old = blk; }
                }
            }
            if (en_v4) {
                for (;;) {
                    const uint64_t m = __ballot(F4 != 0);
                    if (!m) break;
                    if (F4) {
                        const uint32_t bit = (uint32_t)(__ffs((int)F4) - 1);
                        rv4[(v4t + mbcnt64(m)) & (QCAP - 1)] = pos0 + ((bit & 7) << 2) + (bit >> 3);
                        F4 &= F4 - 1;
                    }
                    if (v4t == v4h) v4_old = blk;
                    v4t += (uint32_t)__popcll(m);
                    if (v4t - v4h >= 64) { drain_v4(rv4, v4h, v4t, 64u, false, cx, pend, cw_cand); v4_old = blk; }
                }
            }
            if (en_rare_row) {
                for (;;) {
                    const uint64_t m = __ballot(Fm != 0);
                    if (!m) break;
                    uint2 v = S64;
                    const bool has = Fm != 0;
                    if (has) {
                        const uint32_t bit = (uint32_t)(__ffs((int)Fm) - 1);
                        v = make_uint2(pos0 + ((bit & 3) << 2) + (bit >> 3), (bit & 4) ? (uint32_t)RARE_AT : (uint32_t)RARE_V6);
                        Fm &= Fm - 1;
                    }
                    cw_misc.append(has, v, rare_out, p.rare_cap, &p.counters->n_rare);
                }
            }
            if (en_tok) {
                // Z[k]: bit L set when dword k of lane L holds no boundary byte (dword index in the block = 4 L + k)
                uint64_t Z[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) Z[k] = __ballot(bl[k] == 0);
                // A token of >= 26 bytes that ends in dword i leaves dwords i-1..i-5 boundary-free (necessary); only the
                // lowest boundary byte of a dword can close a long token and it must follow a non-boundary byte.
                uint32_t ck = 4;  // dword of this lane that may close a long token (at most one can)
                uint64_t c5[4];   // scalar masks: bit L = dwords i-1..i-5 of dword (L, k) are boundary-free
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint64_t c = ~0ull;
#pragma unroll
                    for (int t = 1; t <= 5; ++t) {
                        int idx = k - t, sh = 0;
                        while (idx < 0) { idx += 4; ++sh; }
                        c &= sh ? shl_carry(Z[idx], Zp[idx], sh) : Z[idx];
                    }
                    c5[k] = c;
                }
                // logs rarely hold 20+ boundary-free bytes: the per-lane tests run only when some dword qualifies
                if (c5[0] | c5[1] | c5[2] | c5[3]) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t low = bl[k] & (0u - bl[k]);
                        if ((low & ~A1s[k]) != 0 && ((c5[k] >> lane) & 1)) ck = k;
                    }
                }
                const uint64_t anyB = ~(Z[0] & Z[1] & Z[2] & Z[3]);  // lanes with at least one boundary byte
                const uint64_t cm = __ballot(ck < 4);
                if (cm) {
                    // exact length: the last boundary before the closing dword is the highest boundary byte of the
                    // nearest lower lane that has one (the 5 dwords in between are free), else the carried lastB
                    const uint32_t hb = bl[3] ? 12 + top_byte(bl[3]) : bl[2] ? 8 + top_byte(bl[2]) : bl[1] ? 4 + top_byte(bl[1]) : top_byte(bl[0] | 1u);
                    const uint64_t mlt = anyB & lt_mask;
                    const uint32_t e = mlt ? (uint32_t)(63 - __clzll((unsigned long long)mlt)) : 0u;
                    const uint32_t hbe = (uint32_t)__shfl((int)hb, (int)e);
                    const int32_t lb = mlt ? (int32_t)(blk + e * 16 + hbe) : lastB;
                    const uint32_t blk_k = ck == 0 ? bl[0] : ck == 1 ? bl[1] : ck == 2 ? bl[2] : bl[3];
                    const uint32_t low = blk_k & (0u - blk_k);
                    const uint32_t j = pos0 + (ck & 3) * 4 + (low ? ((uint32_t)(__ffs((int)low) - 1) >> 3) : 0u);
                    const uint32_t tl = (uint32_t)((int32_t)j - 1 - lb);
                    const bool tok = ck < 4 && ((tl >= 26 && tl <= 62) || tl == 64 || (tl >= 90 && tl <= 110) || tl == 128);
                    cw_tok.append(tok, make_uint2(j - tl, (uint32_t)RARE_TOK | (tl << 8)), tok_out, p.tok_cap, &p.counters->n_tok);
                }
                // carry: last boundary byte of this block (scalar: the highest lane with a boundary), and the Z bits
                if (anyB) {
                    const uint32_t e = (uint32_t)(63 - __clzll((unsigned long long)anyB));
                    const uint32_t s3 = (uint32_t)__builtin_amdgcn_readlane((int)bl[3], (int)e), s2 = (uint32_t)__builtin_amdgcn_readlane((int)bl[2], (int)e);
                    const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane((int)bl[1], (int)e), s0 = (uint32_t)__builtin_amdgcn_readlane((int)bl[0], (int)e);
                    const uint32_t hbs = s3 ? 12 + top_byte(s3) : s2 ? 8 + top_byte(s2) : s1 ? 4 + top_byte(s1) : top_byte(s0 | 1u);
                    lastB = (int32_t)(blk + e * 16 + hbs);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) Zp[k] = Z[k];
            }
        }
        // the next segment of this wave is not contiguous: finish the rings while their bytes are still in the window
        if (dt != dh) drain_dom(rdom, dh, dt, dt - dh, true, cx, cw_dom);
        if (v4t != v4h) drain_v4(rv4, v4h, v4t, v4t - v4h, true, cx, pend, cw_cand);
    }
    commit_v4(pend, cx, cw_cand);
    cw_misc.flush(rare_out, p.rare_cap, &p.counters->n_rare);
    cw_tok.flush(tok_out, p.tok_cap, &p.counters->n_tok);
    // mark the unused tail of every open chunk
    cw_dom.pad_rest(p.dom_list, p.dom_cap);
    cw_cand.flush(p.cands, p.cand_cap, &p.counters->n_cand);
    {
        uint32_t nv = pend.n_valid;  // validated IPv4 candidates, listed or not
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nv += __shfl_down(nv, off);
        if (lane == 0 && nv) atomicAdd(&p.counters->cand_true, nv);
    }
    // line count: wave reduction of the per-lane counts, one atomic per wave
    unsigned long long lines = nl_count;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lines += __shfl_down(lines, off);
    if (lane == 0 && lines) atomicAdd(&p.counters->lines, lines);
}

// workgroups of k_anchor that are resident on one CU at the same time (register / LDS limited)
int anchor_blocks_per_cu(bool small_window) {
    int n = 0;
    const hipError_t e = small_window ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_anchor<4096>, AW * 64, 0)
                                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_anchor<8192>, AW * 64, 0);
    if (e != hipSuccess || n < 1) n = small_window ? 4 : 3;
    return n;
}

void launch_anchor(const TokParams& p, const DevDb& db, int grid, hipStream_t stream) {
    if (p.small_window) hipLaunchKernelGGL(k_anchor<4096>, dim3(grid), dim3(AW * 64), 0, stream, p, db);
    else hipLaunchKernelGGL(k_anchor<8192>, dim3(grid), dim3(AW * 64), 0, stream, p, db);
}

}  // namespace mxy
