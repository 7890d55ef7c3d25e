// k_anchor — stage A1 of the `matchy match` hot path on gfx950: the streaming pass over the log.
//
// One wavefront owns one segment of the log at a time (grid-stride; 8-64 KiB, TokParams::seg_bytes) and walks it in blocks of
// 2 KiB = 8 rows of 256 bytes. Per block:
//   * every lane loads dword L of each row (8 coalesced global_load_dword, 256 contiguous bytes per wave instruction; the
//     next block's loads are issued before the current block is processed) and copies them to a per-wave circular LDS
//     window (8 KiB) in natural byte order, so that anchors can be processed later without touching HBM again;
//   * BIT-SLICED front end (anchor_planes.h): an 8 x 8 bit-matrix transpose turns the lane's 8 dwords into the 8 bit
//     planes of its 32 bytes (bit 8 b + q of plane c = bit c of byte b of row q), the byte classes (boundary, digit, '.',
//     ':', '@', '\n', label byte, possible first byte of a public-suffix label) are boolean functions of those planes
//     — one 32-bit operation classifies 32 log positions per lane, no table look-ups — and every anchor pattern is
//     AND / OR of class planes shifted by 1..4 positions. "k positions earlier" is v_alignbyte(P, P of the previous
//     dword, 4 - k): consecutive dwords of a row sit in consecutive lanes (one DPP wave shift per plane; lane 0 / lane 63
//     wrap to the neighbouring row of lane 63 / lane 0, fixed up with a readlane and two scalar operations);
//   * lanes with a hit are compacted with ballot + v_mbcnt (wavefront ballot / prefix-sum) into per-type LDS rings;
//   * when a ring holds 64 anchors the whole wave processes them, one anchor per lane, from the LDS window:
//     IPv4 anchors are fully validated (dotted-quad rules), checked against the database's /24 bitmap and leave as
//     candidates; domain anchors pass a SWAR prefilter (a later dot owns the run; the run ends at a boundary; the last
//     label is some public suffix's last label — Bloom filter) and the survivors leave with 32 bytes of context for
//     k_validate_dom. Anchors whose look-ahead lies in the block that is not staged yet go back into the ring.
//     Lists are written through wave-private chunks (one atomic per chunk, coalesced stores).
// Tokens long enough to be hashes / crypto addresses (>= 26 bytes) are found without per-byte work: every accepted form is
// made of ASCII letters and digits, so one bit per dword ("all four bytes alphanumeric") travels down the wave (DPP rotate);
// only when three such dwords stand in a row somewhere (rare in logs) is the chain taken to five, and a dword with a
// boundary byte behind five of them closes a candidate whose exact length is then computed from ballots.
//
// Anchor rules (exact-coverage arguments in DESIGN.md §Anchors; differential-tested against oracle/). The streaming
// pass may list MORE positions than these rules (the look-ahead of the last bytes of a block is taken as "anything", the
// first byte of a public-suffix label is a superset class): the drains and the validation kernels decide exactly.
//   IPv4    '.' at j preceded by 1-3 digits preceded by a boundary / buffer start, followed by 1-3 digits and '.'
//                                                                                    (ext:1120-1179, 813-869)
//   domain  byte that can start a PSL last label at j, '.' at j-1 (ext:537-628)
//   IPv6    "::" ending at j, no third ':' before it                                (ext:1044-1116)
//   e-mail  '@' at j                                                                 (ext:1182-1196)
//   token   boundary at j closing a token of length 26..62, 64, 90..110 or 128       (ext:1212-1409)
#include "anchor_planes.h"
#include "device_shared.h"

namespace mxy {

constexpr int AW = 4;                       // waves per workgroup
// Wave priorities (s_setprio). The four waves of a SIMD run the same program at different places, and at equal priority the oldest
// ready wave issues: a wave in the middle of a dependent chain — ring entry -> window bytes -> class table in a drain, the six DPP
// steps of the ring-position scan, the rotate chain of the long-token test — then loses issue slots to a wave that is in the 100-odd
// independent instructions of the bit-matrix transpose and the class functions, and its chain (which nothing else can overlap)
// stretches. So the transpose and the class functions run at priority 0, everything with cross-lane, LDS or memory latency in it at 1
// and the drains at 2: the bulk fills the gaps the chains leave instead of delaying them. Same instructions, 0.768 -> 0.677 ms on the
// headline batch (profiles/r04_k_anchor_priorities.txt).
constexpr int PRIO_BULK = 0, PRIO_CHAIN = 1, PRIO_DRAIN = 2;
constexpr uint32_t BLK_BYTES = AB_BLOCK;    // bytes per wave iteration (8 rows of 256)
constexpr uint32_t QCAP = 128;              // ring entries per wave and type
// Raw-byte window per wave (circular, block granular): four blocks, so that the anchor rings fill up before their oldest
// entry leaves the window (drains run with full lanes).
constexpr uint32_t RAW_BYTES = 8192;
static_assert(SEG_ALIGN % RAW_BYTES == 0 && RAW_BYTES % BLK_BYTES == 0 && RAW_BYTES >= 3 * BLK_BYTES, "window wraps on block edges inside a segment");

// N dwords of the circular window starting at absolute byte position `a`, shifted so that byte `a` is byte 0. The first
// RAW_MIRROR bytes of the window are kept a second time behind its end, so a read that starts near the end runs straight on
// (constant offsets: ds_read2_b32 pairs, no wrap arithmetic per dword).
constexpr uint32_t RAW_MIRROR = 64;
template <int N>
__device__ __forceinline__ void raw_read(const uint32_t* raw32, uint32_t a, uint32_t (&out)[N]) {
    static_assert(4 * (N + 1) <= RAW_MIRROR + 4, "read runs past the mirrored bytes");
    const uint32_t* q = raw32 + ((a & (RAW_BYTES - 1)) >> 2);
    const uint32_t sh = a & 3;
    uint32_t w[N + 1];
#pragma unroll
    for (int i = 0; i <= N; ++i) w[i] = q[i];
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
}

// Ring entry of an anchor found at bit t of lane L of the block that starts at blk (a multiple of 2048): blk | L << 5 | t.
// The compaction loops run once per anchor of the busiest lane, the drains once per 64 anchors, so the position is
// decoded in the drain.
__device__ __forceinline__ uint32_t anchor_pos(uint32_t e) {
    return (e & ~(BLK_BYTES - 1)) + ((e & 7u) << 8) + ((e >> 3) & 0xFCu) + ((e >> 3) & 3u);
}

// LDS budget: 4 workgroups (16 waves) per CU = 40 KiB per workgroup. Per wave: raw window 8 KiB + mirror, two anchor rings,
// and small staging buffers for the sparse output lists (a list that turns dense bypasses its buffer, BufferedWriter).
constexpr uint32_t CAND_STAGE = 16, RARE_STAGE = 8;
typedef StagedChunkWriter<Candidate, CAND_STAGE> CandWriter;
typedef BufferedWriter<uint2, RARE_STAGE> RareWriter;
static_assert(TLD_BLOOM_BITS / 2 == 1u << 14, "the prefilter takes bits 0..13 and 14..27 of the label hash");
constexpr uint32_t BLOOM_FOLD_WORDS = TLD_BLOOM_WORDS / 2;   // words of the prefilter's own Bloom filter (behind the general one in DevDb::tld_bloom)

struct WaveCtx {
    const TokParams* p;
    const uint32_t* raw32;
    const uint8_t* ctab;       // byte classes (LDS)
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

// TokParams::inline_v4: the wave looks its IPv4 candidates up itself. Candidates that pass the /24 bitmap collect as {start, address}
// in the LDS stage the candidate writer would use (V4_STAGE entries of 8 bytes in the same 256 bytes); when the stage is full —
// and at the end of the wave's work — one lane per entry walks the trie (trie_v4: first-levels table + at most 8 / 16 dependent
// node loads) and the hits leave as final records: one atomic on the record counter per flush, device array + pinned host mirror
// like pack_pending in lookup_kernels.hip. A few flushes per wave and batch; the other waves of the SIMD cover the load latency.
// Parameters of the COLD paths (list flushes, the inline lookups: a few times per thousand blocks) are read from the kernel-argument
// segment where they are needed instead of living in scalar registers through the block loop: the loop is short of them (the
// compiler was spilling two dozen to vector-register lanes per block — v_readlane / v_writelane on the vector pipe this kernel is
// bound by). The empty asm makes the pointer opaque, so the loads stay where the code is.
typedef const TokParams __attribute__((address_space(4)))* ColdTok;
typedef const DevDb __attribute__((address_space(4)))* ColdDb;
__device__ __forceinline__ ColdTok cold_tok() {
    ColdTok q = (ColdTok)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}
__device__ __forceinline__ ColdDb cold_db() {   // k_anchor(TokParams, DevDb): the second argument follows the first
    static_assert(sizeof(TokParams) % 8 == 0 && alignof(DevDb) == 8, "kernel-argument layout");
    ColdTok q = cold_tok();
    return (ColdDb)((const char __attribute__((address_space(4)))*)q + sizeof(TokParams));
}

// Writer for k_anchor's two sparse lists (WHICH: 0 = rare anchors: IPv6 / e-mail, 1 = long tokens), with the destination (list,
// capacity, counter) fetched from the kernel arguments when a flush happens instead of being passed — and kept in scalar registers — at
// every append. Entries collect in a small LDS stage of the wave and leave together.
// SPARSE OR DENSE. In a web-server log a wave meets a handful of these anchors and every flush reserves exactly its entries with one
// atomic: a list without padding. In an endpoint log with two file hashes per line, or an application log with a trace id per line, the
// same scheme is one returning atomic per seven entries from each of 4096 waves on ONE counter — atomics on one line are served one
// after the other, ~12 ns each: 2.2 M of them were 27 of the kernel's 28 ms on the hash-dense shape (profiles/r04_log_shapes.txt). So the
// reservation grows with the wave's own history: the first two flushes reserve what they carry, the next six a chunk of 64 slots, every
// later one a chunk of 512 that the following flushes fill without an atomic; what a wave leaves unused of its last chunk is marked with
// the sentinel every consumer of these lists skips (kind 0xFF). The chunk state lives in the stage's last slot (LDS, read at a flush),
// not in scalar registers.
constexpr uint32_t SPARSE_STAGE = RARE_STAGE - 1;   // entries per flush; slot RARE_STAGE - 1 holds {chunk base, used | capacity << 12 | flushes << 24}
template <int WHICH>
struct SparseWriter {
    uint2* buf;          // RARE_STAGE slots of LDS owned by this wave
    uint32_t cnt = 0;    // wave-uniform
    __device__ __forceinline__ explicit SparseWriter(uint2* lds) : buf(lds) {
        if (lane_id() == 0) buf[SPARSE_STAGE] = make_uint2(0xFFFFFFFFu, 0u);
    }
    struct Dest { uint2* out; uint32_t cap; uint32_t* counter; };
    __device__ __forceinline__ static Dest dest() {
        const ColdTok kp = cold_tok();
        if (WHICH == 0) return Dest{reinterpret_cast<uint2*>(kp->rare), kp->rare_cap, &kp->counters->n_rare};
        return Dest{reinterpret_cast<uint2*>(kp->tok), kp->tok_cap, &kp->counters->n_tok};
    }
    // slots for n entries (n <= 64): from the wave's current chunk, or from a new reservation; returns the first slot
    __device__ __forceinline__ uint32_t reserve(const Dest& d, uint32_t n) {
        __builtin_amdgcn_wave_barrier();
        const uint2 st = buf[SPARSE_STAGE];
        uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.x);
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.y);
        uint32_t used = w & 0xFFFu, cap = (w >> 12) & 0xFFFu, flushes = w >> 24;
        if (base == 0xFFFFFFFFu || used + n > cap) {
            // the rest of the old chunk stays unused: sentinels
            if (base != 0xFFFFFFFFu) for (uint32_t k = used + lane_id(); k < cap; k += 64) if (base + k < d.cap) d.out[base + k] = make_uint2(0xFFFFFFFFu, 0xFFu);
            // Dense lists (round 5): the steps above are a BURST on a log that has these anchors in every line — every wave starts with nine small
            // reservations in its first blocks, all 4096 waves at once: 37 K returning atomics on one line, each wave waiting its turn nine times
            // (hash-dense: 0.49 of k_anchor's 1.24 ms, measured by handing out the slots without the atomic). The host knows the density from the
            // previous batch (TokParams::tok_chunk / rare_chunk = a quarter of a wave's share of that list): two to five reservations per wave,
            // spread over its life.
            const uint32_t hinted = WHICH ? cold_tok()->tok_chunk : cold_tok()->rare_chunk;
            const uint32_t chunk = hinted ? hinted : (flushes < 4 ? n : (flushes < 32 ? 64u : 512u));
            cap = chunk < n ? n : chunk;
            uint32_t b = 0;
            if (lane_id() == 0) b = atomicAdd(d.counter, cap);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
            used = 0;
        }
        if (flushes < 255) ++flushes;
        __builtin_amdgcn_wave_barrier();
        if (lane_id() == 0) buf[SPARSE_STAGE] = make_uint2(base, (used + n) | (cap << 12) | (flushes << 24));
        __builtin_amdgcn_wave_barrier();
        return base + used;
    }
    __device__ __forceinline__ void flush() {
        if (cnt == 0) return;
        const Dest d = dest();
        const uint32_t b = reserve(d, cnt);
        if (lane_id() < cnt && b + lane_id() < d.cap) d.out[b + lane_id()] = buf[lane_id()];
        __builtin_amdgcn_wave_barrier();
        cnt = 0;
    }
    // end of the wave's work: staged entries out, the unused rest of the last chunk marked
    __device__ __forceinline__ void finish() {
        flush();
        __builtin_amdgcn_wave_barrier();
        const uint2 st = buf[SPARSE_STAGE];
        const uint32_t base = st.x, used = st.y & 0xFFFu, cap = (st.y >> 12) & 0xFFFu;
        if (base == 0xFFFFFFFFu || used >= cap) return;
        const Dest d = dest();
        for (uint32_t k = used + lane_id(); k < cap; k += 64) if (base + k < d.cap) d.out[base + k] = make_uint2(0xFFFFFFFFu, 0xFFu);
    }
    // all lanes of the (converged) wave call this
    __device__ __forceinline__ void append(bool emit, const uint2& v) {
        const uint64_t m = __ballot(emit);
        if (m == 0) return;
        const uint32_t n = (uint32_t)__popcll(m);
        const uint32_t rank = mbcnt64(m);
        if (n > SPARSE_STAGE) {   // more entries than the stage holds (dense phases): they leave directly, through the same chunks
            const Dest d = dest();
            const uint32_t b = reserve(d, n);
            if (emit && b + rank < d.cap) d.out[b + rank] = v;
            return;
        }
        if (cnt + n > SPARSE_STAGE) flush();
        if (emit) buf[cnt + rank] = v;
        cnt += n;
    }
};

// One entry per lane held in REGISTERS in front of a SparseWriter (the long tokens): the LDS stage has seven slots — this kernel's 40 KiB per
// workgroup are spoken for — and an endpoint or application log carries eight long tokens per 2 KiB block, so the stage was flushed in nearly
// every block: kernel-argument loads, the chunk state through LDS, a store, three wave barriers, with the wave waiting at each step (hash-dense:
// 392 M vector instructions but 1.24 ms, where 328 M take 0.65). A lane now keeps its token until it finds another one; then every lane's
// entry leaves in ONE reservation (the direct path of SparseWriter::append). A web-server log flushes once, at the end of the wave's work.
template <int WHICH>
struct LaneHeldWriter {
    SparseWriter<WHICH> sw;
    uint2 held = make_uint2(0u, 0xFFu);
    bool has = false;
    __device__ __forceinline__ explicit LaneHeldWriter(uint2* lds) : sw(lds) {}
    __device__ __forceinline__ void flush_held() {
        const uint64_t m = __ballot(has);
        if (m == 0) return;
        const typename SparseWriter<WHICH>::Dest d = SparseWriter<WHICH>::dest();
        const uint32_t b = sw.reserve(d, (uint32_t)__popcll(m));
        const uint32_t rank = mbcnt64(m);
        if (has && b + rank < d.cap) d.out[b + rank] = held;
        has = false;
    }
    // all lanes of the (converged) wave call this
    __device__ __forceinline__ void append(bool emit, const uint2& v) {
        if (__ballot(emit && has)) flush_held();
        if (emit) { held = v; has = true; }
    }
    __device__ __forceinline__ void finish() { flush_held(); sw.finish(); }
};

constexpr uint32_t V4_STAGE = CAND_STAGE * sizeof(Candidate) / sizeof(uint2);
struct V4Lookup {
    uint2* stage;        // V4_STAGE entries of LDS owned by this wave (the CandWriter's buffer)
    uint32_t cnt = 0;    // wave-uniform
};
// length of the canonical dotted-quad text of an address (what the extractor accepted: no leading zeros)
__device__ __forceinline__ uint32_t v4_text_len(uint32_t a) {
    uint32_t n = 7;   // four digits and three dots
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint32_t o = (a >> (8 * k)) & 0xFF; n += (o > 9) + (o > 99); }
    return n;
}
__device__ __forceinline__ void v4_lookup_flush(V4Lookup& vl) {
    if (vl.cnt == 0) return;
    const uint32_t lane = lane_id();
    __builtin_amdgcn_wave_barrier();
    const bool have = lane < vl.cnt;
    uint2 e = make_uint2(0u, 0u);
    if (have) e = vl.stage[lane];
    uint32_t off = 0, pfx = 0;
    const ColdDb kd = cold_db();
    const IpTables tabs{kd->ip_l24, kd->ip_l1, kd->ip_leaf, kd->ip_nodes, kd->node_count};
    const bool hit = have && trie_v4_tables(tabs, e.y, off, pfx);
    const uint64_t m = __ballot(hit);
    if (m) {
        const ColdTok kp = cold_tok();
        ScanCounters* const ctr = kp->pk.counters;
        if (uint2* const c4 = kp->pk.c4_out) {   // compact records (MATCHY_SCAN_FETCH_COMPACT)
            uint2* const host_c4 = kp->pk.host_c4;
            const uint32_t c4_cap = kp->pk.c4_cap, host_cap = kp->pk.host_c4_cap;
            uint32_t slot0 = 0;
            if (lane == 0) slot0 = atomicAdd(&ctr->n_c4, (uint32_t)__popcll(m));
            slot0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot0);
            if (hit) {
                const uint32_t slot = slot0 + mbcnt64(m);
                const uint2 r = c4_pack(e.x, v4_text_len(e.y), off, pfx);
                if (slot < c4_cap) c4[slot] = r;
                if (slot < host_cap) host_c4[slot] = r;
            }
        } else {
            FinalHit* const out = kp->pk.out;
            FinalHit* const host_out = kp->pk.host_out;
            const uint32_t out_cap = kp->pk.out_cap, host_cap = kp->pk.host_cap;
            uint32_t slot0 = 0;
            if (lane == 0) slot0 = atomicAdd(&ctr->n_final, (uint32_t)__popcll(m));
            slot0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot0);
            if (hit) {
                const uint32_t slot = slot0 + mbcnt64(m);
                FinalHit f{};
                f.start = e.x; f.len_type = v4_text_len(e.y) | ((uint32_t)IT_IPV4 << 24);
                f.value = off; f.kind = 2; f.prefix_len = (uint8_t)pfx; f.n_ids = 0;
                if (slot < out_cap) out[slot] = f;
                if (slot < host_cap) host_out[slot] = f;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    vl.cnt = 0;
}

template <bool INL>
__device__ __forceinline__ void commit_v4(PendingV4& pd, const WaveCtx& cx, CandWriter& cw, V4Lookup& vl) {
    const TokParams& p = *cx.p;
    const bool emit = pd.ok && ((pd.word >> ((pd.c.v4 >> 8) & 31)) & 1);
    if constexpr (INL) {
        const uint64_t m = __ballot(emit);
        if (m) {
            // the emitting lanes append in lane order; whenever the stage is full it is flushed (every entry used), the rest follows
            int32_t idx = (int32_t)(vl.cnt + mbcnt64(m));          // position of this lane's entry in the stream of entries
            uint32_t total = vl.cnt + (uint32_t)__popcll(m);        // wave-uniform
            for (;;) {
                if (emit && idx >= 0 && idx < (int32_t)V4_STAGE) vl.stage[idx] = make_uint2(pd.c.start, pd.c.v4);
                if (total < V4_STAGE) { vl.cnt = total; break; }
                vl.cnt = V4_STAGE;
                v4_lookup_flush(vl);
                idx -= (int32_t)V4_STAGE;
                total -= V4_STAGE;
            }
        }
    } else {
        cw.append(emit, pd.c, p.cands_a, p.cand_a_cap, &p.counters->n_cand_a, Candidate{0u, 0xFFFFFFFFu, 0u, 0u});
    }
    pd.ok = false;
}

// IPv4 dotted-quad rules (ext:813-869, 1120-1179) on the 20 bytes w[0] = [dot-4, dot), w[1..4] = [dot, dot+16) of a '.':
// accepts iff the maximal [0-9.] run around the dot is a valid dotted quad whose FIRST dot this is, delimited by
// boundaries (N1). Octet values come from v_dot4_u32_u8 (digit bytes x decimal weights), the digit tests are byte-lane
// SWAR, and the rejections are collected in one integer (`bad`) instead of per-test branch masks.
__device__ __forceinline__ bool d_ipv4_lean(const uint32_t (&w)[5], const uint8_t* ctab, uint32_t dot, uint32_t& start, uint32_t& end, uint32_t& addr) {
    constexpr uint32_t WEIGHTS = 0x00010A64u;   // bytes 0..2 = 100, 10, 1
    // bit 7 of every byte of (bytes ^ '0') that is not a digit value 0..9
    auto nondigit = [](uint32_t x) { return (((x & 0x7F7F7F7Fu) + 0x76767676u) | x) & 0x80808080u; };
    // first octet, right to left from dot-1 (byte 3 of w[0]): n1 = digits in front of the dot (a 4th digit shows up as a
    // non-boundary byte in front of the octet)
    const uint32_t x0 = w[0] ^ 0x30303030u;
    const uint32_t lz = (uint32_t)__clz((int)(nondigit(x0) | 0x80u));   // 0, 8, 16 or 24
    const uint32_t before = (w[0] >> ((24u - lz) & 31u)) & 0xFFu;
    uint32_t bad = (lz == 0) | !(ctab[before] & C_B);
    const uint32_t d0 = x0 >> ((32u - lz) & 31u);                        // the octet's digits, most significant in byte 0
    uint32_t a = __builtin_amdgcn_udot4(d0, WEIGHTS >> ((24u - lz) & 31u), 0u, false);
    bad |= (a >> 8) | ((lz > 8) & ((d0 & 0xFFu) == 0));                  // > 255, leading zero
    // r0..r2 = the 12 bytes after the dot
    uint32_t r0 = __builtin_amdgcn_alignbyte(w[2], w[1], 1), r1 = __builtin_amdgcn_alignbyte(w[3], w[2], 1), r2 = __builtin_amdgcn_alignbyte(w[4], w[3], 1);
    uint32_t pos = dot + 1;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const uint32_t x = r0 ^ 0x30303030u;
        const uint32_t f = (uint32_t)__builtin_ctz(nondigit(x) | 0x80000000u);   // 7, 15, 23, 31 <-> 0..3 digits (a 4th digit = bad separator)
        const uint32_t n = f >> 3;
        const uint32_t sep = (r0 >> (f - 7u)) & 0xFFu;
        const uint32_t v = __builtin_amdgcn_udot4(x, WEIGHTS >> (31u - f), 0u, false);
        bad |= (f == 7) | (v >> 8) | ((f > 15) & ((x & 0xFFu) == 0));
        a = (a << 8) | v;
        pos += n;
        if (o < 2) {
            bad |= sep ^ (uint32_t)'.';
            pos += 1;
            // drop n + 1 bytes: first n (0..3), then one more
            r0 = __builtin_amdgcn_alignbyte(r1, r0, n); r1 = __builtin_amdgcn_alignbyte(r2, r1, n); if (o == 0) r2 = __builtin_amdgcn_alignbyte(0u, r2, n);
            r0 = __builtin_amdgcn_alignbyte(r1, r0, 1); if (o == 0) { r1 = __builtin_amdgcn_alignbyte(r2, r1, 1); }
        } else {
            bad |= !(ctab[sep] & C_B);
        }
    }
    start = dot - (lz >> 3); end = pos; addr = a;
    return bad == 0;
}

template <bool INL>
__device__ __forceinline__ void drain_v4(uint32_t* ring, uint32_t& head, uint32_t& tail, uint32_t n, bool final, const WaveCtx& cx,
                                         PendingV4& pd, CandWriter& cw, V4Lookup& vl) {
    const uint32_t lane = lane_id();
    const TokParams& p = *cx.p;
    __builtin_amdgcn_s_setprio(PRIO_DRAIN);
    commit_v4<INL>(pd, cx, cw, vl);
    __builtin_amdgcn_wave_barrier();
    const bool have = lane < n;
    uint32_t ent = 0;
    if (have) ent = ring[(head + lane) & (QCAP - 1)];
    const uint32_t dot = anchor_pos(ent);
    head += n;
    // an anchor in the last bytes of the newest block has its look-ahead in the block that is not staged yet: it goes
    // back into the ring and is validated by a later drain
    const bool later = have && !final && dot + 16 > cx.res_hi;
    const uint64_t rm = __ballot(later);
    if (rm) {
        __builtin_amdgcn_wave_barrier();
        if (later) ring[(tail + mbcnt64(rm)) & (QCAP - 1)] = ent;
        tail += (uint32_t)__popcll(rm);
    }
#ifdef MXY_ANCHOR_DEBUG
    if (p.debug & 1) { __builtin_amdgcn_s_setprio(PRIO_CHAIN); return; }
#endif
    const bool go = have && !later;
    // the 20 bytes around the dot come from the LDS window (positions >= len are staged as ' ', a boundary like the end of the
    // buffer); only anchors in the first bytes of a segment or at its very end (final drain) read the log itself
    const bool in_window = dot >= cx.res_lo + 4 && dot + 16 <= cx.res_hi;
    // every lane parses the 20 bytes at its window address, wanted or not (one definition of s / e / a instead of zeroed registers
    // merged on every path); lanes without an anchor in the window drop the result
    uint32_t s, e, a;
    uint32_t w[5];
    raw_read<5>(cx.raw32, dot - 4, w);
    bool ok = d_ipv4_lean(w, cx.ctab, dot, s, e, a) && go && in_window;
    if (__ballot(go && !in_window)) {
        if (go && !in_window) ok = val_ipv4(LogView{p.log, p.len}, dot, s, e, a);
    }
    if (ok) {
        pd.c.start = s; pd.c.len_type = (e - s) | ((uint32_t)IT_IPV4 << 24); pd.c.v4 = a;
        pd.ok = true;
        pd.n_valid += 1;
        pd.word = p.filter_v4 ? cx.bm24[a >> 13] : 0xFFFFFFFFu;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_setprio(PRIO_CHAIN);
}

// Prefilter up to 64 domain anchors (first byte of the last label), one per lane; survivors go to the domain list.
// Mirrors the first steps of val_domain (validate_kernels.hip): a '.' inside the label means a later dot owns the run, the
// run must end at a boundary, and the label must be the last label of some public suffix. Undecidable cases (label
// longer than 8 bytes, bytes not resident) are kept.
__device__ __forceinline__ void drain_dom(uint32_t* ring, uint32_t& head, uint32_t& tail, uint32_t n, bool final, const WaveCtx& cx,
                                          DomWriter& dw) {
    const uint32_t lane = lane_id();
    const TokParams& p = *cx.p;
    __builtin_amdgcn_s_setprio(PRIO_DRAIN);
    __builtin_amdgcn_wave_barrier();
    uint32_t ent = 0;
    const bool have = lane < n;
    if (have) ent = ring[(head + lane) & (QCAP - 1)];
    const uint32_t j = have ? anchor_pos(ent) : 0xFFFFFFFFu;
    head += n;
    // label window not staged yet (anchor in the last bytes of the newest block): back into the ring for a later drain
    const bool later = have && !final && j + 8 > cx.res_hi;
    const uint64_t rm = __ballot(later);
    if (rm) {
        __builtin_amdgcn_wave_barrier();
        if (later) ring[(tail + mbcnt64(rm)) & (QCAP - 1)] = ent;
        tail += (uint32_t)__popcll(rm);
    }
#ifdef MXY_ANCHOR_DEBUG
    if (p.debug & 2) { __builtin_amdgcn_s_setprio(PRIO_CHAIN); return; }
#endif
    // Every lane reads its 32 context bytes, wanted or not (a window address is a window address; lanes without an anchor, or with
    // one whose bytes are not resident, read bytes nobody looks at): ONE definition of ctx[] — as conditional assignments the eight
    // registers were zeroed and merged on every path, 16 moves per drain.
    const bool go = have && !later;
    const bool have_ctx = go && j >= cx.res_lo + 24 && j + 8 <= cx.res_hi;
    uint32_t ctx[8];
    raw_read<8>(cx.raw32, j - 24, ctx);   // log[j-24, j+8): the last label starts at byte 24
    bool keep = go;   // anchors that cannot be decided here (bytes not resident, long last label) are kept
    {
        // the label's 8-byte window as SWAR masks (no per-byte loop)
        constexpr uint64_t H = 0x8080808080808080ull;
        const uint64_t w64 = (uint64_t)ctx[6] | ((uint64_t)ctx[7] << 32);
        const ByteMasks mw = domain_masks(w64);
        const uint64_t ndc = ~mw.dc & H;
        if (ndc) {   // the label ends inside the window: ll = its length (>= 1: byte j can start a TLD)
            const uint32_t ll = (uint32_t)(__ffsll((long long)ndc) - 1) >> 3;
            const uint64_t below = (1ull << (8 * ll)) - 1ull;
            const uint32_t stop = (uint32_t)(w64 >> (8 * ll)) & 0xFF;
            const uint32_t h8 = tld_hash8((uint32_t)(w64 & below), (uint32_t)((w64 & below) >> 32));
            const uint32_t bit = h8 & (TLD_BLOOM_BITS / 2 - 1), bit2 = (h8 >> 14) & (TLD_BLOOM_BITS / 2 - 1);
            // a '.' inside the label: a later dot owns the run; the run must end at a boundary; the label must be
            // some public suffix's last label (two bits of the prefilter's Bloom filter)
            const bool k = (mw.dot & below) == 0 && d_is_boundary(stop) && ((cx.bloom[bit >> 5] >> (bit & 31)) & (cx.bloom[bit2 >> 5] >> (bit2 & 31)) & 1);
            if (have_ctx) keep = k;
        } else if (mw.dot) {
            if (have_ctx) keep = false;         // 8 domain chars with a dot among them: a later dot owns the run
        } else if (have_ctx && j + 24 <= cx.res_hi) {
            // a label of 8+ bytes: usually not the last one ("www.examplesite.com" at 'e'). Look 16 bytes further: a
            // dot before the first non-domain byte settles it; otherwise it stays undecided (long last label)
            uint32_t more[4];
            raw_read<4>(cx.raw32, j + 8, more);
            const ByteMasks ma = domain_masks((uint64_t)more[0] | ((uint64_t)more[1] << 32));
            const ByteMasks mb = domain_masks((uint64_t)more[2] | ((uint64_t)more[3] << 32));
            const uint64_t na = ~ma.dc & H, nb = ~mb.dc & H;
            const uint64_t below_a = na ? ((1ull << ((uint32_t)(__ffsll((long long)na) - 1) & ~7u)) - 1ull) : ~0ull;
            const uint64_t below_b = na ? 0ull : (nb ? ((1ull << ((uint32_t)(__ffsll((long long)nb) - 1) & ~7u)) - 1ull) : ~0ull);
            keep = ((ma.dot & below_a) | (mb.dot & below_b)) == 0;
        }
    }
    const uint32_t slot = dw.reserve(keep, p.dom_list, p.dom_cap, [] { return &cold_tok()->counters->n_dom; },
                                     [] { const uint32_t st = cold_tok()->dom_static; return st ? (blockIdx.x * 4u + (threadIdx.x >> 6)) * st : 0xFFFFFFFFu; },
                                     [] { return cold_tok()->dom_chunk; });
    if (slot != 0xFFFFFFFFu) {
        uint32_t* rec = p.dom_list + dom_plane_index(slot, 0);   // the planes of a 64-slot tile are 256 bytes apart
        rec[0] = have_ctx ? j : (j | 0x80000000u);
#pragma unroll
        for (int k = 0; k < 8; ++k) rec[(1 + k) * DOM_TILE] = ctx[k];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_setprio(PRIO_CHAIN);
}

// The shipped public-suffix list only has last labels that start with 'a'..'z' or a byte >= 0x80 (ClassPlanes::TL); any other
// list switches the first-byte class to "any label byte or '-'".
__host__ __device__ inline bool anchor_tl_wide(const DevDb& db) {
    uint32_t outside = 0;
    for (int k = 0; k < 8; ++k) {
        const uint32_t allowed = k >= 4 ? 0xFFFFFFFFu : (k == 3 ? 0x07FFFFFEu : 0u);   // 0x61..0x7A, 0x80..0xFF
        outside |= db.tld_first[k] & ~allowed;
    }
    return outside != 0;
}

// ---- cross-lane plumbing of the bit planes
constexpr int DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138, DPP_WAVE_ROR1 = 0x13C;
// Plane word of the PREVIOUS dword of the byte stream for every lane: lane L-1's word; lane 0 gets lane 63's word moved
// one row up (bit 8 b + q <- bit 8 b + q - 1) with the last row of the previous block (`carry`: that block's lane-63 word,
// whose bits 8 b + 7 are row 7) coming in as row -1. Updates `carry` for the next block.
__device__ __forceinline__ uint32_t plane_prev_dword(uint32_t P, uint32_t& carry) {
    const uint32_t s63 = (uint32_t)__builtin_amdgcn_readlane((int)P, 63);
    const uint32_t fix0 = ((s63 << 1) & 0xFEFEFEFEu) | ((carry >> 7) & 0x01010101u);
    carry = s63;
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fix0, (int)P, DPP_WAVE_SHR1, 0xF, 0xF, false);
}
// Plane word of the NEXT dword: lane L+1's word; lane 63 gets lane 0's word moved one row down, and `ahead` for the
// first dword of the next block (not loaded yet: all ones keeps look-ahead tests conservative)
__device__ __forceinline__ uint32_t plane_next_dword(uint32_t P, uint32_t ahead) {
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)P, 0);
    const uint32_t fix63 = ((s0 >> 1) & 0x7F7F7F7Fu) | ahead;
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fix63, (int)P, DPP_WAVE_SHL1, 0xF, 0xF, false);
}
// inclusive prefix sum over the 64 lanes: four row shifts (zeros shifted in), then the row totals broadcast into the following rows
__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return v;
}
// class of the byte k positions earlier / later (k = 1..4), aligned to this position
template <int K> __device__ __forceinline__ uint32_t back(uint32_t P, uint32_t PV) { return K == 4 ? PV : __builtin_amdgcn_alignbyte(P, PV, 4 - K); }
template <int K> __device__ __forceinline__ uint32_t ahead(uint32_t P, uint32_t NV) { return K == 4 ? NV : __builtin_amdgcn_alignbyte(NV, P, K); }

// ALL: every extractor is enabled and the public-suffix first-byte class is the narrow one (the command line's and the
// bulk scan's configuration): no run-time flag tests in the block loop.
// INL: TokParams::inline_v4 (the wave looks its IPv4 candidates up itself)
template <bool ALL, bool INL>
__global__ __launch_bounds__(AW * 64) void k_anchor(TokParams p, DevDb db) {
    constexpr uint32_t RAW_DW = RAW_BYTES / 4;
    __shared__ uint8_t ctab[256];    // byte classes for the few bytes in front of a segment (the blocks themselves are bit-sliced)
    __shared__ uint32_t bloom[BLOOM_FOLD_WORDS];
    __shared__ __attribute__((aligned(16))) uint32_t rawst[AW][RAW_DW + RAW_MIRROR / 4];
    __shared__ uint32_t q_v4[AW][QCAP];
    __shared__ uint32_t q_dom[AW][QCAP];
    __shared__ uint2 wb_misc[AW][RARE_STAGE], wb_tok[AW][RARE_STAGE];   // BufferedWriter staging
    __shared__ Candidate wb_cand[AW][CAND_STAGE];

    ctab[threadIdx.x] = (uint8_t)class_of(threadIdx.x);
    for (uint32_t i = threadIdx.x; i < BLOOM_FOLD_WORDS; i += AW * 64) bloom[i] = db.tld_bloom[TLD_BLOOM_WORDS + i];   // the prefilter's own filter
    __syncthreads();

    // readfirstlane: tells the compiler that the wave index — and with it the segment loop, the block position and the ring
    // heads / tails — is wave-uniform (scalar registers and scalar branches instead of vector ones)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * AW + wave, nw = gridDim.x * AW;
    uint32_t* raw32 = rawst[wave];
    uint32_t* rv4 = q_v4[wave];
    uint32_t* rdom = q_dom[wave];
    const uint32_t len = p.len;
    const bool en_v4 = ALL || (p.flags & EX_IPV4) != 0, en_dom = ALL || (p.flags & EX_DOMAINS) != 0;
    const bool en_v6 = ALL || (p.flags & EX_IPV6) != 0, en_at = ALL || (p.flags & EX_EMAILS) != 0;
    const bool en_tok = ALL || (p.flags & (EX_HASHES | EX_BITCOIN | EX_ETHEREUM | EX_MONERO)) != 0;
    const bool tl_wide = !ALL && anchor_tl_wide(db);

    uint32_t nl_count = 0;                       // per-lane '\n' count, reduced once at the end
    uint32_t v4h = 0, v4t = 0, dh = 0, dt = 0;   // ring heads / tails (wave-uniform)
    uint32_t v4_old = 0, dom_old = 0;            // block start of the oldest ring entry (valid while the ring is non-empty)
    CandWriter cw_cand(wb_cand[wave], p.cand_chunk);   // IPv4 candidates: sparse when the /24 bitmap filters, else one per line
    V4Lookup vl{reinterpret_cast<uint2*>(wb_cand[wave]), 0u};   // TokParams::inline_v4: the same LDS holds {start, address} pairs
    DomWriter cw_dom;
    SparseWriter<0> cw_misc(wb_misc[wave]);   // rare anchors and long tokens are sparse: dense lists
    LaneHeldWriter<1> cw_tok(wb_tok[wave]);
    WaveCtx cx{&p, raw32, ctab, bloom, db.ip_bm24, 0u, 0u};
    PendingV4 pend;
    const uint32_t lane_off = lane << 2;
    const uint32_t lane0_one = lane == 0 ? 1u : 0u;   // shift count: the value that wraps from lane 63 into lane 0 moves one row up (x << 1 there, x elsewhere)

    // dword `lane` of the 8 rows of block `b`; positions >= len read as ' ' (a boundary, like the end of the buffer)
    auto load_block = [&](uint32_t b, uint32_t (&w)[8]) {
        if (b + BLK_BYTES <= len) {
            // streamed once: non-temporal loads keep the log from pushing the database's /24 bitmap (the one table this
            // kernel reads at random) out of L2
            const uint32_t* src = reinterpret_cast<const uint32_t*>(p.log + b) + lane;
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = __builtin_nontemporal_load(src + 64 * q);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t pos0 = b + AB_ROW_BYTES * q + lane_off;
                uint32_t x = 0x20202020u;
                if (pos0 + 4 <= len) {
                    x = *reinterpret_cast<const uint32_t*>(p.log + pos0);
                } else if (pos0 < len) {
                    x = 0;
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) x |= (pos0 + bb < len ? (uint32_t)p.log[pos0 + bb] : (uint32_t)' ') << (8 * bb);
                }
                w[q] = x;
            }
        }
    };

    __builtin_amdgcn_s_setprio(PRIO_CHAIN);
    for (uint32_t seg = gw;; seg += nw) {
        const ColdTok ks = cold_tok();   // the segment geometry is needed once per segment: not held through the block loop
        if (seg >= ks->n_segs) break;
        const uint32_t seg_start = ks->seg_base + seg * ks->seg_bytes;
        // positions 0..len are scanned: position `len` (padding, class "boundary") closes a trailing token (scan_end = len + 1
        // for the launch that covers the end of the batch)
        const uint32_t seg_end = min(seg_start + ks->seg_bytes, ks->scan_end);
        // Plane carries = the word a lane 63 of a block in front of the segment would hold; only its row-7 bits (8 b + 7 =
        // class of byte seg_start - 4 + b) are ever used. In front of the buffer: boundary.
        uint32_t cB = 0x80808080u, cX = 0;   // cX: carry of the packed word X = [C.b2, C.b3, D.b3, T.b3] (see the block loop)
        if (seg_start) {
            uint32_t c4 = 0;
            for (uint32_t k = 0; k < 4; ++k) c4 |= (uint32_t)ctab[p.log[seg_start - 4 + k]] << (8 * k);
            c4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c4);
            static_assert(C_B == 1 && C_DIG == 2 && C_DOT == 4 && C_COLON == 8 && C_LD == 32, "carry shifts");
            cB = (c4 << 7) & 0x80808080u;
            const uint32_t c2 = (c4 >> 16) & 0xFFu, c3 = c4 >> 24;   // classes of the bytes at seg_start - 2 and seg_start - 1
            cX = ((c2 << 4) & 0x80u) | (((c3 << 4) & 0x80u) << 8) | (((c3 << 6) & 0x80u) << 16) | (((c3 << 5) & 0x80u) << 24);
        }
        // Token state: the boundary plane of the previous block (row 7 = the 256 bytes in front of this block) and its
        // "dword holds no boundary byte" bits. In front of the buffer: a boundary at position -1.
        uint32_t Bprev = lane == 63 ? 0x80000000u : 0u, Aprev = 0;
        if (en_tok && seg_start) {
            const uint32_t x = *reinterpret_cast<const uint32_t*>(p.log + seg_start - AB_ROW_BYTES + lane_off);
            const uint32_t c = (uint32_t)ctab[x & 0xFF] | ((uint32_t)ctab[(x >> 8) & 0xFF] << 8) | ((uint32_t)ctab[(x >> 16) & 0xFF] << 16) |
                               ((uint32_t)ctab[x >> 24] << 24);
            Bprev = (c << 7) & 0x80808080u;
            // all four bytes ASCII alphanumerics (label bytes below 0x80): row 7 of the previous block for the token chain
            Aprev = ((c & (C_LD * 0x01010101u)) == C_LD * 0x01010101u && (x & 0x80808080u) == 0) ? 0x80u : 0u;
        }

        uint32_t nx[8];
        load_block(seg_start, nx);
        for (uint32_t blk = seg_start; blk < seg_end; blk += BLK_BYTES) {
            // ---- the prefetched block is taken HERE — the first stage of the bit transpose reads it, which is where the wait for its loads
            // goes — and not where the window write needs it, behind the drains below: the counter the wait uses (vmcnt) counts loads and
            // stores alike, in order, so behind a drain it also sits out the latency of the stores (and the /24 bitmap load) that drain has
            // issued a moment ago. 0.795 -> 0.766 ms. The stage writes new registers (bit_transpose8_first): the raw bytes stay where they
            // were loaded until the window has them, and the next block is then loaded into the same registers — no copies. (Moving the
            // drains themselves — all of them in front of the prefetch, or the domain drain behind it — costs more instructions than the
            // waits it saves: 0.772-0.803 ms.)
            uint32_t w[8];
            bit_transpose8_first(nx, w);
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(w[q]));   // ... here, not sunk to the rest of the transpose behind the drains
            // ---- anchors whose look-back bytes are about to be overwritten in the raw window leave first
            if (blk >= seg_start + RAW_BYTES - BLK_BYTES) {
                const uint32_t lim = blk - (RAW_BYTES - BLK_BYTES);   // entries of blocks <= lim expire
                if (v4t != v4h && v4_old <= lim) { drain_v4<INL>(rv4, v4h, v4t, v4t - v4h, false, cx, pend, cw_cand, vl); v4_old = blk - BLK_BYTES; }
                if (dt != dh && dom_old <= lim) { drain_dom(rdom, dh, dt, dt - dh, false, cx, cw_dom); dom_old = blk - BLK_BYTES; }
            }
            // ---- this lane's 8 dwords: raw bytes into the window (natural byte order)
            __builtin_amdgcn_wave_barrier();
            {
                uint32_t* dst = &raw32[((blk & (RAW_BYTES - 1)) >> 2) + lane];
#pragma unroll
                for (int q = 0; q < 8; ++q) dst[64 * q] = nx[q];
                if ((blk & (RAW_BYTES - 1)) == 0 && lane < RAW_MIRROR / 4) raw32[RAW_DW + lane] = nx[0];   // mirror of the window's first bytes
            }
            __builtin_amdgcn_wave_barrier();
            // the next block, always: behind the last block of a segment this reads a block of the next segment or the padded end of the
            // batch, and nobody looks at it (a conditional load would make the compiler keep the registers defined on the other path too)
            load_block(blk + BLK_BYTES, nx);
            cx.res_hi = blk + BLK_BYTES;
            cx.res_lo = cx.res_hi - seg_start > RAW_BYTES ? cx.res_hi - RAW_BYTES : seg_start;
            __builtin_amdgcn_s_setprio(PRIO_BULK);   // transpose + class functions: independent instructions, they fill gaps

            // ---- bit planes and byte classes of the lane's 32 positions: bit t <-> position blk + 256 (t & 7) + 4 lane + (t >> 3)
            bit_transpose8_rest(w);
            const ClassPlanes cl = classify_planes(w, tl_wide);
            nl_count += __popc(cl.NL);
            __builtin_amdgcn_s_setprio(PRIO_CHAIN);  // from here on cross-lane steps, LDS and memory: see PRIO_CHAIN
            const uint32_t pos_base = blk + lane_off;
            const uint32_t ent_base = blk | (lane << 5);   // ring entries: anchor_pos()

            uint32_t Fd = 0, F4 = 0, F6 = 0;
            // The patterns need one byte of the previous dword's '.' and digit planes and two of its ':' plane: they travel
            // together as X = [C.b2, C.b3, D.b3, T.b3] (one cross-lane step instead of three; `v_perm` puts the bytes in place)
            const uint32_t X = __builtin_amdgcn_perm(cl.T, __builtin_amdgcn_perm(cl.D, cl.C, 0x07070302u), 0x07020100u);
            const uint32_t PV_X = plane_prev_dword(X, cX);
            if (en_v4) {
                // '.' at j, digit at j-1, a boundary 2..4 positions back ...
                const uint32_t PV_B = plane_prev_dword(cl.B, cB);
                const uint32_t lookback = __builtin_amdgcn_perm(cl.D, PV_X, 0x06050402u) & (back<2>(cl.B, PV_B) | back<3>(cl.B, PV_B) | back<4>(cl.B, PV_B));
                // ... and a digit at j+1 and a second dot 2..4 positions ahead (necessary for a dotted quad whose first dot this is;
                // drops "HTTP/1.1", "Mozilla/5.0", "Safari/537.36" style anchors). The drain checks the digits in between.
                const uint32_t NV_D = plane_next_dword(cl.D, 0x80808080u), NV_T = plane_next_dword(cl.T, 0x80808080u);
                const uint32_t lookahead = ahead<1>(cl.D, NV_D) & (ahead<2>(cl.T, NV_T) | ahead<3>(cl.T, NV_T) | ahead<4>(cl.T, NV_T));
                F4 = cl.T & lookback & lookahead;
            }
            if (en_dom) {
                // byte that can start a public suffix's last label at j, '.' at j-1 (what stands at j-2 is the validators' business)
                Fd = cl.TL & back<1>(cl.T, PV_X);
            }
            if (en_v6) {
                // "::" ending at j without a third ':'
                F6 = cl.C & __builtin_amdgcn_perm(cl.C, PV_X, 0x06050401u) & ~__builtin_amdgcn_perm(cl.C, PV_X, 0x05040100u);
            }
#ifdef MXY_ANCHOR_DEBUG
            if (p.debug & 4) F4 = 0;
            if (p.debug & 8) Fd = 0;
            if (p.debug & 64) { nl_count += (F4 | Fd | F6) & 1; F4 = 0; Fd = 0; F6 = 0; }
#endif
            // ---- anchors into the rings. One inclusive wave scan of the per-lane anchor counts (both kinds in one word, six DPP adds)
            // gives every lane the ring positions of its own anchors: the lanes then write them without any further cross-lane step
            // (the ballot loop below costs a ballot, two mbcnt and a dozen scalar instructions per anchor of the busiest lane).
            // A block with more anchors than the ring has room for (dense runs of dots) takes the ballot loop, which drains as it goes.
            const uint32_t cnt2 = (en_dom ? (uint32_t)__popc(Fd) : 0u) | ((en_v4 ? (uint32_t)__popc(F4) : 0u) << 16);
            const uint32_t incl2 = wave_incl_scan_add(cnt2);
            const uint32_t tot2 = (uint32_t)__builtin_amdgcn_readlane((int)incl2, 63);
            const uint32_t excl2 = incl2 - cnt2;
            if (en_dom && (tot2 & 0xFFFFu)) {
                const uint32_t total = tot2 & 0xFFFFu;
                if (dt - dh + total <= QCAP) {
                    if (dt == dh) dom_old = blk;
                    uint32_t idx = dt + (excl2 & 0xFFFFu);
                    while (Fd) {
                        rdom[idx & (QCAP - 1)] = ent_base | (uint32_t)__builtin_ctz(Fd);
                        Fd &= Fd - 1;
                        ++idx;
                    }
                    dt += total;
                    while (dt - dh >= 64) { drain_dom(rdom, dh, dt, 64u, false, cx, cw_dom); dom_old = blk; }
                } else {
                    for (;;) {
                        const uint64_t m = __ballot(Fd != 0);
                        if (!m) break;
                        if (Fd) {
                            rdom[mbcnt64_add(m, dt) & (QCAP - 1)] = ent_base | (uint32_t)__builtin_ctz(Fd);
                            Fd &= Fd - 1;
                        }
                        if (dt == dh) dom_old = blk;
                        dt += (uint32_t)__popcll(m);
                        if (dt - dh >= 64) { drain_dom(rdom, dh, dt, 64u, false, cx, cw_dom); dom_old = blk; }
                    }
                }
            }
            if (en_v4 && (tot2 >> 16)) {
                const uint32_t total = tot2 >> 16;
                if (v4t - v4h + total <= QCAP) {
                    if (v4t == v4h) v4_old = blk;
                    uint32_t idx = v4t + (excl2 >> 16);
                    while (F4) {
                        rv4[idx & (QCAP - 1)] = ent_base | (uint32_t)__builtin_ctz(F4);
                        F4 &= F4 - 1;
                        ++idx;
                    }
                    v4t += total;
                    while (v4t - v4h >= 64) { drain_v4<INL>(rv4, v4h, v4t, 64u, false, cx, pend, cw_cand, vl); v4_old = blk; }
                } else {
                    for (;;) {
                        const uint64_t m = __ballot(F4 != 0);
                        if (!m) break;
                        if (F4) {
                            rv4[mbcnt64_add(m, v4t) & (QCAP - 1)] = ent_base | (uint32_t)__builtin_ctz(F4);
                            F4 &= F4 - 1;
                        }
                        if (v4t == v4h) v4_old = blk;
                        v4t += (uint32_t)__popcll(m);
                        if (v4t - v4h >= 64) { drain_v4<INL>(rv4, v4h, v4t, 64u, false, cx, pend, cw_cand, vl); v4_old = blk; }
                    }
                }
            }
            if (en_v6 || en_at) {
                uint32_t FA = en_at ? cl.AT : 0u;
                while (__ballot((F6 | FA) != 0)) {
                    // one anchor per lane and round: "::" anchors first
                    const bool has = (F6 | FA) != 0;
                    uint2 v = make_uint2(0xFFFFFFFFu, 0xFFu);
                    if (has) {
                        const bool six = F6 != 0;
                        const uint32_t f = six ? F6 : FA;
                        const uint32_t t = (uint32_t)(__ffs((int)f) - 1);
                        v = make_uint2(pos_base + ((t & 7) << 8) + (t >> 3), six ? (uint32_t)RARE_V6 : (uint32_t)RARE_AT);
                        if (six) F6 &= F6 - 1; else FA &= FA - 1;
                    }
                    cw_misc.append(has, v);
                }
            }
#ifdef MXY_ANCHOR_DEBUG
            if (!(p.debug & 16))
#endif
            if (en_tok) {
                // Every token the validators accept (hex hashes, Base58 / Bech32 / 0x-hex addresses) consists of ASCII letters
                // and digits only, and a token of >= 26 such bytes that ends in dword i makes dwords i-1 .. i-5 all-alphanumeric.
                // a4: bit q set when all four bytes of this lane's dword of row q are ASCII alphanumerics.
                const uint32_t an = cl.LD & ~w[7];
                const uint32_t a4 = an & (an >> 8) & (an >> 16) & (an >> 24) & 0xFFu;
                // x: bit q + 1 = row q of this lane, bit 0 = row 7 of the previous block. The value travels down the wave (rotate
                // by one lane per step); when it wraps from lane 63 to lane 0 it moves one row up, which is one bit to the left.
                // After k steps x is the state of the dword k places earlier in the byte stream.
                uint32_t x = (a4 << 1) | (Aprev >> 7), r = 0xFFFFFFFFu;
                Aprev = a4;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    x = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, DPP_WAVE_ROR1, 0xF, 0xF, false) << lane0_one;   // every lane has a source: no `old` value to set up
                    r &= x;
                }
                // three all-alphanumeric dwords in a row (12+ such bytes) are rare in logs: the rest only then
                uint32_t cand = 0;
                if (__ballot(r != 0)) {
#pragma unroll
                    for (int k = 3; k < 5; ++k) {
                        x = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, DPP_WAVE_ROR1, 0xF, 0xF, false) << lane0_one;
                        r &= x;
                    }
                    // only the lowest boundary byte of a dword can close a long token
                    cand = (cl.B | (cl.B >> 8) | (cl.B >> 16) | (cl.B >> 24)) & (r >> 1) & 0xFFu;
                }
                if (__ballot(cand != 0)) {
                    // Exact length from the boundary-free dwords (G) around the candidate. One candidate per lane and round, whatever row it
                    // is in (a loop over the eight rows ran its ~60 instructions once per row that holds a candidate: every row of a block
                    // of an endpoint or JSON log, where each line carries hashes or ids). The rows' free-dword masks — eight ballots and the
                    // previous block's last row — are parked in the lanes of one register (lane r: low word of row r - 1, lane 16 + r: high
                    // word; r = 0: the previous block's row 7) and every lane fetches the two rows it needs with ds_bpermute.
                    const uint32_t G = ~(cl.B | (cl.B >> 8) | (cl.B >> 16) | (cl.B >> 24)) & 0xFFu;
                    const uint32_t Gprev = (Bprev & 0x80808080u) ? 0u : 0x80u;   // row 7 of the previous block
                    // (v_writelane through inline assembly: the compiler offers no builtin for it here and does not see the hazard between the
                    // v_cmp that has just written VCC and a v_writelane that reads it — without the wait states in front of each one the lanes
                    // received stale masks)
                    int zt = 0;
                    {
                        const uint64_t z = __ballot((Gprev & 0x80u) != 0);
                        asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, 0" : "+v"(zt) : "s"((uint32_t)z));
                        asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, 16" : "+v"(zt) : "s"((uint32_t)(z >> 32)));
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const uint64_t z = __ballot(((G >> q) & 1u) != 0);
                        asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(zt) : "s"((uint32_t)z), "n"(q + 1));
                        asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(zt) : "s"((uint32_t)(z >> 32)), "n"(q + 17));
                    }
                    uint32_t cleft = cand;
                    while (__ballot(cleft != 0)) {
                        const bool mine = cleft != 0;
                        const uint32_t q = mine ? (uint32_t)__builtin_ctz(cleft) : 0u;   // this lane's row in this round
                        cleft &= cleft - 1;
                        const uint64_t Zq = (uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)(4 * (q + 1)), zt) |
                                            ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)(4 * (q + 17)), zt) << 32);
                        const uint64_t Zprev_row = (uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)(4 * q), zt) |
                                                   ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)(4 * (q + 16)), zt) << 32);
                        // free dwords directly below this one: in this row, then at the end of the previous row
                        const uint64_t below = lane ? (Zq << (64 - lane)) : 0ull;          // bit 63 = dword lane-1
                        uint32_t run = lane ? (uint32_t)__clzll((long long)~below) : 0u;   // ~below != 0 for lane < 64... see min
                        run = min(run, lane);
                        bool too_long = false;
                        if (run == lane) {
                            const uint32_t r2 = ~Zprev_row ? (uint32_t)__clzll((long long)~Zprev_row) : 64u;
                            too_long = r2 == 64u;   // 64 + free dwords = more than 256 bytes: longer than any token
                            run += r2;
                        }
                        // the dword below the run holds the last boundary byte before the token
                        const int32_t it = (int32_t)(64 * q + lane) - (int32_t)run - 1;    // dword index in the block (negative: previous block)
                        const uint32_t src_lane = (uint32_t)it & 63u;
                        const uint32_t wb_cur = (uint32_t)__shfl((int)cl.B, (int)src_lane);
                        const uint32_t wb_prev = (uint32_t)__shfl((int)Bprev, (int)src_lane);
                        const uint32_t rowbits = it >= 0 ? ((wb_cur >> (it >> 6)) & 0x01010101u) : ((wb_prev >> 7) & 0x01010101u);
                        const uint32_t hb = rowbits ? (31u - (uint32_t)__clz((int)rowbits)) >> 3 : 0u;
                        const int32_t s = (int32_t)blk + 4 * it + (int32_t)hb + 1;         // token start
                        const uint32_t mybits = (cl.B >> q) & 0x01010101u;
                        const uint32_t b0 = mybits ? ((uint32_t)__ffs((int)mybits) - 1u) >> 3 : 0u;
                        const uint32_t e = blk + AB_ROW_BYTES * q + lane_off + b0;          // closing boundary
                        const uint32_t tl = e - (uint32_t)s;
                        const bool tok = mine && !too_long && rowbits != 0 &&
                                         ((tl >= 26 && tl <= 62) || tl == 64 || (tl >= 90 && tl <= 110) || tl == 128);
                        cw_tok.append(tok, make_uint2((uint32_t)s, (uint32_t)RARE_TOK | (tl << 8)));
                    }
                }
                Bprev = cl.B;
            }
        }
        // the next segment of this wave is not contiguous: finish the rings while their bytes are still in the window
        if (dt != dh) drain_dom(rdom, dh, dt, dt - dh, true, cx, cw_dom);
        if (v4t != v4h) drain_v4<INL>(rv4, v4h, v4t, v4t - v4h, true, cx, pend, cw_cand, vl);
    }
    commit_v4<INL>(pend, cx, cw_cand, vl);
    if constexpr (INL) v4_lookup_flush(vl);
    cw_misc.finish();
    cw_tok.finish();
    // mark the unused tail of every open chunk
    if (cw_dom.next == 0xFFFFFFFFu && cold_tok()->dom_static) {   // a wave that met no domain anchor: its own chunk (TokParams::dom_static) is all sentinels
        cw_dom.next = (blockIdx.x * 4u + (threadIdx.x >> 6)) * cold_tok()->dom_static;
        cw_dom.left = cold_tok()->dom_static;
    }
    cw_dom.pad_rest(p.dom_list, p.dom_cap);
    cw_cand.finish(p.cands_a, p.cand_a_cap, &p.counters->n_cand_a, Candidate{0u, 0xFFFFFFFFu, 0u, 0u});
    {
        uint32_t nv = pend.n_valid;  // validated IPv4 candidates, listed or not
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nv += __shfl_down(nv, off);
        if (lane == 0 && nv) atomicAdd(&cold_tok()->counters->cand_true, nv);
    }
    // line count: wave reduction of the per-lane counts, one atomic per wave
    unsigned long long lines = nl_count;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lines += __shfl_down(lines, off);
    if (lane == 0 && lines) atomicAdd(&cold_tok()->counters->lines, lines);
}

// workgroups of k_anchor that are resident on one CU at the same time (register / LDS limited)
int anchor_blocks_per_cu() {
    int n = 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_anchor<true, true>, AW * 64, 0);
    if (e != hipSuccess || n < 1) n = 3;
    return n;
}

void launch_anchor(const TokParams& p, const DevDb& db, int grid, hipStream_t stream) {
    const bool all = (p.flags & EX_ALL) == EX_ALL && !anchor_tl_wide(db);
    if (all && p.inline_v4) hipLaunchKernelGGL((k_anchor<true, true>), dim3(grid), dim3(AW * 64), 0, stream, p, db);
    else if (all) hipLaunchKernelGGL((k_anchor<true, false>), dim3(grid), dim3(AW * 64), 0, stream, p, db);
    else if (p.inline_v4) hipLaunchKernelGGL((k_anchor<false, true>), dim3(grid), dim3(AW * 64), 0, stream, p, db);
    else hipLaunchKernelGGL((k_anchor<false, false>), dim3(grid), dim3(AW * 64), 0, stream, p, db);
    check_launch("launch_anchor");
}

}  // namespace mxy
