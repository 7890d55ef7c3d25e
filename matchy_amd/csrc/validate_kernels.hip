// HIP kernels of stage A2 / A3 of the `matchy match` hot path on gfx950 (MI355X, wave64): validation of the anchors that
// k_anchor (stage A1, k_anchor.hip) lists.
//
//   k_validate_dom  stage A2a: one lane per domain anchor with its 32-byte context record: suffix table, label rules as
//               mask arithmetic, XXH64 from registers + literal bitmap (+ the AC-DFA walk for databases with globs).
//               Lean and branch-free on purpose.
//   k_validate  stage A2b: one lane per long token (hex hashes; prefix tests for the address formats) and per rare
//               anchor: IPv6, e-mail, and the domains k_validate_dom leaves undecided (general right-to-left walk).
//   k_rare      stage A3: checksum validators that are rare in logs and heavy in registers (Base58Check, Bech32,
//               EIP-55, Monero) — one lane per prefiltered token.
//
// Semantics follow the reference CPU path; every rule cites the reference function it reproduces
// (matchy-extractor/src/lib.rs = "ext"). The sequential `last_end` / `last_domain_end` state of the reference is
// replaced by stateless per-run ownership rules, differential-tested against oracle/.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "hashes.h"
#include "scan_types.h"

#include "device_shared.h"

namespace mxy {


// ------------------------------------------------------------------------------------------------ stage A validators
// Backward byte reader over log[.., pos): 8 bytes per load, the next 8 prefetched while the current ones are consumed,
// so the dependent-load chain of a right-to-left walk is one L1/L2 round trip per 8 bytes instead of one per byte.
struct BackReader {
    const uint8_t* p;
    uint32_t pos;   // next() returns p[pos - 1]
    uint32_t cb;    // cur holds p[cb, cb + 8), shifted so that the unread bytes [cb, cb + k) sit at the top
    uint32_t k;
    uint64_t cur, nxt;
    __device__ __forceinline__ static uint64_t load8(const uint8_t* a) { uint64_t v; __builtin_memcpy(&v, a, 8); return v; }
    __device__ __forceinline__ void init(const uint8_t* base, uint32_t q) {
        p = base; pos = q; cb = 0; k = 0; cur = 0; nxt = 0;
        if (q >= 16) { cb = q - 8; k = 8; cur = load8(p + cb); nxt = load8(p + cb - 8); }
    }
    __device__ __forceinline__ uint32_t next() {
        --pos;
        if (pos < 16) return p[pos];  // the first bytes of the buffer: plain loads (no read before the buffer)
        if (k == 0) { cur = nxt; cb -= 8; k = 8; nxt = cb >= 8 ? load8(p + cb - 8) : 0ull; }
        const uint32_t c = (uint32_t)(cur >> 56);
        cur <<= 8;
        --k;
        return c;
    }
};

// Right-to-left walk over the domain-char run that ends at `e` (ext:537-689): label rules of is_valid_domain
// (ext:637-689) and the public-suffix test. With HASH the reverse PSL hash is kept and the device PSL table is probed at
// every dot until a suffix is found (find_valid_tld_suffix_bytes, ext:1671-1692: dots right-to-left, first hit wins);
// without it the caller already knows that the last label alone is a public suffix, so the first dot decides.
struct WalkInit {          // state of the walk after the bytes the caller has already consumed (none: the defaults)
    uint64_t rh;
    uint32_t cur = 0, last_c = 0, labels = 1;
    bool high = false, bad = false, found = false;
};
// State of a walk that was interrupted inside a long run (WALK_LONG): the wave finishes the bulk of the run together
// (coop_domain_skip) and the lane resumes from the updated state.
struct WalkState { uint32_t pos, cur, last_c, labels; bool high, bad, found; };
constexpr int WALK_NO = 0, WALK_YES = 1, WALK_LONG = 2;
constexpr uint32_t WALK_BUDGET_WORDS = 128;   // 8-byte words one lane walks alone before it asks the wave for help
template <bool HASH>
__device__ __forceinline__ int domain_walk_back(const LogView& lg, const DevDb& db, uint32_t min_labels, uint32_t e, BackReader& br,
                                                 const WalkInit& wi, uint32_t& start, WalkState* long_out, bool* defer_utf8 = nullptr,
                                                 bool* hit_start = nullptr) {
    uint64_t rh = wi.rh;
    bool found = wi.found, bad = wi.bad, high = wi.high;
    uint32_t labels = wi.labels, cur = wi.cur, last_c = wi.last_c;
    uint32_t first_c = 0x100;  // byte in front of the run (0x100 = buffer start)
    uint32_t words = 0;
    while (br.pos > 0) {
        // Long runs (hostile input: one lane walks the whole run): once the public-suffix question is settled no hash is
        // needed any more, and whole 8-byte words of domain characters are handled with the SWAR masks — the rules only
        // look at neighbouring bytes (dot / dash next to a dot) and count dots. The word that holds the start of the run
        // is left to the byte loop below.
        if ((!HASH || found) && br.pos >= 32 && (cur != 0 || last_c == '.')) {
            constexpr uint64_t H = 0x8080808080808080ull;
            uint32_t pos = br.pos;
            uint64_t w = BackReader::load8(lg.p + pos - 8);
            bool any = false;
            for (;;) {
                const uint64_t wn = pos >= 24 ? BackReader::load8(lg.p + pos - 16) : 0ull;   // next word, loaded early
                const ByteMasks m = domain_masks(w);
                if ((~m.dc & H) != 0) break;                                   // the run starts inside this word
                // right-hand neighbour of every byte moved onto it (byte 7's neighbour is the last byte consumed)
                const uint64_t dr = (m.dot >> 8) | (last_c == '.' ? (0x80ull << 56) : 0ull);
                const uint64_t sr = (m.dash >> 8) | (last_c == '-' ? (0x80ull << 56) : 0ull);
                if (((m.dot & (dr | sr)) | (m.dash & dr)) != 0) bad = true;   // empty label, label starting / ending with '-'
                if (m.dot != 0 && !bad) found = true;
                labels += (uint32_t)__popcll(m.dot);
                high |= m.high != 0;
                last_c = (uint32_t)w & 0xFF;
                cur = last_c == '.' ? 0u : 1u;
                pos -= 8;
                any = true;
                if (pos < 24) break;
                if (long_out && ++words >= WALK_BUDGET_WORDS) {   // a long run: hand the state to the wave
                    *long_out = WalkState{pos, cur, last_c, labels, high, bad, found};
                    return WALK_LONG;
                }
                w = wn;
            }
            if (any) { br.init(lg.p, pos); continue; }
        }
        uint32_t c = br.next();
        if (!d_is_domain_char_fast(c)) { first_c = c; ++br.pos; break; }
        if (HASH && !found && e - br.pos - 1 > db.max_suffix_len) return WALK_NO;   // longer than every public suffix: no TLD
        if (c == '.') {
            if (cur == 0 || last_c == '-') bad = true;
            if (HASH) { if (!found && !bad) found = psl_contains(db, psl_hash_finish(rh), lg.p + br.pos + 1, e - br.pos - 1); }
            else if (!bad) found = true;
            ++labels;
            cur = 0;
        } else {
            if (cur == 0 && c == '-') bad = true;  // label ends with '-'
            ++cur;
            high |= c >= 0x80;
        }
        if (HASH) rh = psl_hash_step(rh, (uint8_t)c);
        last_c = c;
    }
    const uint32_t s = br.pos;
    // the walk ran into the start of what `lg` shows: for a window view of the log (k_validate) the verdict below is not to be trusted
    if (hit_start && first_c == 0x100) *hit_start = true;
    if (cur == 0 || last_c == '-') bad = true;  // leftmost label empty / starts with '-'
    if (bad || !found || labels < min_labels) return WALK_NO;
    if (first_c != 0x100 && !d_is_boundary(first_c)) return WALK_NO;
    // a long run with bytes >= 0x80: the caller validates it with the whole wave (coop_valid_utf8)
    if (high && defer_utf8) *defer_utf8 = true;
    else if (high && !d_valid_utf8(lg.p + s, e - s)) return WALK_NO;
    start = s;
    return WALK_YES;
}

// core::str::from_utf8 acceptance of p[0, n) (wave-uniform arguments) by the whole wave, 64 bytes per step: byte classes
// become lane masks (ballots), a sequence is well formed iff the positions where continuation bytes are expected
// (1 / 2 / 3 after a 2- / 3- / 4-byte lead) are exactly the positions that hold one, plus the four restricted second bytes
// (E0: A0..BF, ED: 80..9F, F0: 90..BF, F4: 80..8F). Expectations that run past a step are carried into the next one.
__device__ __forceinline__ bool coop_valid_utf8(const uint8_t* p, uint32_t n) {
    const uint32_t lane = lane_id();
    uint64_t carry = 0;          // expected-continuation bits for the first positions of the next step
    uint32_t c2 = 0;             // restricted-second-byte kinds pending for position 0 of the next step: 1 E0, 2 ED, 4 F0, 8 F4
    bool ok = true;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t b = base + lane < n ? p[base + lane] : 0u;
        const uint64_t cont = __ballot((b & 0xC0u) == 0x80u);
        const uint64_t l2 = __ballot(b >= 0xC2u && b <= 0xDFu), l3 = __ballot((b & 0xF0u) == 0xE0u), l4 = __ballot(b >= 0xF0u && b <= 0xF4u);
        const uint64_t badb = __ballot(b == 0xC0u || b == 0xC1u || b >= 0xF5u);
        const uint64_t e0 = __ballot(b == 0xE0u), ed = __ballot(b == 0xEDu), f0 = __ballot(b == 0xF0u), f4 = __ballot(b == 0xF4u);
        const uint64_t ltA0 = __ballot(b < 0xA0u), gt9F = __ballot(b > 0x9Fu), lt90 = __ballot(b < 0x90u), gt8F = __ballot(b > 0x8Fu);
        const uint64_t lead = l2 | l3 | l4, lead34 = l3 | l4;
        const uint64_t exp = (lead << 1) | (lead34 << 2) | (l4 << 3) | carry;
        carry = (lead >> 63) | (lead34 >> 62) | (l4 >> 61);
        uint64_t bad2 = ((e0 << 1) & ltA0) | ((ed << 1) & gt9F) | ((f0 << 1) & lt90) | ((f4 << 1) & gt8F);
        if (c2 & 1) bad2 |= ltA0 & 1; if (c2 & 2) bad2 |= gt9F & 1; if (c2 & 4) bad2 |= lt90 & 1; if (c2 & 8) bad2 |= gt8F & 1;
        c2 = (uint32_t)(e0 >> 63) | ((uint32_t)(ed >> 63) << 1) | ((uint32_t)(f0 >> 63) << 2) | ((uint32_t)(f4 >> 63) << 3);
        if ((exp ^ cont) | badb | bad2) { ok = false; break; }
    }
    return ok && carry == 0;
}

// The bulk of a long domain-character run, walked by the whole wave: 512 bytes per step (lane i takes the 8 bytes at
// pos - 512 + 8 i), the same per-word rules as the lane's own word loop above, combined with ballots. All arguments are
// wave-uniform (the state of ONE lane's walk, broadcast by the caller). Stops in front of the 512-byte step that holds the
// start of the run (or near the start of the buffer); the lane finishes from there. One lane alone needs a dependent load
// per 8 bytes — a multi-megabyte token cost tens of milliseconds.
__device__ __forceinline__ void coop_domain_skip(const LogView& lg, WalkState& st) {
    constexpr uint64_t H = 0x8080808080808080ull;
    const uint32_t lane = lane_id();
    while (st.pos >= 512 + 32) {
        const uint32_t base = st.pos - 512;
        const uint64_t w = BackReader::load8(lg.p + base + 8 * lane);
        const ByteMasks m = domain_masks(w);
        const uint64_t stop = __ballot((~m.dc & H) != 0);
        const int hi = stop ? 63 - (int)__clzll((long long)stop) : -1;   // the run starts in this lane's word (the walk consumes lanes 63 down to hi + 1)
        const bool valid = (int)lane > hi;
        // right-hand neighbour of the word's top byte: the lowest byte of the next lane's word; lane 63: the last byte consumed
        uint32_t nb = (uint32_t)__shfl_down((int)((uint32_t)w & 0xFFu), 1);
        if (lane == 63) nb = st.last_c;
        const uint64_t dr = (m.dot >> 8) | (nb == '.' ? (0x80ull << 56) : 0ull);
        const uint64_t sr = (m.dash >> 8) | (nb == '-' ? (0x80ull << 56) : 0ull);
        const bool badl = ((m.dot & (dr | sr)) | (m.dash & dr)) != 0;   // empty label, label starting / ending with '-'
        if (__ballot(valid && badl)) st.bad = true;
        if (__ballot(valid && m.dot != 0) && !st.bad) st.found = true;
        if (__ballot(valid && m.high != 0)) st.high = true;
        uint32_t nd = valid ? (uint32_t)__popcll(m.dot) : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nd += (uint32_t)__shfl_xor((int)nd, off);
        st.labels += nd;
        const uint32_t low = (uint32_t)w & 0xFFu;
        if (hi >= 0) {
            if (hi < 63) {
                st.pos = base + 8 * (uint32_t)(hi + 1);
                st.last_c = (uint32_t)__shfl((int)low, hi + 1);
                st.cur = st.last_c == '.' ? 0u : 1u;
            }
            return;
        }
        st.pos = base;
        st.last_c = (uint32_t)__shfl((int)low, 0);
        st.cur = st.last_c == '.' ? 0u : 1u;
    }
}

// Domain (ext:537-689). `j` is the first byte after a dot (anchor: label-char, '.', label-char). Only the LAST dot
// of a maximal domain-char run owns the run; it validates the run as a whole. `tldtab` is the LDS copy of
// DevDb::tld_tab (exact table of the last labels of <= 7 bytes), `bloom` covers the longer ones.
struct DomLong { WalkState st; uint32_t e; bool alone; };
// `edge` (optional) is set when the scan touched either end of what `lg` shows (the verdict then depends on bytes outside a window view).
__device__ int val_domain(const LogView& lg, const DevDb& db, const uint32_t* bloom, const uint2* tldtab, uint32_t min_labels,
                          uint32_t j, uint32_t& start, uint32_t& end, DomLong* long_out, bool* edge = nullptr) {
    uint32_t p = j, th = 2166136261u;
    bool open = true;  // last label not yet terminated
    uint32_t stop_c = 0x100;  // byte that ended the run on the right (0x100 = buffer end)
    uint2 w = make_uint2(0u, 0u);
    const bool wide = j + 8 <= lg.len;
    if (wide) {
        // the first 8 bytes of the last label in one load; most labels (com, net, css, html, ...) end inside it
        __builtin_memcpy(&w, lg.p + j, 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (open) {
                const uint32_t c = ((k < 4 ? w.x : w.y) >> (8 * (k & 3))) & 0xFF;
                if (!d_is_domain_char_fast(c)) { open = false; stop_c = c; }
                else {
                    if (c == '.') return false;  // a later dot owns this run
                    th = tld_hash_step(th, c);
                    ++p;
                }
            }
        }
    }
    while (open && p < lg.len) {
        uint32_t c = lg.at(p);
        if (!d_is_domain_char_fast(c)) { stop_c = c; break; }
        if (c == '.') return false;  // a later dot owns this run
        if (p - j >= db.max_tld_len) return false;   // longer than every last label of the list (bounds the scan on hostile input)
        th = tld_hash_step(th, c);
        ++p;
    }
    const uint32_t e = p, ll = e - j;
    if (edge && stop_c == 0x100) *edge = true;
    if (ll > db.max_tld_len) return false;
    if (stop_c != 0x100 && !d_is_boundary(stop_c)) return false;  // boundary (or buffer end) after the run (ext:600-606)
    bool alone = false;  // the last label by itself is a public suffix: the first dot of the walk decides
    if (wide && ll <= 7) {
        // exact: a last label that no suffix ends with can never pass the PSL test
        const uint32_t lo = ll >= 4 ? w.x : (w.x & ((1u << (8 * ll)) - 1u));
        const uint32_t hi = ll > 4 ? (w.y & ((1u << (8 * (ll - 4))) - 1u)) : 0u;
        uint32_t slot = tld_tab_slot(lo, hi);
        for (;;) {
            const uint2 t = tldtab[slot];
            if ((t.y >> 24) == 0) return false;
            if (t.x == lo && (t.y & 0xFFFFFFu) == hi) { alone = (t.y >> 24) & 1; break; }
            slot = (slot + 1) & ((1u << TLD_TAB_BITS) - 1);
        }
    } else {
        const uint32_t bit = tld_hash_bit(th);
        if (!((bloom[bit >> 5] >> (bit & 31)) & 1)) return false;  // last label is no suffix's last label -> no PSL hit possible
    }
    uint32_t s;
    BackReader br;
    br.init(lg.p, e);
    WalkInit wi;
    wi.rh = psl_hash_init();
    WalkState ws{};
    const int r = alone ? domain_walk_back<false>(lg, db, min_labels, e, br, wi, s, long_out ? &ws : nullptr, nullptr, edge)
                        : domain_walk_back<true>(lg, db, min_labels, e, br, wi, s, long_out ? &ws : nullptr, nullptr, edge);
    if (r == WALK_LONG) { long_out->st = ws; long_out->e = e; long_out->alone = alone; return WALK_LONG; }
    if (r != WALK_YES) return WALK_NO;
    start = s; end = e;
    return WALK_YES;
}
// the rest of a walk that coop_domain_skip has brought near the start of the run
__device__ bool val_domain_resume(const LogView& lg, const DevDb& db, uint32_t min_labels, const DomLong& dl, uint32_t& start, uint32_t& end, bool& need_utf8) {
    BackReader br;
    br.init(lg.p, dl.st.pos);
    WalkInit wi;
    wi.rh = psl_hash_init();   // the public-suffix question is settled by the time a walk reports a long run
    wi.cur = dl.st.cur; wi.last_c = dl.st.last_c; wi.labels = dl.st.labels; wi.high = dl.st.high; wi.bad = dl.st.bad; wi.found = dl.st.found;
    uint32_t s;
    // HASH only matters while `found` is open; it is closed here (word mode is entered only then)
    if (domain_walk_back<false>(lg, db, min_labels, dl.e, br, wi, s, nullptr, &need_utf8) != WALK_YES) return false;
    start = s; end = dl.e;
    return true;
}

// The same for an anchor that comes with its context record (k_anchor copies log[j-24, j+8) from its LDS window):
// w = log[j, j+8), b0 = log[j-8, j), b1 = log[j-16, j-8), b2 = log[j-24, j-16). The common case — last label of <= 7
// bytes that alone is a public suffix, name no longer than the context — is decided with mask arithmetic on these
// registers, without a per-byte loop and without divergent control flow (k_validate is bound by scalar-ALU issue, i.e.
// by control flow): the rules of is_valid_domain (ext:637-689) become tests on the dot / dash / domain-char masks.
// Everything else is left to the general path (val_domain, in k_validate).
// Returns 0 (no domain), 1 (domain: start / end set, the whole name lies in the context) or 2 (undecided: general path).
__device__ __forceinline__ int val_domain_pre(const uint2* tldtab, uint32_t min_labels, uint32_t j, uint2 w, uint64_t b0, uint64_t b1,
                                              uint64_t b2, uint32_t& start, uint32_t& end) {
    constexpr uint64_t H = 0x8080808080808080ull;
    const uint64_t w64 = (uint64_t)w.x | ((uint64_t)w.y << 32);
    const ByteMasks mw = domain_masks(w64);
    // last label = leading domain-char bytes of w; ll = its length (8: not terminated within the window)
    const uint64_t ndc = ~mw.dc & H;
    const uint32_t ll = ndc ? (uint32_t)(__ffsll((long long)ndc) - 1) >> 3 : 8u;
    if (ll <= 7 && ll >= 1) {
        const uint64_t below = (1ull << (8 * ll)) - 1ull;            // the label's bytes
        if (mw.dot & below) return 0;                                // a later dot owns this run
        const uint32_t stop_c = (uint32_t)(w64 >> (8 * ll)) & 0xFF;
        if (!d_is_boundary(stop_c)) return 0;
        const uint32_t lo = (uint32_t)(w64 & below), hi = (uint32_t)((w64 & below) >> 32);
        uint32_t slot = tld_tab_slot(lo, hi);
        bool alone = false;
        for (;;) {
            const uint2 t = tldtab[slot];
            if ((t.y >> 24) == 0) return 0;
            if (t.x == lo && (t.y & 0xFFFFFFu) == hi) { alone = (t.y >> 24) & 1; break; }
            slot = (slot + 1) & ((1u << TLD_TAB_BITS) - 1);
        }
        if (alone) {
            // the 24 bytes in front of the label, address order: b2 | b1 | b0; the run extends leftwards from the top
            // byte of b0 (the dot at j-1) while bytes are domain chars
            const ByteMasks m0 = domain_masks(b0), m1 = domain_masks(b1), m2 = domain_masks(b2);
            const uint64_t n0m = ~m0.dc & H, n1m = ~m1.dc & H, n2m = ~m2.dc & H;
            const uint32_t c0 = n0m ? (uint32_t)__clzll((long long)n0m) >> 3 : 8u;   // domain chars at the top of b0
            const uint32_t c1 = n1m ? (uint32_t)__clzll((long long)n1m) >> 3 : 8u, c2 = n2m ? (uint32_t)__clzll((long long)n2m) >> 3 : 8u;
            const uint32_t consumed = c0 < 8 ? c0 : (c1 < 8 ? 8 + c1 : 16 + c2);
            if (consumed < 24 || j == 24) {   // the run starts inside the context (or at the start of the buffer)
                // region masks: the top `k` bytes of each word that belong to the run
                auto top = [](uint32_t k) -> uint64_t { return k == 0 ? 0ull : (~0ull << (64 - 8 * k)); };
                const uint64_t r0 = top(min(consumed, 8u)), r1 = top(consumed > 8 ? min(consumed - 8, 8u) : 0u),
                               r2 = top(consumed > 16 ? consumed - 16 : 0u);
                // "byte to the right" (one position closer to the label) moved onto each byte: shift towards lower
                // addresses; the byte right of b0's top byte is the label's first byte (neither dot nor dash)
                const uint64_t d0 = m0.dot >> 8, d1 = (m1.dot >> 8) | (m0.dot << 56), d2 = (m2.dot >> 8) | (m1.dot << 56);
                const uint64_t s0 = m0.dash >> 8, s1 = (m1.dash >> 8) | (m0.dash << 56), s2 = (m2.dash >> 8) | (m1.dash << 56);
                // bad events: empty label (two dots in a row), label starting with '-' (dot, then dash to its right),
                // label ending with '-' (dash, then dot to its right)
                const uint64_t bad = ((m0.dot & (d0 | s0)) | (m0.dash & d0)) & r0;
                const uint64_t bad1 = ((m1.dot & (d1 | s1)) | (m1.dash & d1)) & r1, bad2 = ((m2.dot & (d2 | s2)) | (m2.dash & d2)) & r2;
                const uint32_t ndots = (uint32_t)(__popcll(m0.dot & r0) + __popcll(m1.dot & r1) + __popcll(m2.dot & r2));
                // leftmost byte of the run: must not be a dot (empty leftmost label) or a dash
                const uint32_t li = 24 - consumed;                       // its index in address order (consumed >= 1: the dot)
                const uint64_t lw = li < 8 ? b2 : li < 16 ? b1 : b0;
                const uint32_t left_c = (uint32_t)(lw >> (8 * (li & 7))) & 0xFF;
                const uint32_t fi = 23 - consumed;                       // byte in front of the run (if inside the context)
                const uint64_t fw = fi < 8 ? b2 : fi < 16 ? b1 : b0;
                const uint32_t first_c = consumed < 24 ? (uint32_t)(fw >> (8 * (fi & 7))) & 0xFF : 0x100u;
                const uint32_t lastc = (uint32_t)(w64 >> (8 * (ll - 1))) & 0xFF;
                const bool any_bad = (bad | bad1 | bad2) != 0 || lastc == '-' || consumed == 0 || left_c == '.' || left_c == '-';
                if (any_bad || ndots == 0 || 1 + ndots < min_labels) return 0;
                if (first_c != 0x100 && !d_is_boundary(first_c)) return 0;
                const uint32_t s_pos = j - consumed;
                const bool high = ((m0.high & r0) | (m1.high & r1) | (m2.high & r2) | (mw.high & below)) != 0;
                if (high) return 2;   // needs the UTF-8 check of the general path
                start = s_pos; end = j + ll;
                return 1;
            }
            // the name reaches further back than the context: general path below
        }
    }
    return 2;
}

// val_domain_pre over NB context words instead of three, for the anchors k_validate_dom could not decide from its 24 context bytes (a proxy
// log: 2 M host names of 25..48 bytes in front of their last label per batch, which otherwise all take the byte-and-word loops of the general
// walk): w = log[j, j+8), b[k] = log[j-8(k+1), j-8k). The same rules on the same masks, written as loops over the words that unroll. A name
// decided here has at most 8 NB + 7 <= 63 bytes, so the 63-byte label and 253-byte name limits of is_valid_domain (ext:637-689) cannot bite
// (NB <= 7). Returns 0 (no domain), 1 (domain: start / end set) or 2 (undecided: the general walk).
template <int NB>
__device__ __forceinline__ int val_domain_ctx(const uint2* tldtab, uint32_t min_labels, uint32_t j, uint2 w, const uint64_t (&b)[NB],
                                              uint32_t& start, uint32_t& end) {
    static_assert(NB >= 1 && NB <= 7, "a label must not reach 64 bytes inside the context");
    constexpr uint64_t H = 0x8080808080808080ull;
    const uint64_t w64 = (uint64_t)w.x | ((uint64_t)w.y << 32);
    const ByteMasks mw = domain_masks(w64);
    const uint64_t ndc = ~mw.dc & H;
    const uint32_t ll = ndc ? (uint32_t)(__ffsll((long long)ndc) - 1) >> 3 : 8u;
    if (ll > 7 || ll < 1) return 2;
    const uint64_t below = (1ull << (8 * ll)) - 1ull;
    if (mw.dot & below) return 0;                                    // a later dot owns this run
    const uint32_t stop_c = (uint32_t)(w64 >> (8 * ll)) & 0xFF;
    if (!d_is_boundary(stop_c)) return 0;
    const uint32_t lo = (uint32_t)(w64 & below), hi = (uint32_t)((w64 & below) >> 32);
    uint32_t slot = tld_tab_slot(lo, hi);
    bool alone = false;
    for (;;) {
        const uint2 t = tldtab[slot];
        if ((t.y >> 24) == 0) return 0;
        if (t.x == lo && (t.y & 0xFFFFFFu) == hi) { alone = (t.y >> 24) & 1; break; }
        slot = (slot + 1) & ((1u << TLD_TAB_BITS) - 1);
    }
    if (!alone) return 2;
    ByteMasks m[NB];
    uint32_t consumed = 0;
    bool open = true;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        m[k] = domain_masks(b[k]);
        const uint64_t nm = ~m[k].dc & H;
        const uint32_t ck = nm ? (uint32_t)__clzll((long long)nm) >> 3 : 8u;   // domain chars at the top of word k
        if (open) { consumed += ck; open = ck == 8; }
    }
    if (consumed == 8 * NB && j != 8 * NB) return 2;                 // the name reaches further back than the context
    uint64_t bad = 0, high = mw.high & below;
    uint32_t ndots = 0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const uint32_t in_k = consumed > 8u * k ? min(consumed - 8u * k, 8u) : 0u;     // bytes of word k that belong to the run (its top ones)
        const uint64_t r = in_k == 0 ? 0ull : (~0ull << (64 - 8 * in_k));
        // the byte to the right (one position closer to the label) moved onto each byte; right of b[0]'s top byte stands the label's first byte
        const uint64_t d = (m[k].dot >> 8) | (k ? (m[k ? k - 1 : 0].dot << 56) : 0ull);
        const uint64_t sd = (m[k].dash >> 8) | (k ? (m[k ? k - 1 : 0].dash << 56) : 0ull);
        bad |= ((m[k].dot & (d | sd)) | (m[k].dash & d)) & r;        // empty label, label starting with '-', label ending with '-'
        ndots += (uint32_t)__popcll(m[k].dot & r);
        high |= m[k].high & r;
    }
    // leftmost byte of the run and the byte in front of it, by their index in address order (word NB-1 holds the lowest addresses)
    const uint32_t li = 8 * NB - consumed, fi = li - 1;
    uint32_t left_c = 0, first_c = 0x100u;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const uint32_t wi = (uint32_t)(NB - 1 - k);                   // index in address order of word k
        if ((li >> 3) == wi) left_c = (uint32_t)(b[k] >> (8 * (li & 7))) & 0xFF;
        if (consumed < 8 * NB && (fi >> 3) == wi) first_c = (uint32_t)(b[k] >> (8 * (fi & 7))) & 0xFF;
    }
    const uint32_t lastc = (uint32_t)(w64 >> (8 * (ll - 1))) & 0xFF;
    const bool any_bad = bad != 0 || lastc == '-' || consumed == 0 || left_c == '.' || left_c == '-';
    if (any_bad || ndots == 0 || 1 + ndots < min_labels) return 0;
    if (first_c != 0x100u && !d_is_boundary(first_c)) return 0;
    if (high) return 2;                                              // needs the UTF-8 check of the general path
    start = j - consumed; end = j + ll;
    return 1;
}

__device__ bool all_hex(const LogView& lg, uint32_t s, uint32_t n) {
    for (uint32_t k = 0; k < n; ++k) if (!d_is_hex(lg.at(s + k))) return false;
    return true;
}
// all four bytes of x are ASCII hex digits (SWAR; no carries cross bytes because every addend keeps bytes below 0x100)
__device__ __forceinline__ bool hex4(uint32_t x) {
    const uint32_t t = x & 0x7F7F7F7Fu, l = t | 0x20202020u;
    const uint32_t dig = (t + 0x50505050u) & ~(t + 0x46464646u);   // >= '0' and not >= ':'
    const uint32_t alp = (l + 0x1F1F1F1Fu) & ~(l + 0x19191919u);   // >= 'a' and not >= 'g' (case folded)
    return (((dig | alp) & ~x) & 0x80808080u) == 0x80808080u;
}
// all_hex for the hash lengths (multiples of 8): independent 8-byte loads instead of a dependent byte-load chain
__device__ __forceinline__ bool all_hex_wide(const uint8_t* p, uint32_t n) {
    bool ok = true;
#pragma unroll 4
    for (uint32_t k = 0; k < n; k += 8) {
        uint2 v;
        __builtin_memcpy(&v, p + k, 8);
        ok = ok && hex4(v.x) && hex4(v.y);
    }
    return ok;
}


// ------------------------------------------------------------------------------------------------ rare validators
__device__ __forceinline__ uint32_t d_lower(uint32_t c) { return (c - 'A' < 26u) ? c + 32 : c; }

// IPv6 (ext:1044-1116): `p2` is the index of the second ':' of a "::" that is not preceded by a third ':'.
__device__ bool val_ipv6(const LogView& lg, uint32_t p2, uint32_t& start, uint32_t& end) {
    uint32_t s = p2 - 1;
    while (s > 0) { uint32_t c = lg.at(s - 1); if (!d_is_hex(c) && c != ':') break; --s; }
    uint32_t e = p2 + 1;
    while (e < lg.len) { uint32_t c = lg.at(e); if (!d_is_hex(c) && c != ':') break; ++e; }
    uint32_t n = e - s;
    if (n < 8 || n > 64) return false;  // > 39 can never parse; 64 bounds the local work
    const uint8_t* c = lg.p + s;
    if ((c[0] == ':' && c[1] == ':') || (c[n - 2] == ':' && c[n - 1] == ':')) return false;
    // fe80::/10 text prefix filter (ext:1425-1456)
    if (d_lower(c[0]) == 'f' && d_lower(c[1]) == 'e') { uint32_t x = d_lower(c[2]); if (x == '8' || x == '9' || x == 'a' || x == 'b') return false; }
    uint16_t seg[8];
    if (!d_parse_ipv6(c, n, seg)) return false;
    start = s; end = e;
    return true;
}

// The same on an 80-byte window win[0..80) = log[p2-40, p2+40) held in LDS (k_validate): a run that reaches either
// window edge is longer than 39 bytes and can never parse, so the window is exact.
__device__ bool val_ipv6_win(const uint8_t* win, uint32_t p2, uint32_t& start, uint32_t& end) {
    uint32_t s = 39;  // index of the first ':' of the "::"
    while (s > 0) { uint32_t c = win[s - 1]; if (!d_is_hex(c) && c != ':') break; --s; }
    if (s == 0) return false;
    uint32_t e = 41;
    while (e < 80) { uint32_t c = win[e]; if (!d_is_hex(c) && c != ':') break; ++e; }
    if (e == 80) return false;
    const uint32_t n = e - s;
    if (n < 8 || n > 64) return false;
    const uint8_t* c = win + s;
    if ((c[0] == ':' && c[1] == ':') || (c[n - 2] == ':' && c[n - 1] == ':')) return false;
    if (d_lower(c[0]) == 'f' && d_lower(c[1]) == 'e') { uint32_t x = d_lower(c[2]); if (x == '8' || x == '9' || x == 'a' || x == 'b') return false; }
    uint16_t seg[8];
    if (!d_parse_ipv6(c, n, seg)) return false;
    start = p2 - 40 + s; end = p2 - 40 + e;
    return true;
}

// E-mail (ext:891-950, 1182-1196). Both scans go a word at a time over long runs (a multi-megabyte local part is one
// lane's work): the rules need the class of each byte, "two dots in a row" and "any letter", all of which the SWAR masks give.
struct LocalMasks { uint64_t loc, dot, alp; };
struct LocalMasks32 { uint32_t loc, dot, alp; };
__device__ __forceinline__ LocalMasks32 email_local_masks32(uint32_t x) {   // 32-bit halves: see domain_masks32
    constexpr uint32_t H = 0x80808080u, L7 = 0x7F7F7F7Fu;
    const uint32_t t = x & L7, l = t | 0x20202020u;
    const uint32_t dig = (t + 0x50505050u) & ~(t + 0x46464646u);
    const uint32_t alp = (l + 0x1F1F1F1Fu) & ~(l + 0x05050505u);
    const uint32_t ndot = (t ^ 0x2E2E2E2Eu) + L7, ndash = (t ^ 0x2D2D2D2Du) + L7;
    const uint32_t nus = (t ^ 0x5F5F5F5Fu) + L7, npl = (t ^ 0x2B2B2B2Bu) + L7;
    LocalMasks32 m;
    m.dot = ~ndot & ~x & H;
    m.alp = alp & ~x & H;
    m.loc = (dig | alp | ~ndot | ~ndash | ~nus | ~npl) & ~x & H;   // is_email_local_char (ext:1644)
    return m;
}
__device__ __forceinline__ LocalMasks email_local_masks(uint64_t x) {
    const LocalMasks32 lo = email_local_masks32((uint32_t)x), hi = email_local_masks32((uint32_t)(x >> 32));
    LocalMasks m;
    m.loc = (uint64_t)lo.loc | ((uint64_t)hi.loc << 32);
    m.dot = (uint64_t)lo.dot | ((uint64_t)hi.dot << 32);
    m.alp = (uint64_t)lo.alp | ((uint64_t)hi.alp << 32);
    return m;
}
// State of the leftward scan over the local part: s = its current start, prev = the last byte consumed.
struct EmailState { uint32_t s, prev; bool has_letter, dotdot; };
// Scans the local part leftwards from `at`. With `budget` the word loop gives up after WALK_BUDGET_WORDS words and returns true
// ("long run": the wave continues with coop_email_skip, then the lane calls this again without budget to finish).
__device__ __forceinline__ bool email_local_scan(const LogView& lg, EmailState& es, bool budget) {
    constexpr uint64_t H = 0x8080808080808080ull;
    uint32_t s = es.s, prev = es.prev, words = 0;
    bool has_letter = es.has_letter, dotdot = es.dotdot, is_long = false;
    if (s >= 32) {
        uint64_t w = BackReader::load8(lg.p + s - 8);
        for (;;) {
            const uint64_t wn = BackReader::load8(lg.p + s - 16);
            const LocalMasks m = email_local_masks(w);
            if ((~m.loc & H) != 0) break;
            const uint64_t dr = (m.dot >> 8) | (prev == '.' ? (0x80ull << 56) : 0ull);
            dotdot |= (m.dot & dr) != 0;
            has_letter |= m.alp != 0;
            prev = (uint32_t)w & 0xFF;
            s -= 8;
            if (s < 32) break;
            if (budget && ++words >= WALK_BUDGET_WORDS) { is_long = true; break; }
            w = wn;
        }
    }
    if (!is_long) {
        while (s > 0) {
            const uint32_t c = lg.at(s - 1);
            if (!d_is_email_local(c)) break;
            dotdot |= c == '.' && prev == '.';
            has_letter |= d_is_alpha(c);
            prev = c;
            --s;
        }
    }
    es = EmailState{s, prev, has_letter, dotdot};
    return is_long;
}
// the whole wave on one lane's long local part: 512 bytes per step, same rules as the word loop above (wave-uniform state)
__device__ __forceinline__ void coop_email_skip(const LogView& lg, EmailState& es) {
    constexpr uint64_t H = 0x8080808080808080ull;
    const uint32_t lane = lane_id();
    while (es.s >= 512 + 32) {
        const uint32_t base = es.s - 512;
        const uint64_t w = BackReader::load8(lg.p + base + 8 * lane);
        const LocalMasks m = email_local_masks(w);
        const uint64_t stop = __ballot((~m.loc & H) != 0);
        const int hi = stop ? 63 - (int)__clzll((long long)stop) : -1;
        const bool valid = (int)lane > hi;
        uint32_t nb = (uint32_t)__shfl_down((int)((uint32_t)w & 0xFFu), 1);
        if (lane == 63) nb = es.prev;
        const uint64_t dr = (m.dot >> 8) | (nb == '.' ? (0x80ull << 56) : 0ull);
        if (__ballot(valid && (m.dot & dr) != 0)) es.dotdot = true;
        if (__ballot(valid && m.alp != 0)) es.has_letter = true;
        const uint32_t low = (uint32_t)w & 0xFFu;
        if (hi >= 0) {
            if (hi < 63) { es.s = base + 8 * (uint32_t)(hi + 1); es.prev = (uint32_t)__shfl((int)low, hi + 1); }
            return;
        }
        es.s = base;
        es.prev = (uint32_t)__shfl((int)low, 0);
    }
}
// everything after the local part: boundary in front of it, the domain part, the public-suffix test
__device__ bool val_email_finish(const LogView& lg, const DevDb& db, uint32_t at, const EmailState& es, uint32_t& start, uint32_t& end,
                                 bool* edge = nullptr) {
    constexpr uint64_t H = 0x8080808080808080ull;
    const uint32_t s = es.s;
    if (s == at) return false;
    if (s > 0 && !d_is_boundary(lg.at(s - 1))) return false;
    uint32_t e = at + 1;
    bool has_dot = false;
    if (e + 32 <= lg.len) {
        uint64_t w = BackReader::load8(lg.p + e);
        for (;;) {
            const uint64_t wn = BackReader::load8(lg.p + e + 8);
            const ByteMasks m = domain_masks(w);
            if ((~(m.dc & ~m.high) & H) != 0) break;                  // is_domain_char (ext:1639): ASCII only
            has_dot |= m.dot != 0;
            e += 8;
            if (e + 32 > lg.len) break;
            w = wn;
        }
    }
    while (e < lg.len) {
        const uint32_t c = lg.at(e);
        if (!d_is_domain_char(c)) break;
        has_dot |= c == '.';
        ++e;
    }
    if (edge && (e >= lg.len || s == 0)) *edge = true;   // the scans touched an end of what `lg` shows
    if (e == at + 1) return false;
    if (e < lg.len && !d_is_boundary(lg.at(e))) return false;
    if (es.dotdot || !es.has_letter || !has_dot) return false;
    if (!psl_suffix_exists(db, lg.p, at + 1, e)) return false;
    start = s; end = e;  // all bytes are ASCII: from_utf8 always succeeds
    return true;
}

// extract_email_at (ext:891-950) on context words in registers instead of the byte and word loops above, for an '@' in the middle of a line (an
// application log has one in every fourth line: 1.9 M per batch, each a chain of a dozen dependent loads through the loops): Lw[k] =
// log[at-8(k+1), at-8k) — the local part grows leftwards from the top byte of Lw[0] —, Rw[k] = log[at+1+8k, at+9+8k) — the domain part
// grows rightwards from the low byte of Rw[0]. The rules are conjunctive, so their order does not matter: local part not empty, a boundary in
// front of it, no two dots in a row, a letter; domain part (ASCII domain characters) not empty, a boundary behind it, a dot in it, a public
// suffix at its end. The suffix test takes the shortcut of val_domain: a last label of <= 7 bytes is looked up in the exact table of last labels
// (absent: no suffix can end the name; alone a suffix: done); everything else asks psl_suffix_exists as before.
// Returns 0 (no address), 1 (address: start / end set) or 2 (a part is longer than its context: the loops decide).
template <int NL, int NR>
__device__ __forceinline__ int val_email_ctx(const uint2* tldtab, const DevDb& db, const LogView& lg, uint32_t at, const uint64_t (&Lw)[NL],
                                             const uint64_t (&Rw)[NR], uint32_t& start, uint32_t& end) {
    constexpr uint64_t H = 0x8080808080808080ull;
    uint32_t nl = 0, nr = 0;
    bool open = true;
    LocalMasks lm[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        lm[k] = email_local_masks(Lw[k]);
        const uint64_t nm = ~lm[k].loc & H;
        const uint32_t ck = nm ? (uint32_t)__clzll((long long)nm) >> 3 : 8u;   // local-part characters at the top of word k
        if (open) { nl += ck; open = ck == 8; }
    }
    if (nl == 8 * NL) return 2;
    ByteMasks rm[NR];
    open = true;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        rm[k] = domain_masks(Rw[k]);
        const uint64_t nm = ~(rm[k].dc & ~rm[k].high) & H;                     // is_domain_char (ext:1639): ASCII only
        const uint32_t ck = nm ? (uint32_t)(__ffsll((long long)nm) - 1) >> 3 : 8u;   // domain characters at the bottom of word k
        if (open) { nr += ck; open = ck == 8; }
    }
    if (nr == 8 * NR) return 2;
    if (nl == 0 || nr == 0) return 0;
    bool dotdot = false, has_letter = false, has_dot = false;
    uint32_t first_c = 0, stop_c = 0, last_dot = 0;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const uint32_t in_k = nl > 8u * k ? min(nl - 8u * k, 8u) : 0u;          // bytes of word k that belong to the local part (its top ones)
        const uint64_t r = in_k == 0 ? 0ull : (~0ull << (64 - 8 * in_k));
        const uint64_t d = (lm[k].dot >> 8) | (k ? (lm[k ? k - 1 : 0].dot << 56) : 0ull);   // the byte to the right is a dot (right of Lw[0]'s top byte stands the '@')
        dotdot |= (lm[k].dot & d & r) != 0;
        has_letter |= (lm[k].alp & r) != 0;
        const uint32_t fi = 8 * NL - 1 - nl;                                      // byte in front of the local part, index in address order
        if ((fi >> 3) == (uint32_t)(NL - 1 - k)) first_c = (uint32_t)(Lw[k] >> (8 * (fi & 7))) & 0xFF;
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const uint32_t in_k = nr > 8u * k ? min(nr - 8u * k, 8u) : 0u;          // bytes of word k that belong to the domain part (its low ones)
        const uint64_t r = in_k == 8 ? ~0ull : ((1ull << (8 * in_k)) - 1ull);
        const uint64_t dots = rm[k].dot & r;
        if (dots) { has_dot = true; last_dot = 8u * k + ((63u - (uint32_t)__clzll((long long)dots)) >> 3); }
        if ((nr >> 3) == (uint32_t)k) stop_c = (uint32_t)(Rw[k] >> (8 * (nr & 7))) & 0xFF;
    }
    if (!d_is_boundary(first_c) || !d_is_boundary(stop_c)) return 0;
    if (dotdot || !has_letter || !has_dot) return 0;
    const uint32_t e = at + 1 + nr, ll = nr - 1 - last_dot;
    bool suffix = false, decided = false;
    if (ll >= 1 && ll <= 7) {
        uint2 w;
        __builtin_memcpy(&w, lg.p + at + 2 + last_dot, 8);                        // the last label (the caller keeps 8 bytes behind the context readable)
        const uint32_t lo = ll >= 4 ? w.x : (w.x & ((1u << (8 * ll)) - 1u));
        const uint32_t hi = ll > 4 ? (w.y & ((1u << (8 * (ll - 4))) - 1u)) : 0u;
        uint32_t slot = tld_tab_slot(lo, hi);
        for (;;) {
            const uint2 t = tldtab[slot];
            if ((t.y >> 24) == 0) { decided = true; break; }                      // no suffix ends with this label
            if (t.x == lo && (t.y & 0xFFFFFFu) == hi) { if ((t.y >> 24) & 1) { decided = true; suffix = true; } break; }
            slot = (slot + 1) & ((1u << TLD_TAB_BITS) - 1);
        }
    }
    if (!decided) suffix = psl_suffix_exists(db, lg.p, at + 1, e);
    if (!suffix) return 0;
    start = at - nl; end = e;
    return 1;
}

// ---- SHA-256, Keccak-f[1600], base58, bech32 for the checksum validators of k_rare. All state lives in registers
// (fully unrolled rounds, static indices); byte strings (the token and the decoded address) live in a per-lane LDS
// scratch, so nothing goes to scratch memory.
struct Sha256 {
    uint32_t h[8];
    __device__ __forceinline__ void init() {
        h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a; h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
    }
    __device__ __forceinline__ static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    // one 64-byte block given as 16 big-endian words
    __device__ void block(uint32_t (&w)[16]) {
        constexpr uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
            0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
            0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
            0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
            0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
            0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (i >= 16) {  // rolling 16-word message schedule
                const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
                const uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3), s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
                w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
            }
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i & 15];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    // hash of data[0, len) (len < 120: at most two blocks); data may be in LDS
    __device__ void hash(const uint8_t* data, uint32_t len) {
        init();
        const uint32_t total = ((len + 9 + 63) / 64) * 64;
        for (uint32_t off = 0; off < total; off += 64) {
            uint32_t w[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                uint32_t x = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const uint32_t q = off + 4 * i + bb;
                    uint32_t v = 0;
                    if (q < len) v = data[q];
                    else if (q == len) v = 0x80;
                    else if (q >= total - 4) v = ((len * 8) >> (8 * (total - 1 - q))) & 0xFF;
                    x = (x << 8) | v;
                }
                w[i] = x;
            }
            block(w);
        }
    }
};

// Keccak-256 with the original 0x01 padding (tiny-keccak Keccak::v256), input <= 135 bytes (one block); out = the first
// 4 bytes of the digest for the Monero check, or all 32 for EIP-55.
__device__ void d_keccak256_1blk(const uint8_t* data, uint32_t len, uint64_t (&out)[4]) {
    constexpr uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
                                 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
                                 0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
                                 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    constexpr int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    constexpr int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    uint64_t st[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) st[i] = 0;
#pragma unroll
    for (int i = 0; i < 17; ++i) {  // rate = 136 bytes = 17 lanes
        uint64_t x = 0;
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) {
            const uint32_t q = 8 * i + bb;
            uint32_t v = q < len ? data[q] : 0;
            if (q == len) v ^= 0x01;
            if (q == 135) v ^= 0x80;
            x |= (uint64_t)v << (8 * bb);
        }
        st[i] = x;
    }
#pragma unroll 1
    for (int round = 0; round < 24; ++round) {
        uint64_t bc[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const uint64_t t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
#pragma unroll
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        uint64_t t = st[1];
#pragma unroll
        for (int i = 0; i < 24; ++i) { const int j = PILN[i]; const uint64_t b = st[j]; st[j] = rotl64(t, ROTC[i]); t = b; }
#pragma unroll
        for (int j = 0; j < 25; j += 5) {
#pragma unroll
            for (int i = 0; i < 5; ++i) bc[i] = st[j + i];
#pragma unroll
            for (int i = 0; i < 5; ++i) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[round];
    }
    out[0] = st[0]; out[1] = st[1]; out[2] = st[2]; out[3] = st[3];
}

__device__ __forceinline__ int d_b58_val(uint32_t c) {
    // "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz"
    if (c >= '1' && c <= '9') return (int)c - '1';
    if (c >= 'A' && c <= 'H') return (int)c - 'A' + 9;
    if (c >= 'J' && c <= 'N') return (int)c - 'J' + 17;
    if (c >= 'P' && c <= 'Z') return (int)c - 'P' + 22;
    if (c >= 'a' && c <= 'k') return (int)c - 'a' + 33;
    if (c >= 'm' && c <= 'z') return (int)c - 'm' + 44;
    return -1;
}
// bs58::decode(..).into_vec() into `dec` (big-endian), length returned, 0 on an invalid character. The number is kept
// in NL 32-bit limbs in registers (58^62 < 2^384: 12 limbs for Bitcoin; 58^110 < 2^672: 21 limbs for Monero).
template <int NL>
__device__ uint32_t d_base58_decode(const uint8_t* s, uint32_t n, uint8_t* dec) {
    uint32_t L[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) L[i] = 0;
    uint32_t zeros = 0;
    bool leading = true;
    for (uint32_t k = 0; k < n; ++k) {
        const int v = d_b58_val(s[k]);
        if (v < 0) return 0;
        if (leading && v == 0) { ++zeros; continue; }
        leading = false;
        uint32_t carry = (uint32_t)v;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const uint64_t t = (uint64_t)L[i] * 58u + carry;
            L[i] = (uint32_t)t;
            carry = (uint32_t)(t >> 32);
        }
    }
    // significant bytes of the number
    uint32_t nbytes = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) if (L[i]) nbytes = 4 * i + 1 + ((31u - (uint32_t)__clz((int)L[i])) >> 3);
    for (uint32_t k = 0; k < zeros; ++k) dec[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const uint32_t j = 4 * i + bb;  // little-endian byte index
            if (j < nbytes) dec[zeros + nbytes - 1 - j] = (uint8_t)(L[i] >> (8 * bb));
        }
    }
    return zeros + nbytes;
}
__device__ bool val_btc_base58(const uint8_t* s, uint32_t n, uint8_t* dec) {  // ext:1799-1822
    const uint32_t dl = d_base58_decode<12>(s, n, dec);
    if (dl < 5) return false;
    Sha256 h1, h2;
    h1.hash(dec, dl - 4);
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = h1.h[i];
    w[8] = 0x80000000u;
#pragma unroll
    for (int i = 9; i < 15; ++i) w[i] = 0;
    w[15] = 256;
    h2.init();
    h2.block(w);
    const uint32_t chk = ((uint32_t)dec[dl - 4] << 24) | ((uint32_t)dec[dl - 3] << 16) | ((uint32_t)dec[dl - 2] << 8) | dec[dl - 1];
    return h2.h[0] == chk;
}
__device__ bool val_monero(const uint8_t* s, uint32_t n, uint8_t* dec) {  // ext:1895-1920
    const uint32_t dl = d_base58_decode<21>(s, n, dec);
    if (dl < 5) return false;
    uint64_t h[4];
    d_keccak256_1blk(dec, dl - 4, h);  // dl <= 81
    const uint32_t chk = (uint32_t)dec[dl - 4] | ((uint32_t)dec[dl - 3] << 8) | ((uint32_t)dec[dl - 2] << 16) | ((uint32_t)dec[dl - 1] << 24);
    return (uint32_t)h[0] == chk;
}
__device__ int d_bech32_val(uint32_t c) {
    const char* CS = "qpzry9x8gf2tvdw0s3jn54khce6mua7l";
    for (int i = 0; i < 32; ++i) if ((uint32_t)CS[i] == c) return i;
    return -1;
}
// bech32::decode(addr) succeeds and hrp == "bc" (ext:1825-1835) for a token that starts with "bc1"
__device__ bool val_btc_bech32(const uint8_t* s, uint32_t n) {
    // separator = last '1'; it must be the one at index 2, so no '1' may follow; data symbols lower-case charset only
    // (hrp is lower-case, any upper-case letter makes the string mixed-case); non-ASCII bytes are invalid symbols.
    if (n < 3 + 6) return false;
    auto step = [](uint32_t chk, uint32_t v) {
        const uint32_t GEN[5] = {0x3b6a57b2, 0x26508e6d, 0x1ea119fa, 0x3d4233dd, 0x2a1462b3};
        uint32_t top = chk >> 25;
        chk = ((chk & 0x1ffffff) << 5) ^ v;
        for (int i = 0; i < 5; ++i) if ((top >> i) & 1) chk ^= GEN[i];
        return chk;
    };
    uint32_t chk = 1;
    chk = step(chk, 'b' >> 5); chk = step(chk, 'c' >> 5); chk = step(chk, 0); chk = step(chk, 'b' & 31); chk = step(chk, 'c' & 31);
    for (uint32_t k = 3; k < n; ++k) {
        int v = d_bech32_val(s[k]);
        if (v < 0) return false;
        chk = step(chk, (uint32_t)v);
    }
    return chk == 1 || chk == 0x2bc830a3u;
}
__device__ bool val_eth(const uint8_t* a, uint8_t* lower) {  // ext:1328-1361, 1840-1892: "0x" + 40 hex, EIP-55 when mixed case
    bool all_lower = true, all_upper = true;
    for (int i = 0; i < 40; ++i) {
        uint32_t c = a[2 + i];
        if (!d_is_hex(c)) return false;
        if (d_is_alpha(c)) { if (c >= 'a') all_upper = false; else all_lower = false; }
        lower[i] = (uint8_t)d_lower(c);
    }
    if (all_lower || all_upper) return true;
    uint64_t h[4];
    d_keccak256_1blk(lower, 40, h);
    for (int i = 0; i < 40; ++i) {
        uint32_t c = a[2 + i];
        if (d_is_alpha(c)) {
            const uint32_t byte = (uint32_t)(h[i >> 4] >> (8 * ((i >> 1) & 7))) & 0xFF;
            const uint32_t nib = (i & 1) ? (byte & 0x0f) : (byte >> 4);
            if ((c < 'a') != (nib >= 8)) return false;
        }
    }
    return true;
}

// k_validate_dom — stage A2a: one lane per domain anchor with its context record (planes written by k_anchor, read
// coalesced). Lean on purpose (no general walk, no global log reads): it is latency-bound, so its speed is the number of
// resident waves. Anchors it cannot decide (no context, last label > 7 bytes or not a suffix on its own, name longer than
// the context, non-ASCII) go to the rare list as RARE_DOM for k_validate.
// AC (p.filter_ac): databases with globs — a name is listed if its hash is in the literal bitmap or its text reaches an output
// state of the glob automaton (walked here from the context bytes, shallow rows in LDS), so k_lookup does not have to fetch
// the text of every valid name from the log again.
// MODE 2 (DevDb::sfx_bm): databases whose globs are all *LITERAL — the name must END with a literal, which a few hashed suffixes
// decide (no automaton walk, no chain of dependent look-ups); the candidate carries the verdict (Candidate::pad).
template <int MODE>
__global__ __launch_bounds__(256) void k_validate_dom(TokParams p, DevDb db) {
    constexpr bool AC = MODE == 1, SFX = MODE == 2;
    __shared__ uint2 tldtab[1u << TLD_TAB_BITS];
    __shared__ __attribute__((aligned(16))) uint32_t strbuf[256][8];   // per-lane context bytes for hashing
    __shared__ uint8_t dcls[AC ? 256 : 4];
    __shared__ uint32_t drows[AC ? DFA_LDS_ENTRIES_VAL : 1];
    for (uint32_t i = threadIdx.x; i < (1u << TLD_TAB_BITS); i += blockDim.x) tldtab[i] = db.tld_tab[i];
    DfaView dv{dcls, drows, 0};
    if constexpr (AC) dv = dfa_stage<DFA_LDS_ENTRIES_VAL>(db, dcls, drows);
    __syncthreads();
    // candidates: chunked when every valid domain is listed (high rate), buffered otherwise (no padding for k_lookup)
    ChunkWriter<Candidate, CAND_CHUNK> cw_dense;
    __shared__ Candidate wb_cand[4][64];
    BufferedWriter<Candidate> cw(wb_cand[threadIdx.x >> 6]);   // prefiltered (sparse) candidates
    __shared__ RareAnchor wb_slow[4][64];
    BufferedWriter<RareAnchor> sw(wb_slow[threadIdx.x >> 6]);   // undecided anchors -> rare list (sparse: dense, no padding)
    const Candidate SC{0, 0xFFFFFFFFu, 0, 0};
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t nd = min(p.counters->n_dom, p.dom_cap);
    struct Rec { uint32_t j; uint32_t c[8]; };
    auto load_rec = [&](uint32_t i, Rec& r) {
        r.j = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 8; ++k) r.c[k] = 0;
        if (i < nd) {
            r.j = p.dom_list[dom_plane_index(i, 0)];
#pragma unroll
            for (int k = 0; k < 8; ++k) r.c[k] = p.dom_list[dom_plane_index(i, 1 + k)];
        }
    };
    Rec cur, nxt;
    load_rec(blockIdx.x * blockDim.x + threadIdx.x, cur);
    {
        // consume the first record here, so that inside the loop no wait is placed between the store at the top of an
        // iteration and the first use of `cur` (the wait would also cover that store)
        uint32_t g = cur.j;
#pragma unroll
        for (int k = 0; k < 8; ++k) g ^= cur.c[k];
        asm volatile("" ::"v"(g));
    }
    // The candidate of iteration k is stored at the top of iteration k+1, before the next loads are issued: the wait for
    // those loads (vmcnt counts loads and stores in order) then never waits for a store in flight. The same delay gives
    // the literal-bitmap load (p.filter_lit) a whole iteration to arrive.
    uint32_t pend_start = 0, pend_lt = 0, pend_word = 0xFFFFFFFFu, pend_bit = 0, n_valid = 0;
    uint32_t pend_sw[SFX ? 4 : 1] = {0}, pend_sb[SFX ? 4 : 1] = {0};   // suffix filter: bitmap words in flight and their bits
    bool pend = false, pend_over = false;
    uint32_t* sb = strbuf[threadIdx.x];
    // TokParams::glob_work_d: the candidates this kernel flags CAND_GLOB go onto a work list of their own as their stage is flushed (their
    // indices in `cands` are known then), so that the glob pass over them can start when this kernel ends — beside the lean lookup pass over the
    // rest, which skips them (LookupParams::early_glob)
    auto queue_glob = [&](uint32_t b, uint32_t n) {
        if (!p.glob_work_d) return;
        const uint32_t lane = lane_id();
        const bool g = lane < n && b + lane < p.cand_cap && cw.buf[lane].pad == CAND_GLOB;
        const uint64_t m = __ballot(g);
        if (!m) return;
        uint32_t wb = 0;
        if (lane == 0) wb = atomicAdd(&p.counters->n_glob_work_d, (uint32_t)__popcll(m));
        wb = __builtin_amdgcn_readfirstlane(wb);
        const uint32_t slot = wb + mbcnt64(m);
        if (g && slot < p.glob_work_d_cap) p.glob_work_d[slot] = b + lane;
    };
    // the candidate of the previous iteration: listed if a literal key or (databases with globs) a glob can match it
    auto emit_pending = [&]() {
        if (!p.filter_lit) { cw_dense.append(pend, Candidate{pend_start, pend_lt, 0u, 0u}, p.cands, p.cand_cap, p.n_cand, SC); return; }
        bool lit = (pend_word >> pend_bit) & 1;
        uint32_t flag = 0;
        if constexpr (SFX) {
            bool g = pend_over;
#pragma unroll
            for (int q = 0; q < 4; ++q) g = g || ((pend_sw[q] >> pend_sb[q]) & 1);
            flag = g ? CAND_GLOB : CAND_NO_GLOB;
            lit = lit || g;
        }
        // the automaton was walked over the whole name right here: k_lookup need not walk it again to route the candidate
        if constexpr (AC) flag = pend_over ? CAND_GLOB : CAND_NO_GLOB;
        if constexpr (AC || SFX) cw.append_then(pend && lit, Candidate{pend_start, pend_lt, 0u, flag}, p.cands, p.cand_cap, p.n_cand, queue_glob);
        else cw.append(pend && lit, Candidate{pend_start, pend_lt, 0u, flag}, p.cands, p.cand_cap, p.n_cand);
    };
    for (uint32_t base = blockIdx.x * blockDim.x; base < nd; base += stride) {
        const uint32_t i = base + threadIdx.x;
        emit_pending();
        pend = false;
        load_rec(i + stride, nxt);
        bool slow = false;
        if (cur.j != 0xFFFFFFFFu) {
            uint32_t s = 0, e = 0;
            int r = 2;
            if (!(cur.j & 0x80000000u)) {
                const uint64_t b2 = (uint64_t)cur.c[0] | ((uint64_t)cur.c[1] << 32), b1 = (uint64_t)cur.c[2] | ((uint64_t)cur.c[3] << 32),
                               b0 = (uint64_t)cur.c[4] | ((uint64_t)cur.c[5] << 32);
                r = val_domain_pre(tldtab, p.min_labels, cur.j, make_uint2(cur.c[6], cur.c[7]), b0, b1, b2, s, e);
            }
            slow = r == 2;
            if (r == 1) {
                pend_start = s; pend_lt = (e - s) | ((uint32_t)IT_DOMAIN << 24); pend = true;
                pend_word = 0xFFFFFFFFu; pend_bit = 0;
                n_valid += 1;
                if (p.filter_lit) {
                    // hash of the name straight from the context record (no second read of the log): the 32 context
                    // bytes go through a per-lane LDS buffer to get the name aligned to 8-byte lanes
                    *reinterpret_cast<uint4*>(sb) = make_uint4(cur.c[0], cur.c[1], cur.c[2], cur.c[3]);
                    *reinterpret_cast<uint4*>(sb + 4) = make_uint4(cur.c[4], cur.c[5], cur.c[6], cur.c[7]);
                    const uint32_t o = 24 - (cur.j - s), n = e - s;   // offset and length of the name in the context
                    const uint32_t i0 = o >> 2, sh = o & 3;
                    uint32_t wd[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) wd[k] = (i0 + k) < 8 ? sb[(i0 + k) & 7] : 0u;
                    uint64_t ln[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        ln[k] = (uint64_t)__builtin_amdgcn_alignbyte(wd[2 * k + 1], wd[2 * k], sh) |
                                ((uint64_t)__builtin_amdgcn_alignbyte(wd[2 * k + 2], wd[2 * k + 1], sh) << 32);
                    if (db.ci) {   // the keys were lower-cased when the table was built; names decided here are pure ASCII
#pragma unroll
                        for (int k = 0; k < 4; ++k) ln[k] = ascii_lower8(ln[k]);
                    }
                    const uint32_t b = name_hash31(ln[0], ln[1], ln[2], ln[3], n) & db.lit_bm_mask;
                    pend_word = db.lit_bm ? db.lit_bm[b >> 5] : 0u;
                    pend_bit = b & 31;
                    if constexpr (SFX) {
                        // positions of the literals' first byte inside the name (bit k <-> byte k), from byte-lane SWAR on the name
                        const uint64_t fb = 0x0101010101010101ull * db.sfx_first;
                        uint32_t M = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint64_t x = ln[k] ^ fb;
                            const uint64_t z = ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x) & 0x8080808080808080ull;   // bit 7 of the bytes that are equal
                            const uint32_t lo = (uint32_t)z >> 7, hi = (uint32_t)(z >> 32) >> 7;
                            M |= ((((lo * 0x01020408u) >> 24) & 0xFu) | ((((hi * 0x01020408u) >> 24) & 0xFu) << 4)) << (8 * k);
                        }
                        M &= n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
                        pend_over = false;
                        uint32_t dm = db.sfx_dots;   // wave-uniform
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            pend_sw[q] = 0; pend_sb[q] = 0;
                            if (dm) {   // uniform
                                const uint32_t d = (uint32_t)__builtin_ctz(dm) + 1;
                                dm &= dm - 1;
                                uint32_t m = M;
                                for (uint32_t j = 1; j < d; ++j) m &= ~(m ? (0x80000000u >> __builtin_clz(m)) : 0u);   // drop the d-1 last occurrences
                                if (m) {
                                    const uint32_t kk = 31u - (uint32_t)__builtin_clz(m), sl = n - kk;   // the suffix starts at byte kk
                                    if (sl >= 3) {
                                        const uint32_t o2 = o + kk, j0 = o2 >> 2, s2 = o2 & 3;
                                        uint32_t w2[9];
#pragma unroll
                                        for (int k = 0; k < 9; ++k) w2[k] = (j0 + k) < 8 ? sb[(j0 + k) & 7] : 0u;
                                        uint64_t l2[4];
#pragma unroll
                                        for (int k = 0; k < 4; ++k)
                                            l2[k] = (uint64_t)__builtin_amdgcn_alignbyte(w2[2 * k + 1], w2[2 * k], s2) |
                                                    ((uint64_t)__builtin_amdgcn_alignbyte(w2[2 * k + 2], w2[2 * k + 1], s2) << 32);
                                        if (db.ci) {
#pragma unroll
                                            for (int k = 0; k < 4; ++k) l2[k] = ascii_lower8(l2[k]);
                                        }
                                        const uint32_t b2 = name_hash31(l2[0], l2[1], l2[2], l2[3], sl) & db.sfx_mask;
                                        pend_sw[q] = db.sfx_bm[b2 >> 5];
                                        pend_sb[q] = b2 & 31;
                                    }
                                }
                            }
                        }
                    }
                    if constexpr (AC) {
                        // the name is in ln[] already: byte -> class look-ups do not depend on the state and go out
                        // together, the chain is one LDS row look-up per byte
                        uint32_t st = 0, any = 0;
#pragma unroll
                        for (int k = 0; k < 31; ++k) {
                            const uint32_t c = dcls[(uint32_t)(ln[k >> 3] >> (8 * (k & 7))) & 0xFF];
                            if ((uint32_t)k < n) {
                                const uint32_t t = st < dv.lds_states ? drows[st * db.dfa_k + c] : db.dfa[(size_t)st * db.dfa_k + c];
                                st = t & 0x7FFFFFFFu;
                                any |= t;
                            }
                        }
                        pend_over = (any >> 31) != 0;
                        if (pend_over) { pend_word = 0xFFFFFFFFu; pend_bit = 0; }
                    }
                }
            }
        }
        sw.append(slow, RareAnchor{cur.j & 0x7FFFFFFFu, (uint32_t)RARE_DOM}, p.rare_dom, p.rare_dom_cap, &p.counters->n_rare_dom);
        cur = nxt;
    }
    emit_pending();
    if constexpr (AC || SFX) cw.flush_then(p.cands, p.cand_cap, p.n_cand, queue_glob);
    else cw.flush(p.cands, p.cand_cap, p.n_cand);
    cw_dense.pad_rest(p.cands, p.cand_cap, SC);
    sw.flush(p.rare_dom, p.rare_dom_cap, &p.counters->n_rare_dom);
    // validated domain candidates, listed or not
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n_valid += __shfl_down(n_valid, off);
    if (lane_id() == 0 && n_valid) atomicAdd(&p.counters->cand_true, n_valid);
}

// "Can a literal key be the text log[s, s + n)?" for the passes that list candidates one by one (DevDb::lit_bm: every key enters with its
// first 32 bytes and its length — engine.cpp). false = no key can be it: the candidate is counted and not listed. The keys of a
// case-insensitive database were lower-cased with the Unicode tables, which can change the LENGTH of a text that is not pure ASCII: such a
// text is decided here only when all of it was looked at (n <= 32, no byte >= 0x80) or the caller knows it is ASCII (e-mail addresses).
__device__ __forceinline__ bool lit_bm_may_hit(const DevDb& db, const LogView& lg, uint32_t s, uint32_t n, bool ascii) {
    uint64_t l[4] = {0, 0, 0, 0};
    if (s + 32 <= lg.len) {
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_memcpy(&l[k], lg.p + s + 8 * k, 8);
    } else {
        for (uint32_t k = 0; k < 32 && s + k < lg.len; ++k) l[k >> 3] |= (uint64_t)lg.p[s + k] << (8 * (k & 7));
    }
    if (db.ci) {
        if (!ascii) {
            if (n > 32) return true;
            uint64_t high = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rem = (int)n - 8 * k;
                const uint64_t m = rem >= 8 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << (8 * rem)) - 1ull));
                high |= l[k] & m & 0x8080808080808080ull;
            }
            if (high) return true;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) l[k] = ascii_lower8(l[k]);
    }
    const uint32_t b = name_hash31(l[0], l[1], l[2], l[3], n) & db.lit_bm_mask;
    return ((db.lit_bm[b >> 5] >> (b & 31)) & 1u) != 0;
}

// k_validate — stage A2b: one lane per long token (hex hashes; prefix tests for the address formats) and per rare
// anchor: IPv6, e-mail, and the domain anchors k_validate_dom could not decide (general right-to-left walk).
// VM = TokParams::vmode of the launch (bit 0: k_anchor's rare anchors, bit 1: the undecided domains, bit 2: the long tokens — a launch of
// their own in forked scans, so that k_rare, which takes what they leave, can start beside the rare anchors). The two halves
// need different LDS — the IPv6 windows (20 KB) for the first, the public-suffix tables (20 KB) for the second — and the first runs
// BESIDE k_validate_dom, whose workgroups hold most of a CU's LDS: with 26 KB instead of 47 its workgroups find room at once
// instead of waiting for k_validate_dom's to retire.
template <uint32_t VM>
__global__ __launch_bounds__(256) void k_validate(TokParams p, DevDb db) {
    constexpr bool MISC = (VM & 1u) != 0, DOM = (VM & 2u) != 0, TOK = (VM & 4u) != 0;
    __shared__ uint32_t bloom[DOM ? TLD_BLOOM_WORDS : 1];
    constexpr bool TLD_LDS = DOM && (MISC || TOK);   // the launch that only walks domains probes the table in global memory (LDS budget: below)
    __shared__ uint2 tldtab[TLD_LDS ? (1u << TLD_TAB_BITS) : 1];
    const uint2* const tldtab_p = TLD_LDS ? tldtab : db.tld_tab;
    // One 128-byte window of the log per lane (round 5; until then 80 bytes, IPv6 only). The e-mail and domain walks read the log a byte or
    // a word at a time at per-lane addresses — every such load is 64 scattered requests to the memory pipeline, 2.5 M of them per batch of an
    // application log (k_validate<1>: 0.85 ms for 2.9 M anchors). A lane now copies log[anchor - 64, anchor + 64) (domains: [j - 104, j + 24))
    // into its window with eight 16-byte loads and runs the SAME walk on a view of the window; when the walk touches an end of the window its
    // verdict would depend on bytes outside, and the lane walks the log itself as before.
    // LDS is what limits the resident waves of these latency-bound passes (they share the CUs with k_validate_dom and each other): the window
    // is 80 bytes with the IPv6 / e-mail list (what the IPv6 windows took before: e-mail [at - 48, at + 32)) and 64 bytes in the launch that only
    // walks domains ([j - 48, j + 16)), which in turn reads the suffix table from global memory (one probe per walk) instead of staging 16 KB
    // of it. With 128-byte windows the walks were twice as fast and the kernels no faster: two workgroups per CU instead of six.
    constexpr uint32_t WIN = MISC ? 80 : 64;
    constexpr uint32_t WIN_BACK = WIN - (MISC ? 32 : 16);   // bytes in front of the anchor
    __shared__ __attribute__((aligned(16))) uint8_t winbuf[(MISC || DOM) ? 256 * WIN : 16];
    if constexpr (DOM && !MISC && !TOK) {
        // the undecided domains are a few thousand per batch: workgroups beyond the list leave before they stage 20 KB of tables
        if (blockIdx.x * blockDim.x >= min(p.counters->n_rare_dom, p.rare_dom_cap)) return;
    }
    if constexpr (TOK && !MISC && !DOM) {
        if (blockIdx.x * blockDim.x >= min(p.counters->n_tok, p.tok_cap)) return;
    }
    if constexpr (DOM) {
        for (uint32_t i = threadIdx.x; i < TLD_BLOOM_WORDS; i += blockDim.x) bloom[i] = db.tld_bloom[i];
        if constexpr (TLD_LDS) for (uint32_t i = threadIdx.x; i < (1u << TLD_TAB_BITS); i += blockDim.x) tldtab[i] = db.tld_tab[i];
        __syncthreads();
    }
    LogView lg{p.log, p.len};
    __shared__ Candidate wb_cand[4][64];
    BufferedWriter<Candidate> cw(wb_cand[threadIdx.x >> 6]);
    const uint32_t stride = gridDim.x * blockDim.x;
    // Long tokens: hex hashes are decided here; the checksum validators (Base58Check, Bech32, EIP-55, Monero) need
    // SHA-256 / Keccak and hundreds of registers, so tokens that pass their cheap prefix tests go to the `heavy` list
    // for k_rare. A token made of non-boundary bytes may contain high bytes; every accepted form is pure ASCII, so the
    // reference's from_utf8 precondition is implied by the per-symbol checks.
    __shared__ RareAnchor wb_heavy[4][64];
    BufferedWriter<RareAnchor> hw(wb_heavy[threadIdx.x >> 6]);
    const uint32_t nt = TOK ? min(p.counters->n_tok, p.tok_cap) : 0u;
    uint32_t unlisted_tok = 0;   // valid hashes this lane did not list: no literal key can be them (lit_bm)
    // a hash can only hit through the literal table when the database has no glob section (with one, a substring literal or a glob may
    // match it: every valid hash is listed); DevDb::lit_bm then also holds the keys of 32 bytes and more, hashed from their first 32
    // bytes and their length
    // ... or when no literal of its automaton consists of letters and digits only (DevDb::ac_alnum: a million domain names as substring patterns,
    // globs over host names): neither a substring literal nor a glob can match a token then, and the walk of a 64-byte hash through the automaton —
    // 64 dependent row loads, in the last pass of the step — is not needed to know it
    const bool lit_only = p.filter_lit && !db.has_glob && db.lit_bm != nullptr;
    const bool tok_filter = p.filter_lit && db.lit_bm != nullptr && (!db.has_glob || !db.ac_alnum);
    // the list entry of the next round is fetched while this round's token is looked at (two registers; one round trip less in a loop that is a
    // chain of them: entry -> token bytes -> bitmap word)
    RareAnchor ra_next{0, 0xFF};
    if (TOK && blockIdx.x * blockDim.x + threadIdx.x < nt) ra_next = p.tok[blockIdx.x * blockDim.x + threadIdx.x];
    for (uint32_t base = blockIdx.x * blockDim.x; base < nt; base += stride) {
        const RareAnchor ra = ra_next;
        ra_next = RareAnchor{0, 0xFF};
        if (base + stride + threadIdx.x < nt) ra_next = p.tok[base + stride + threadIdx.x];
        const bool live = (ra.len_kind & 0xFF) == RARE_TOK;
        const uint32_t tl = ra.len_kind >> 8;
        const uint8_t* s = lg.p + (live ? ra.pos : 0);
        // The token's first 32 bytes (every token is at least 26 long; the bytes behind a short one belong to the log or to the 64
        // bytes of padding behind it... a token that ends within 32 bytes of the buffer's end is read byte-wise) in ONE round trip:
        // the prefix tests, the hex test of an MD5 and the bitmap hash all come out of these registers. One lane per token and a
        // dependent 8-byte load per step made this loop a chain of round trips to HBM per token (18 M tokens of a hash-dense log: 3.6 ms).
        uint64_t w[4] = {0, 0, 0, 0};
        if (live) {
            if (ra.pos + 32 <= lg.len) {
#pragma unroll
                for (int k = 0; k < 4; ++k) __builtin_memcpy(&w[k], s + 8 * k, 8);
            } else {
                for (uint32_t k = 0; k < 32 && ra.pos + k < lg.len; ++k) w[k >> 3] |= (uint64_t)s[k] << (8 * (k & 7));
            }
        }
        // ... and bytes 32..63 of a longer token in the SAME round trip (a SHA-1 / SHA-256 is looked at to its end in most cases: the second load
        // behind the test of the first half was another trip to memory for six tokens in ten of an endpoint log)
        uint64_t v2[4] = {0, 0, 0, 0};
        const bool wide2 = live && tl > 32 && ra.pos + 64 <= lg.len;
        if (wide2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) __builtin_memcpy(&v2[k], s + 32 + 8 * k, 8);
        }
        const uint32_t c0 = (uint32_t)w[0] & 0xFF, c1 = (uint32_t)(w[0] >> 8) & 0xFF, c2 = (uint32_t)(w[0] >> 16) & 0xFF;
        // hashes: token length 32/40/64/96/128 and all hex (ext:1212-1250)
        {
            Candidate c{0, 0, 0, 0};
            bool emit = false;
            if (live && (p.flags & EX_HASHES)) {
                const int ht = tl == 32 ? IT_MD5 : tl == 40 ? IT_SHA1 : tl == 64 ? IT_SHA256 : tl == 96 ? IT_SHA384 : tl == 128 ? IT_SHA512 : -1;
                if (ht >= 0) {
                    bool hex = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) hex = hex && hex4((uint32_t)w[k]) && hex4((uint32_t)(w[k] >> 32));
                    if (hex && tl > 32) {
                        // bytes 32..63 in one round trip (the hash lengths are multiples of 8: whole words only); a SHA-384 / SHA-512 goes on from byte 64.
                        // (all_hex_wide alone compiled to a chain: load 4 bytes, wait, test, branch — eight round trips for a SHA-256)
                        if (wide2) {
                            bool h2 = true;
#pragma unroll
                            for (int k = 0; k < 4; ++k) h2 = h2 & (((int)tl - 32 - 8 * k < 8) | (hex4((uint32_t)v2[k]) & hex4((uint32_t)(v2[k] >> 32))));
                            hex = h2;
                            if (hex && tl > 64) hex = all_hex_wide(s + 64, tl - 64);
                        } else {
                            hex = all_hex_wide(s + 32, tl - 32);
                        }
                    }
                    if (hex) {
                        c.start = ra.pos; c.len_type = tl | ((uint32_t)ht << 24); emit = true;
                        if (tok_filter) {
                            uint64_t l0 = w[0], l1 = w[1], l2 = w[2], l3 = w[3];
                            if (db.ci) { l0 = ascii_lower8(l0); l1 = ascii_lower8(l1); l2 = ascii_lower8(l2); l3 = ascii_lower8(l3); }
                            const uint32_t b = name_hash31(l0, l1, l2, l3, tl) & db.lit_bm_mask;
                            if (!((db.lit_bm[b >> 5] >> (b & 31)) & 1u)) { emit = false; ++unlisted_tok; }
                        }
                    }
                }
            }
            cw.append(emit, c, p.cands, p.cand_cap, p.n_cand);
        }
        // a token can yield several items: hash, Bitcoin, Ethereum and Monero are independent extractors
        {
            uint32_t hk = 0;
            if (live && (p.flags & EX_BITCOIN) && tl >= 26 && tl <= 62) {
                if (c0 == 'b' && c1 == 'c' && c2 == '1') hk = HEAVY_BECH32;
                else if (c0 == '1' || c0 == '3') {
                    // Base58Check starts with decoding, and decoding fails on a symbol outside the Bitcoin alphabet (bs58: lib.rs:1799-1822):
                    // a token with '0', 'O', 'I' or 'l' in it — every second lower-case hex hash that starts with 1 or 3 — need not go to the
                    // checksum kernel, whose waves would run the whole decode + double SHA-256 for the one lane in 64 that can pass.
                    // The first 32 bytes are tested here (bytes behind a shorter token are masked out); k_rare decides the rest.
                    auto has_byte = [](uint64_t x, uint32_t v) {   // bit 7 of every byte of x that equals v
                        const uint64_t y = x ^ (0x0101010101010101ull * v);
                        return ~(((y & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | y) & 0x8080808080808080ull;
                    };
                    uint64_t bad = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int rem = (int)tl - 8 * k;
                        const uint64_t m = rem >= 8 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << (8 * rem)) - 1ull));
                        bad |= (has_byte(w[k], '0') | has_byte(w[k], 'O') | has_byte(w[k], 'I') | has_byte(w[k], 'l')) & m;
                    }
                    // ... and the bytes behind them, for the one token in sixty that is longer and got this far (a 64-character hash without a
                    // '0' in its first half: 13 % of those that start with 1 or 3; without one anywhere: 2 %): one more round trip for those lanes
                    // instead of a Base58 decode and two SHA-256 in k_rare for eight times as many entries
                    if (!bad && wide2) {
                        const uint64_t (&w2)[4] = v2;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int rem = (int)tl - 32 - 8 * k;
                            const uint64_t m = rem >= 8 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << (8 * rem)) - 1ull));
                            bad |= (has_byte(w2[k], '0') | has_byte(w2[k], 'O') | has_byte(w2[k], 'I') | has_byte(w2[k], 'l')) & m;
                        }
                    }
                    if (!bad) hk = HEAVY_B58;
                }
            }
            hw.append(hk != 0, RareAnchor{ra.pos, (tl << 8) | hk}, p.heavy, p.heavy_cap, &p.counters->n_heavy);
            hk = 0;
            if (live && (p.flags & EX_ETHEREUM) && tl == 42 && c0 == '0' && c1 == 'x') hk = HEAVY_ETH;
            if (live && (p.flags & EX_MONERO) && tl >= 90 && tl <= 110 && (c0 == '4' || c0 == '8')) hk = HEAVY_XMR;
            hw.append(hk != 0, RareAnchor{ra.pos, (tl << 8) | hk}, p.heavy, p.heavy_cap, &p.counters->n_heavy);
        }
    }
    hw.flush(p.heavy, p.heavy_cap, &p.counters->n_heavy);
    // IPv6 ("::") and e-mail ('@') anchors from the rare list. The IPv6 parser reads its bytes many times: each lane
    // copies log[p2-40, p2+40) into its LDS window with five wide loads first.
    uint8_t* win = winbuf + ((MISC || DOM) ? threadIdx.x * WIN : 0);
    // copies log[from, from + 128) into the lane's window (callers make sure the range lies inside the buffer)
    auto fill_window = [&](uint32_t from) {
        uint4 v[WIN / 16];
#pragma unroll
        for (uint32_t k = 0; k < WIN / 16; ++k) __builtin_memcpy(&v[k], lg.p + from + 16 * k, 16);
#pragma unroll
        for (uint32_t k = 0; k < WIN / 16; ++k) *reinterpret_cast<uint4*>(win + 16 * k) = v[k];
    };
    // two lists through the same code: k_anchor's rare anchors (vmode bit 0) and the domain anchors k_validate_dom left (bit 1);
    // the engine runs the first beside k_validate_dom on a stream of its own and the second behind it
    uint32_t unlisted = unlisted_tok;   // valid candidates this lane did not list (they cannot hit)
    for (uint32_t li = 0; li < 2; ++li) {
    if (!((VM >> li) & 1u)) continue;
    const RareAnchor* rlist = li ? p.rare_dom : p.rare;
    const uint32_t nr = li ? min(p.counters->n_rare_dom, p.rare_dom_cap) : min(p.counters->n_rare, p.rare_cap);
    for (uint32_t base = blockIdx.x * blockDim.x; base < nr; base += stride) {
        const uint32_t i = base + threadIdx.x;
        RareAnchor ra{0, 0xFF};
        if (i < nr) ra = rlist[i];
        const uint32_t kind = ra.len_kind & 0xFF;
        Candidate c{0, 0, 0, 0};
        bool emit = false;
        if (MISC && kind == RARE_V6) {
            uint32_t s, e;
            bool ok;
            if (ra.pos >= 40 && ra.pos + 40 <= lg.len) {
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    uint4 v;
                    __builtin_memcpy(&v, lg.p + ra.pos - 40 + 16 * k, 16);
                    *reinterpret_cast<uint4*>(win + 16 * k) = v;
                }
                ok = val_ipv6_win(win, ra.pos, s, e);
            } else {
                ok = val_ipv6(lg, ra.pos, s, e);
            }
            if (ok) { c.start = s; c.len_type = (e - s) | ((uint32_t)IT_IPV6 << 24); emit = true; }
            // An IPv4 tree answers an IPv6 address from its first 32 bits only (SearchTree::lookup_v6 walks from node 0 whatever the
            // tree's version, tree.rs:92-125, and a 32-level tree ends there): the /24 bitmap of the database decides for most
            // addresses that nothing can be found — such candidates are counted, not listed (like the IPv4 candidates of k_anchor).
            if (ok && p.filter_v4 && db.ip_version == 4 && ra.pos >= 40 && ra.pos + 40 <= lg.len && e - s <= 39) {
                uint16_t seg[8];
                if (d_parse_ipv6(win + (s - (ra.pos - 40)), e - s, seg)) {
                    const uint32_t top24 = ((uint32_t)seg[0] << 8) | (seg[1] >> 8);
                    if (!((db.ip_bm24[top24 >> 5] >> (top24 & 31)) & 1)) { emit = false; ++unlisted; }
                }
            }
        }
        // e-mail anchors: the local part leftwards (a lane that meets a long run gets the wave's help), then the rest
        {
            bool is_at = MISC && kind == RARE_AT;
            if constexpr (MISC) {
                // 48 bytes in front of the '@' and 32 behind it in registers first (val_email_ctx)
                const bool wide = is_at && ra.pos >= 48 && ra.pos + 1 + 32 + 8 <= lg.len;
                if (__ballot(wide)) {
                    if (wide) {
                        uint64_t lw[6], rw[4];
#pragma unroll
                        for (int k = 0; k < 6; ++k) __builtin_memcpy(&lw[k], lg.p + ra.pos - 8 * (k + 1), 8);
#pragma unroll
                        for (int k = 0; k < 4; ++k) __builtin_memcpy(&rw[k], lg.p + ra.pos + 1 + 8 * k, 8);
                        uint32_t s2 = 0, e2 = 0;
                        const int rw_ = val_email_ctx<6, 4>(db.tld_tab, db, lg, ra.pos, lw, rw, s2, e2);
                        if (rw_ != 2) {
                            is_at = false;
                            if (rw_ == 1) { c.start = s2; c.len_type = (e2 - s2) | ((uint32_t)IT_EMAIL << 24); emit = true; }
                        }
                    }
                }
            }
            if constexpr (MISC) {
                // the window first: most addresses are short and stand in the middle of a line
                const bool inner = is_at && ra.pos >= WIN_BACK && ra.pos + (WIN - WIN_BACK) <= lg.len;
                if (__ballot(inner)) {
                    if (inner) fill_window(ra.pos - WIN_BACK);
                    __builtin_amdgcn_wave_barrier();
                    if (inner) {
                        const LogView wv{win, WIN};
                        EmailState ws{WIN_BACK, (uint32_t)'@', false, false};
                        bool edge = email_local_scan(wv, ws, false);   // (no budget: a window is ten words)
                        uint32_t s2 = 0, e2 = 0;
                        const bool okw = val_email_finish(wv, db, WIN_BACK, ws, s2, e2, &edge);
                        if (!edge) {
                            is_at = false;   // decided
                            if (okw) { c.start = s2 + (ra.pos - WIN_BACK); c.len_type = (e2 - s2) | ((uint32_t)IT_EMAIL << 24); emit = true; }
                        }
                    }
                }
            }
            EmailState es{ra.pos, (uint32_t)'@', false, false};
            bool el = is_at && email_local_scan(lg, es, true);
            for (uint64_t lm = __ballot(el); lm; lm &= lm - 1) {
                const int src = __ffsll((long long)lm) - 1;
                EmailState st;
                st.s = (uint32_t)__shfl((int)es.s, src); st.prev = (uint32_t)__shfl((int)es.prev, src);
                st.has_letter = __shfl((int)es.has_letter, src) != 0; st.dotdot = __shfl((int)es.dotdot, src) != 0;
                coop_email_skip(lg, st);
                if ((int)lane_id() == src) es = st;
            }
            if (el) (void)email_local_scan(lg, es, false);
            uint32_t s, e;
            if (is_at && val_email_finish(lg, db, ra.pos, es, s, e)) {
                if (e - s > 0xFFFFFFu) atomicOr(&p.counters->error, 4u);   // the record format holds 24-bit lengths
                c.start = s; c.len_type = (e - s) | ((uint32_t)IT_EMAIL << 24); emit = true;
            }
        }
        // domain anchors k_validate_dom left undecided: the general walk; a lane that meets a long run (hostile input: one
        // token of megabytes) reports it, the wave walks the bulk of that run together, the lane finishes
        DomLong dl{};
        int dr = WALK_NO;
        uint32_t ds = 0, de = 0;
        if constexpr (DOM) {
            bool todo = kind == RARE_DOM;
            // 48 context bytes in registers first (val_domain_ctx): most of what k_validate_dom leaves is a name that is merely longer than ITS context
            const bool wide = todo && ra.pos >= 48 && ra.pos + 8 <= lg.len;
            if (__ballot(wide)) {
                if (wide) {
                    uint64_t cb[6];
                    uint2 cw8;
#pragma unroll
                    for (int k = 0; k < 6; ++k) __builtin_memcpy(&cb[k], lg.p + ra.pos - 8 * (k + 1), 8);
                    __builtin_memcpy(&cw8, lg.p + ra.pos, 8);
                    uint32_t s2 = 0, e2 = 0;
                    const int rw = val_domain_ctx<6>(tldtab_p, p.min_labels, ra.pos, cw8, cb, s2, e2);
                    if (rw != 2) { todo = false; dr = rw == 1 ? WALK_YES : WALK_NO; ds = s2; de = e2; }
                }
            }
            const bool inner = todo && ra.pos >= WIN_BACK && ra.pos + (WIN - WIN_BACK) <= lg.len;
            if (__ballot(inner)) {
                if (inner) fill_window(ra.pos - WIN_BACK);
                __builtin_amdgcn_wave_barrier();
                if (inner) {
                    const LogView wv{win, WIN};
                    bool edge = false;
                    uint32_t s2 = 0, e2 = 0;
                    const int rw = val_domain(wv, db, bloom, tldtab_p, p.min_labels, WIN_BACK, s2, e2, nullptr, &edge);
                    if (!edge) { todo = false; dr = rw; ds = s2 + (ra.pos - WIN_BACK); de = e2 + (ra.pos - WIN_BACK); }
                }
            }
            if (todo) {
                // The walk over the log itself is a chain of dependent 8-byte loads going left from the anchor, and the lanes of a wave take them
                // one after the other (different lengths, different branches). Touch the two 128-byte lines in front of the anchor first — one
                // round trip to HBM for both — and the chain runs out of the cache.
                const uint32_t a0 = ra.pos & ~127u;
                uint32_t t0 = lg.p[a0], t1 = a0 >= 128 ? lg.p[a0 - 128] : 0u;
                asm volatile("" ::"v"(t0), "v"(t1));
                dr = val_domain(lg, db, bloom, tldtab_p, p.min_labels, ra.pos, ds, de, &dl);
            }
        }
        for (uint64_t lm = __ballot(dr == WALK_LONG); lm; lm &= lm - 1) {
            const int src = __ffsll((long long)lm) - 1;
            WalkState st;
            st.pos = (uint32_t)__shfl((int)dl.st.pos, src); st.cur = (uint32_t)__shfl((int)dl.st.cur, src);
            st.last_c = (uint32_t)__shfl((int)dl.st.last_c, src); st.labels = (uint32_t)__shfl((int)dl.st.labels, src);
            st.high = __shfl((int)dl.st.high, src) != 0; st.bad = __shfl((int)dl.st.bad, src) != 0; st.found = __shfl((int)dl.st.found, src) != 0;
            coop_domain_skip(lg, st);
            if ((int)lane_id() == src) dl.st = st;
        }
        bool need_u8 = false;
        if (dr == WALK_LONG) dr = val_domain_resume(lg, db, p.min_labels, dl, ds, de, need_u8) ? WALK_YES : WALK_NO;
        for (uint64_t lm = __ballot(dr == WALK_YES && need_u8); lm; lm &= lm - 1) {   // long runs with non-ASCII bytes
            const int src = __ffsll((long long)lm) - 1;
            const uint32_t us = (uint32_t)__shfl((int)ds, src), ue = (uint32_t)__shfl((int)de, src);
            const bool good = coop_valid_utf8(lg.p + us, ue - us);
            if ((int)lane_id() == src && !good) dr = WALK_NO;
        }
        if (dr == WALK_YES) {
            if (de - ds > 0xFFFFFFu) atomicOr(&p.counters->error, 4u);
            c.start = ds; c.len_type = (de - ds) | ((uint32_t)IT_DOMAIN << 24); emit = true;
        }
        // E-mail addresses and the names of the general walk, like the names k_validate_dom decides and the hashes above: in a database
        // without globs only a literal key can match them, and the bitmap of the keys says which of them could (an application log with
        // an address in every fourth line, a proxy log with 2 M long host names: the lookup pass behind this one read every one of them
        // again, 0.1-0.2 ms at the end of the step)
        if (lit_only && emit) {
            const uint32_t ty = c.len_type >> 24;
            if ((ty == IT_EMAIL || ty == IT_DOMAIN) && !lit_bm_may_hit(db, lg, c.start, c.len_type & 0xFFFFFFu, ty == IT_EMAIL)) { emit = false; ++unlisted; }
        }
        cw.append(emit, c, p.cands, p.cand_cap, p.n_cand);
    }
    }
    cw.flush(p.cands, p.cand_cap, p.n_cand);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) unlisted += __shfl_down(unlisted, off);
    if (lane_id() == 0 && cw.total + unlisted) atomicAdd(&p.counters->cand_true, cw.total + unlisted);
}

// k_rare — stage A3: checksum validators (Base58Check, Bech32, EIP-55, Monero): very rare in logs and heavy in
// registers, one lane per entry of the `heavy` list written by k_validate.
__global__ __launch_bounds__(64) void k_rare(TokParams p, DevDb db) {
    // per lane: the token (<= 110 bytes, copied with wide loads) and the decoded address / lower-cased hex
    __shared__ __attribute__((aligned(16))) uint8_t tokbuf[64][112];
    __shared__ __attribute__((aligned(16))) uint8_t decbuf[64][96];
    LogView lg{p.log, p.len};
    __shared__ Candidate wb_cand[64];
    BufferedWriter<Candidate> cw(wb_cand);
    const uint32_t n = min(p.counters->n_heavy, p.heavy_cap);
    uint8_t* tb = tokbuf[threadIdx.x];
    uint8_t* dec = decbuf[threadIdx.x];
    for (uint32_t base = blockIdx.x * 64; base < n; base += gridDim.x * 64) {
        const uint32_t i = base + threadIdx.x;
        RareAnchor ra{0, 0xFF};
        if (i < n) ra = p.heavy[i];
        const uint32_t kind = ra.len_kind & 0xFF, tl = ra.len_kind >> 8;
        bool em = false;
        Candidate ct{0, 0, 0, 0};
        if (kind != 0xFF && tl <= 110) {
            // token bytes -> LDS: 16 bytes per load while they are inside the buffer, bytes at its very end
            for (uint32_t k = 0; k < tl; k += 16) {
                if (ra.pos + k + 16 <= lg.len) {
                    uint4 v;
                    __builtin_memcpy(&v, lg.p + ra.pos + k, 16);
                    *reinterpret_cast<uint4*>(tb + k) = v;
                } else {
                    for (uint32_t b = k; b < tl; ++b) tb[b] = lg.p[ra.pos + b];
                }
            }
            int ty = -1;
            if (kind == HEAVY_BECH32) { if (val_btc_bech32(tb, tl)) ty = IT_BITCOIN; }
            else if (kind == HEAVY_B58) { if (val_btc_base58(tb, tl, dec)) ty = IT_BITCOIN; }
            else if (kind == HEAVY_ETH) { if (val_eth(tb, dec)) ty = IT_ETHEREUM; }
            else if (kind == HEAVY_XMR) { if (val_monero(tb, tl, dec)) ty = IT_MONERO; }
            if (ty >= 0) { ct.start = ra.pos; ct.len_type = tl | ((uint32_t)ty << 24); em = true; }
        }
        cw.append(em, ct, p.cands, p.cand_cap, p.n_cand);
    }
    cw.flush(p.cands, p.cand_cap, p.n_cand);
    if (lane_id() == 0 && cw.total) atomicAdd(&p.counters->cand_true, cw.total);
}

// ------------------------------------------------------------------------------------------------ launch wrappers
int validate_blocks_per_cu(bool ac) {
    int n = 0;
    const hipError_t e = ac ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_validate_dom<1>, 256, 0)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_validate_dom<0>, 256, 0);
    if (e != hipSuccess || n < 1) n = ac ? 3 : 4;
    return n;
}
// grid = workgroups of k_validate_dom
void launch_validate_dom(const TokParams& p, const DevDb& db, int grid, hipStream_t stream) {
    if (!(p.flags & EX_DOMAINS)) return;
    if (p.filter_ac && db.sfx_bm) hipLaunchKernelGGL(k_validate_dom<2>, dim3(grid), dim3(256), 0, stream, p, db);
    else if (p.filter_ac) hipLaunchKernelGGL(k_validate_dom<1>, dim3(grid), dim3(256), 0, stream, p, db);
    else hipLaunchKernelGGL(k_validate_dom<0>, dim3(grid), dim3(256), 0, stream, p, db);
    check_launch("launch_validate_dom");
}
// k_validate (tokens, rare anchors, undecided domains: TokParams::vmode says which lists) has a fraction of the work and is
// latency-bound: every workgroup stages the suffix tables first, so few workgroups (one per CU measured best for the whole job)
static int env_int(const char* name) { const char* v = getenv(name); return v ? atoi(v) : 0; }
void launch_validate_misc(const TokParams& p, const DevDb& db, int grid, hipStream_t stream) {
    // experiments: MATCHY_AMD_MISC_GRID_V<vmode>=<workgroups> overrides the grid of one variant (tools/sweep_misc_grid.sh)
    static const int ov[8] = {0, env_int("MATCHY_AMD_MISC_GRID_V1"), env_int("MATCHY_AMD_MISC_GRID_V2"), 0, env_int("MATCHY_AMD_MISC_GRID_V4"),
                              env_int("MATCHY_AMD_MISC_GRID_V5"), 0, env_int("MATCHY_AMD_MISC_GRID_V7")};
    if (ov[p.vmode & 7u] > 0) grid = ov[p.vmode & 7u];
    if (p.vmode == 5u) hipLaunchKernelGGL(k_validate<5u>, dim3(grid), dim3(256), 0, stream, p, db);
    else if (p.vmode == 4u) hipLaunchKernelGGL(k_validate<4u>, dim3(grid), dim3(256), 0, stream, p, db);
    else if (p.vmode == 1u) hipLaunchKernelGGL(k_validate<1u>, dim3(grid), dim3(256), 0, stream, p, db);
    else if (p.vmode == 2u) hipLaunchKernelGGL(k_validate<2u>, dim3(grid), dim3(256), 0, stream, p, db);
    else hipLaunchKernelGGL(k_validate<7u>, dim3(grid), dim3(256), 0, stream, p, db);
    check_launch("launch_validate_misc");
}
void launch_rare(const TokParams& p, const DevDb& db, int grid, hipStream_t stream) {
    hipLaunchKernelGGL(k_rare, dim3(grid), dim3(64), 0, stream, p, db);
    check_launch("launch_rare");
}

}  // namespace mxy
