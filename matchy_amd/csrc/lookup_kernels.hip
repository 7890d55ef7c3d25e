// HIP kernels of stage B of the `matchy match` hot path on gfx950 (MI355X, wave64): the database lookups.
//
//   k_lookup    one lane per candidate: MMDB trie walk / XXH64 literal probe / Aho-Corasick DFA + glob verification;
//               every hit leaves at once as its final matchy_scan_hit_t record (pack_record: device copy + pinned host
//               mirror, pattern ids resolved to data offsets).
//
// Semantics follow the reference CPU path; every rule cites the reference function it reproduces
// (matchy-format/src/mmdb/tree.rs = "tree", matchy-literal-hash/src/lib.rs = "lh",
// matchy-paraglob/src/paraglob_offset.rs = "pg").
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "hashes.h"
#include "scan_types.h"

#include "device_shared.h"

namespace mxy {

// ------------------------------------------------------------------------------------------------ stage B: lookups
// SearchTree::lookup_v4 (tree:46-90): trie_v4 in device_shared.h (k_anchor uses it too). lookup_v6 (tree:92-125):
__device__ bool trie_v6(const DevDb& db, const uint16_t seg[8], uint32_t& data_off, uint32_t& prefix) {
    uint32_t node = 0;
    for (int bi = 0; bi < 128; ++bi) {
        uint2 nd = db.ip_nodes[node];
        uint32_t bit = (seg[bi >> 4] >> (15 - (bi & 15))) & 1;
        uint32_t rec = bit ? nd.y : nd.x;
        if (rec == db.node_count) return false;
        if (rec < db.node_count) node = rec;
        else {
            uint32_t off = rec - db.node_count;
            if (off < 16) return false;
            data_off = off - 16;
            prefix = (uint32_t)bi + 1;
            return true;
        }
    }
    return false;
}

// ---- case-insensitive databases: Rust str::to_lowercase on the device (texts with non-ASCII characters only; pure ASCII is
// folded inline). The character data comes from matchy_amd/data/lowercase.bin (DevDb::lc_*).
__device__ __forceinline__ uint32_t d_utf8_decode(const uint8_t* s, uint32_t& cp) {   // valid UTF-8 only
    const uint32_t c = s[0];
    if (c < 0x80) { cp = c; return 1; }
    if (c < 0xE0) { cp = ((c & 0x1Fu) << 6) | (s[1] & 0x3Fu); return 2; }
    if (c < 0xF0) { cp = ((c & 0x0Fu) << 12) | ((s[1] & 0x3Fu) << 6) | (s[2] & 0x3Fu); return 3; }
    cp = ((c & 0x07u) << 18) | ((s[1] & 0x3Fu) << 12) | ((s[2] & 0x3Fu) << 6) | (s[3] & 0x3Fu);
    return 4;
}
__device__ bool d_cp_in_ranges(const uint2* r, uint32_t n, uint32_t cp) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (r[mid].y < cp) lo = mid + 1; else hi = mid; }
    return lo < n && r[lo].x <= cp;
}
// case_ignorable_then_cased (alloc/src/str.rs) over text[0, i) right to left / text[i, n) left to right
__device__ bool d_cased_behind(const DevDb& db, const uint8_t* s, uint32_t i) {
    while (i > 0) {
        uint32_t j = i - 1;
        while (j > 0 && (s[j] & 0xC0) == 0x80) --j;
        uint32_t cp;
        d_utf8_decode(s + j, cp);
        if (!d_cp_in_ranges(db.lc_ign, db.lc_n_ign, cp)) return d_cp_in_ranges(db.lc_cased, db.lc_n_cased, cp);
        i = j;
    }
    return false;
}
__device__ bool d_cased_ahead(const DevDb& db, const uint8_t* s, uint32_t i, uint32_t n) {
    while (i < n) {
        uint32_t cp;
        const uint32_t a = d_utf8_decode(s + i, cp);
        if (!d_cp_in_ranges(db.lc_ign, db.lc_n_ign, cp)) return d_cp_in_ranges(db.lc_cased, db.lc_n_cased, cp);
        i += a;
    }
    return false;
}
// Rust str::to_lowercase of the valid UTF-8 text s[0, n) as a byte stream: the lower-cased form is hashed and compared
// while it is produced and never stored, so its length is not limited by a buffer.
struct LowerStream {
    const DevDb& db;
    const uint8_t* s;
    uint32_t n, i = 0, np = 0;
    uint64_t pend = 0;   // lower-cased bytes of the current character (at most 7)
    __device__ LowerStream(const DevDb& d, const uint8_t* text, uint32_t len) : db(d), s(text), n(len) {}
    __device__ bool next(uint32_t& b) {
        if (np == 0) {
            if (i >= n) return false;
            uint32_t cp;
            const uint32_t a = d_utf8_decode(s + i, cp);
            if (cp < 0x80) { pend = ascii_lower1(cp); np = 1; }
            else if (cp == 0x3A3) {   // capital sigma: final form iff preceded by a cased letter and not followed by one
                const bool fin = d_cased_behind(db, s, i) && !d_cased_ahead(db, s, i + a, n);
                pend = 0xCFull | ((fin ? 0x82ull : 0x83ull) << 8); np = 2;
            } else {
                uint32_t lo = 0, hi = db.lc_n;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (db.lc_map[mid * 3] < cp) lo = mid + 1; else hi = mid; }
                if (lo < db.lc_n && db.lc_map[lo * 3] == cp) {
                    const uint32_t e1 = db.lc_map[lo * 3 + 1];
                    np = e1 & 0xFF; pend = (uint64_t)(e1 >> 8) | ((uint64_t)db.lc_map[lo * 3 + 2] << 24);
                } else {
                    pend = 0; np = a;
                    for (uint32_t k = 0; k < a; ++k) pend |= (uint64_t)s[i + k] << (8 * k);
                }
            }
            i += a;
        }
        b = (uint32_t)pend & 0xFF; pend >>= 8; --np;
        return true;
    }
};

// LiteralHash::lookup (lh:467-525) over the re-hashed device table. Case-insensitive databases (lh:469-472): the query is
// lower-cased first — pure-ASCII text inline while it is hashed and compared, anything else through LowerStream.
__device__ bool lit_lookup(const DevDb& db, const uint8_t* s0, uint32_t n0, uint32_t& pattern_id) {
    const uint8_t* s = s0;
    uint32_t n = n0;
    bool fold = false, stream = false;
    uint64_t h;
    if (db.ci) {
        uint64_t hi = 0;
        uint32_t k = 0;
        for (; k + 8 <= n0; k += 8) { uint64_t x; __builtin_memcpy(&x, s0 + k, 8); hi |= x; }
        for (; k < n0; ++k) hi |= s0[k];
        if ((hi & 0x8080808080808080ull) == 0) fold = true;
        else stream = true;
    }
    if (stream) {
        // non-ASCII query of a case-insensitive database: hash the lower-cased stream (first pass), compare it with the
        // key of a matching slot while producing it again (second pass)
        LowerStream ls(db, s0, n0);
        Xxh64Stream xs(0);
        uint32_t b;
        n = 0;
        while (ls.next(b)) {
            xs.push(b);
            if (++n > db.lit_max_len) return false;   // longer than every key
        }
        h = xs.finish(0);
    } else {
        h = fold ? xxh64<true>(s, n, 0) : xxh64<false>(s, n, 0);
    }
    uint32_t slot = (uint32_t)(h ^ (h >> 32)) & db.lit_mask;
    for (;;) {
        LitSlot e = db.lit_slots[slot];
        if (e.str_off == 0xFFFFFFFFu) return false;
        if (e.hash == h) {
            const uint8_t* q = db.lit_pool + e.str_off;
            uint32_t sl = (uint32_t)q[0] | ((uint32_t)q[1] << 8);
            if (sl == n) {
                uint64_t diff = 0;
                if (stream) {
                    LowerStream ls(db, s0, n0);
                    uint32_t b, k = 0;
                    while (ls.next(b)) { diff |= (uint64_t)(q[2 + k] ^ b); ++k; }
                } else {
                    // 8 bytes per step, no early exit: the loads are independent of each other; the last up to 7 bytes are
                    // compared through one 8-byte pair that ends at the end of both strings (overlapping bytes compare equal
                    // again) instead of a byte loop of dependent round trips
                    uint32_t k = 0;
                    for (; k + 8 <= n; k += 8) {
                        uint64_t x, y;
                        __builtin_memcpy(&x, q + 2 + k, 8);
                        __builtin_memcpy(&y, s + k, 8);
                        diff |= x ^ (fold ? ascii_lower8(y) : y);
                    }
                    if (k < n) {
                        if (n >= 8) {
                            uint64_t x, y;
                            __builtin_memcpy(&x, q + 2 + n - 8, 8);
                            __builtin_memcpy(&y, s + n - 8, 8);
                            diff |= x ^ (fold ? ascii_lower8(y) : y);
                        } else {
                            for (; k < n; ++k) diff |= (uint64_t)(q[2 + k] ^ (fold ? ascii_lower1(s[k]) : (uint32_t)s[k]));
                        }
                    }
                }
                if (diff == 0) { pattern_id = e.pattern_id; return true; }
            }
        }
        slot = (slot + 1) & db.lit_mask;
    }
}

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }

// find_ac_transition (pg:1271-1353): returns target node offset or 0xFFFFFFFF
__device__ uint32_t ac_transition(const uint8_t* ac, uint32_t ac_len, uint32_t node_off, uint32_t ch) {
    if (node_off + 20 > ac_len) return 0xFFFFFFFFu;
    uint32_t w0 = ld32(ac + node_off);
    uint32_t kind = w0 & 0xFF;
    if (kind == 1) return ((w0 >> 8) & 0xFF) == ch ? ld32(ac + node_off + 12) : 0xFFFFFFFFu;
    if (kind == 2) {
        uint32_t eo = ld32(ac + node_off + 12), cnt = (w0 >> 16) & 0xFF;
        if (eo + cnt * 8 > ac_len) return 0xFFFFFFFFu;
        for (uint32_t i = 0; i < cnt; ++i) {
            uint32_t ec = ac[eo + i * 8];
            if (ec == ch) return ld32(ac + eo + i * 8 + 4);
            if (ec > ch) return 0xFFFFFFFFu;
        }
        return 0xFFFFFFFFu;
    }
    if (kind == 3) {
        uint32_t t = ld32(ac + node_off + 12) + ch * 4;
        if (t + 4 > ac_len) return 0xFFFFFFFFu;
        uint32_t target = ld32(ac + t);
        return target != 0 ? target : 0xFFFFFFFFu;
    }
    return 0xFFFFFFFFu;
}

__device__ __forceinline__ uint32_t utf8_adv(uint32_t c) { return c < 0x80 ? 1 : c < 0xE0 ? 2 : c < 0xF0 ? 3 : 4; }
__device__ __forceinline__ bool is_rust_char(uint32_t c) { return c < 0xD800 || (c > 0xDFFF && c <= 0x10FFFF); }

// Candidate text as the glob pass reads it: the first GLOB_WIN bytes sit in a per-lane LDS window (the star loop of a
// `*literal` pattern probes the text once per position; from the log each probe would be a dependent global load), the
// rest comes from the log.
constexpr uint32_t GLOB_WIN = 64, GLOB_WIN_WORDS = GLOB_WIN / 8 + 1;
constexpr uint32_t GLOB_OUTQ = 8;   // output states queued per text before they are handled (LDS, per lane)
struct TextView {
    const uint8_t* g;     // the text in the log
    uint32_t n;           // its length
    const uint64_t* w;    // LDS window, GLOB_WIN_WORDS words; bytes past the staged ones are zero
    uint32_t wn;          // bytes staged
    __device__ __forceinline__ uint32_t at(uint32_t i) const { return i < wn ? (uint32_t)(w[i >> 3] >> ((i & 7) * 8)) & 0xFF : g[i]; }
    // bytes i..i+7 little-endian; bytes at or past n are unspecified
    __device__ __forceinline__ uint64_t load8(uint32_t i) const {
        if (i < wn && (i + 8 <= wn || wn >= n)) {   // all 8 bytes staged, or nothing exists past the staged ones
            const uint64_t a = w[i >> 3], b = w[(i >> 3) + 1];
            const uint32_t sh = (i & 7) * 8;
            return sh ? (a >> sh) | (b << (64 - sh)) : a;
        }
        uint64_t v = 0;
        if (i + 8 <= n) __builtin_memcpy(&v, g + i, 8);
        else for (uint32_t b = 0; i + b < n; ++b) v |= (uint64_t)g[i + b] << (8 * b);
        return v;
    }
};
// stage the first bytes of text[0, n) (inside log[0, log_len)) into the lane's window
__device__ __forceinline__ TextView text_stage(const uint8_t* log, uint32_t log_len, uint32_t start, uint32_t n, uint64_t* win) {
    TextView tv{log + start, n, win, min(n, GLOB_WIN)};
#pragma unroll
    for (uint32_t k = 0; k < GLOB_WIN_WORDS; ++k) {
        uint64_t v = 0;
        const uint32_t o = k * 8;
        if (o < tv.wn) {
            if (start + o + 8 <= log_len) __builtin_memcpy(&v, log + start + o, 8);
            else for (uint32_t b = 0; start + o + b < log_len; ++b) v |= (uint64_t)log[start + o + b] << (8 * b);
            if (tv.wn - o < 8) v &= (1ull << ((tv.wn - o) * 8)) - 1;
        }
        win[k] = v;
    }
    return tv;
}

// Storage of the glob matcher. The streaming pass keeps the star stack and the result list of a candidate in a few dozen
// words per lane (StarStackFixed, GlobSetList); a candidate that needs more — a pattern with more nested '*' than
// MAX_GLOB_STARS, more than MAX_GLOB_RESULTS matching patterns — is handed to the spill pass (k_lookup_spill), which gives
// each lane a stack as deep as the longest pattern and one bit per pattern id in global memory (StarStackMem,
// GlobSetBitmap): Paraglob::find_all has no such limits (pg:1028-1182), so neither has the scan.
struct StarStackFixed {
    uint32_t seg[MAX_GLOB_STARS], pos[MAX_GLOB_STARS];
    __device__ __forceinline__ uint32_t cap() const { return MAX_GLOB_STARS; }
    __device__ __forceinline__ void set(uint32_t i, uint32_t s, uint32_t p) { seg[i] = s; pos[i] = p; }
    __device__ __forceinline__ void get(uint32_t i, uint32_t& s, uint32_t& p) const { s = seg[i]; p = pos[i]; }
};
struct StarStackMem {
    uint32_t* mem;   // 2 * n words
    uint32_t n;
    __device__ __forceinline__ uint32_t cap() const { return n; }
    __device__ __forceinline__ void set(uint32_t i, uint32_t s, uint32_t p) { mem[2 * i] = s; mem[2 * i + 1] = p; }
    __device__ __forceinline__ void get(uint32_t i, uint32_t& s, uint32_t& p) const { s = mem[2 * i]; p = mem[2 * i + 1]; }
};
struct GlobSetList {     // sorted unique ids, MAX_GLOB_RESULTS entries
    uint32_t* out;
    uint32_t n = 0;
    bool over = false;
    __device__ __forceinline__ bool contains(uint32_t id) const { for (uint32_t k = 0; k < n; ++k) if (out[k] == id) return true; return false; }
    __device__ __forceinline__ void insert(uint32_t id) {
        uint32_t k = 0;
        while (k < n && out[k] < id) ++k;
        if (k < n && out[k] == id) return;
        if (n >= MAX_GLOB_RESULTS) { over = true; return; }
        for (uint32_t m = n; m > k; --m) out[m] = out[m - 1];
        out[k] = id;
        ++n;
    }
};
struct GlobSetBitmap {   // one bit per pattern id (zeroed by the caller)
    uint32_t* bits;
    uint32_t n_ids;      // ids the bitmap covers
    uint32_t n = 0;
    bool over = false;
    __device__ __forceinline__ bool contains(uint32_t id) const { return id < n_ids && ((bits[id >> 5] >> (id & 31)) & 1u); }
    __device__ __forceinline__ void insert(uint32_t id) {
        if (id >= n_ids || contains(id)) return;
        bits[id >> 5] |= 1u << (id & 31);
        ++n;
    }
};

// match_glob_from_buffer / match_segments_impl (pg:1364-1639), case-sensitive. The recursion is replayed with an
// explicit stack of Star frames; every call of the reference consumes one unit of the 100 000-step budget here too.
// The header of the segment last looked at and the first 8 bytes of its literal stay in registers: a star re-enters the
// same segment once per text position.
template <class ST>
__device__ bool glob_match(const DevDb& db, uint32_t pattern_id, const TextView& text, ST& stk, bool& over) {
    const uint8_t* pg = db.pg;
    const bool ci = db.ci != 0;
    const uint32_t tn = text.n;
    uint32_t io = db.glob_seg_off + pattern_id * 8;
    if (io + 8 > db.pg_len) return false;
    uint32_t first = ld32(pg + io);
    uint32_t count = ld32(pg + io + 4) & 0xFFFF;
    uint32_t steps = 100000;
    uint32_t t_seg = 0, t_pos = 0;
    int sp = 0;
    uint32_t pos = 0, seg = 0;
    bool result = false;
    uint32_t c_seg = 0xFFFFFFFFu, c_h0 = 0, c_dlen = 0, c_doff = 0;
    uint64_t c_lit8 = 0;
    for (;;) {
        // ---- CALL(pos, seg)
        bool ret = false;
        if (steps == 0) return false;  // once exhausted every remaining call returns false (pg:1415-1417)
        --steps;
        if (seg >= count) { result = pos >= tn; ret = true; }
        else {
            uint32_t so = first + seg * 12;
            if (so + 12 > db.pg_len) { result = false; ret = true; }
            else {
                if (seg != c_seg) {
                    c_seg = seg;
                    c_h0 = ld32(pg + so); c_dlen = ld32(pg + so + 4); c_doff = ld32(pg + so + 8);
                    c_lit8 = 0;
                    if ((c_h0 & 0xFF) == 0 && (uint64_t)c_doff + c_dlen <= db.pg_len) {
                        if (c_dlen >= 8) __builtin_memcpy(&c_lit8, pg + c_doff, 8);
                        else for (uint32_t k = 0; k < c_dlen; ++k) c_lit8 |= (uint64_t)pg[c_doff + k] << (8 * k);
                        // case-insensitive (pg:1456-1478): characters compare with eq_ignore_ascii_case, which on UTF-8 bytes
                        // is ASCII folding of both sides (the bytes of other characters are >= 0x80 and must be equal)
                        if (ci) c_lit8 = ascii_lower8(c_lit8);
                    }
                }
                const uint32_t st = c_h0 & 0xFF, fl = (c_h0 >> 8) & 0xFF, dlen = c_dlen, doff = c_doff;
                if (st == 0) {
                    bool ok = (uint64_t)doff + dlen <= db.pg_len && tn - pos >= dlen;
                    if (ok && dlen) {
                        const uint64_t m0 = dlen >= 8 ? ~0ull : (1ull << (dlen * 8)) - 1;
                        const uint64_t t0 = text.load8(pos);
                        uint64_t diff = ((ci ? ascii_lower8(t0) : t0) ^ c_lit8) & m0;
                        if (diff == 0) {
                            for (uint32_t k = 8; k < dlen; k += 8) {   // no early exit: the loads are independent
                                const uint32_t r = dlen - k;
                                uint64_t x = 0;
                                if (r >= 8) __builtin_memcpy(&x, pg + doff + k, 8);
                                else for (uint32_t b = 0; b < r; ++b) x |= (uint64_t)pg[doff + k + b] << (8 * b);
                                const uint64_t tk = text.load8(pos + k);
                                diff |= ((ci ? ascii_lower8(tk) : tk) ^ (ci ? ascii_lower8(x) : x)) & (r >= 8 ? ~0ull : (1ull << (r * 8)) - 1);
                            }
                        }
                        ok = diff == 0;
                    }
                    if (ok) { pos += dlen; ++seg; } else { result = false; ret = true; }
                } else if (st == 1) {
                    if (seg + 1 >= count) { result = true; ret = true; }
                    else if (sp >= (int)stk.cap()) { over = true; return false; }   // deeper than this pass's stack: spill pass
                    else {
                        if (sp > 0) stk.set(sp - 1, t_seg, t_pos);   // the innermost frame lives in registers
                        t_seg = seg; t_pos = pos; ++sp; ++seg;
                    }
                } else if (st == 2) {
                    if (pos < tn) { pos += utf8_adv(text.at(pos)); ++seg; } else { result = false; ret = true; }
                } else if (st == 3) {
                    if (pos >= tn || (uint64_t)doff + dlen > db.pg_len) { result = false; ret = true; }
                    else {
                        uint32_t c = text.at(pos), adv = utf8_adv(c);
                        uint32_t cp = adv == 1 ? c : adv == 2 ? (c & 0x1F) : adv == 3 ? (c & 0x0F) : (c & 0x07);
                        for (uint32_t k = 1; k < adv && pos + k < tn; ++k) cp = (cp << 6) | (text.at(pos + k) & 0x3F);
                        if (ci) cp = ascii_lower1(cp);   // pg:1552-1555
                        bool in_class = false;
                        for (uint32_t k = 0; k < dlen / 12 && !in_class; ++k) {
                            const uint8_t* it = pg + doff + k * 12;
                            uint32_t ty = it[0], c1 = ld32(it + 4), c2 = ld32(it + 8);
                            if (ci) { c1 = ascii_lower1(c1); c2 = ascii_lower1(c2); }   // pg:1584-1607 (values that are no chars fail is_rust_char either way)
                            if (ty == 0) in_class = is_rust_char(c1) && cp == c1;
                            else if (ty == 1) in_class = is_rust_char(c1) && is_rust_char(c2) && cp >= c1 && cp <= c2;
                        }
                        if ((fl & 1) ? !in_class : in_class) { pos += adv; ++seg; } else { result = false; ret = true; }
                    }
                } else { result = false; ret = true; }
            }
        }
        if (!ret) continue;
        // ---- RETURN(result) to the innermost Star frame
        for (;;) {
            if (sp == 0) return result;
            uint32_t fp = t_pos;
            if (result || fp >= tn) {  // star returns true, or is exhausted and returns false: propagate
                --sp;
                if (sp > 0) stk.get(sp - 1, t_seg, t_pos);
                continue;
            }
            fp += utf8_adv(text.at(fp));
            t_pos = fp;
            pos = fp;
            seg = t_seg + 1;
            break;
        }
    }
}

// Paraglob::find_all (pg:1028-1182): collects the matching pattern ids in `rs`; rs.over is set when this pass's storage is too
// small for the candidate (it then goes to the spill pass and the partial result is dropped).
template <class RS, class ST>
__device__ void glob_find_all(const DevDb& db, const DfaView& dv, const TextView& text, uint32_t* oq, RS& rs, ST& stk) {
    const uint32_t tn = text.n;
    auto insert = [&](uint32_t id) { rs.insert(id); };
    auto contains = [&](uint32_t id) { return rs.contains(id); };
    auto consider = [&](uint32_t pid) {
        uint32_t eo = db.patterns_off + pid * 16;
        if (eo + 16 > db.pg_len) return;
        uint32_t entry_id = ld32(db.pg + eo);
        uint32_t ptype = db.pg[eo + 4];
        if (contains(entry_id)) return;
        if (ptype == 0 || glob_match(db, entry_id, text, stk, rs.over)) insert(entry_id);
    };
    for (uint32_t i = 0; i < db.wild_count; ++i) {
        uint32_t wo = db.wild_off + i * 8;
        if (wo + 8 > db.pg_len) continue;
        uint32_t pid = ld32(db.pg + wo);
        if (db.patterns_off + pid * 16 + 16 > db.pg_len) continue;
        if (!contains(pid) && glob_match(db, pid, text, stk, rs.over)) insert(pid);
    }
    if (db.ac_size > 0 && tn > 0) {
        const uint8_t* ac = db.pg + db.ac_start;
        // literals that end at node `cur` -> their patterns (run_ac_matching_into_static collects them per visited node)
        auto outputs = [&](uint32_t cur) {
            const uint32_t pc = ac[cur + 3];
            if (!pc) return;
            const uint32_t po = ld32(ac + cur + 16);
            if ((uint64_t)po + pc * 4 > db.ac_size) return;
            for (uint32_t k = 0; k < pc; ++k) {
                const uint32_t lit = ld32(ac + po + k * 4);
                if (lit >= db.n_ac_lits) continue;
                for (uint32_t q = db.lit2pat_off[lit]; q < db.lit2pat_off[lit + 1]; ++q) consider(db.lit2pat[q]);
            }
        };
        if (db.dfa) {
            // flattened automaton: one table load per byte, the text fetched 8 bytes at a time
            uint32_t st = 0, qn = 0;
            for (uint32_t i = 0; i < tn; i += 8) {
                uint64_t w = text.load8(i);
                const uint32_t m = min(8u, tn - i);
                for (uint32_t b = 0; b < m; ++b) {
                    const uint32_t e = dfa_step(db, dv, st, (uint32_t)w & 0xFF);
                    w >>= 8;
                    st = e & 0x7FFFFFFFu;
                    if (e >> 31) {
                        // Output states are queued and handled after the walk: lanes meet them at different text
                        // positions, and handling them on the spot would run each lane's chain of dependent loads
                        // (node -> literal -> patterns -> segments) one after the other instead of side by side.
                        if (qn < GLOB_OUTQ) oq[qn++] = st;
                        else outputs(db.dfa_node[st]);
                    }
                }
            }
            for (uint32_t k = 0; k < qn; ++k) outputs(db.dfa_node[oq[k]]);
        } else {
            uint32_t cur = 0;
            for (uint32_t i = 0; i < tn; ++i) {
                uint32_t ch = text.at(i);
                if (db.ci) ch = ascii_lower1(ch);   // pg:1198-1206
                for (;;) {
                    uint32_t nx = ac_transition(ac, db.ac_size, cur, ch);
                    if (nx != 0xFFFFFFFFu) { cur = nx; break; }
                    if (cur == 0) break;
                    if (cur + 20 > db.ac_size) break;
                    cur = ld32(ac + cur + 8);
                }
                if (cur + 20 > db.ac_size) continue;
                outputs(cur);
            }
        }
    }
}

// true when the text reaches a state of the flattened AC automaton that has output literals (necessary for any glob
// with a literal part to match)
__device__ __forceinline__ bool ac_touches_output(const DevDb& db, const DfaView& dv, const uint8_t* text, uint32_t tn) {
    uint32_t st = 0, any = 0;
    for (uint32_t i = 0; i < tn; i += 8) {
        uint64_t w = 0;
        const uint32_t m = min(8u, tn - i);
        if (m == 8) __builtin_memcpy(&w, text + i, 8);
        else for (uint32_t b = 0; b < m; ++b) w |= (uint64_t)text[i + b] << (8 * b);
        for (uint32_t b = 0; b < m; ++b) {
            const uint32_t e = dfa_step(db, dv, st, (uint32_t)w & 0xFF);
            w >>= 8;
            st = e & 0x7FFFFFFFu;
            any |= e;
        }
    }
    return (any >> 31) != 0;
}

// One lane's hit -> dense FinalHit record (+ its pattern ids and data offsets), device copy and pinned host mirror: the body of
// k_pack, also called straight from k_lookup for bulk scans (LookupParams::direct), where it overlaps the PCIe writes of
// the records with the lookups instead of running as a kernel of its own afterwards. Wave-uniform call; `valid` marks the
// lanes that hold a hit. Glob ids come from `globs` (k_lookup's own result list) or, when that is null, from pp.ids.
// Pattern results follow Database::lookup_string_uncached (database.rs:911-981): the literal id counts only if it has a data
// mapping, then the glob ids in ascending order; a literal without mapping and no glob is NotFound.
__device__ __forceinline__ void pack_record(const PackParams& pp, bool valid, const Hit& h, const uint32_t* globs) {
    const uint32_t lane = lane_id();
    uint32_t nid = 0, lit_off = 0xFFFFFFFFu;
    if (valid && h.kind == 3) {
        if (h.a != 0xFFFFFFFFu && h.a < pp.n_lit) lit_off = pp.lit_offsets[h.a];
        nid = (lit_off != 0xFFFFFFFFu ? 1u : 0u) + h.n_globs;
        if (nid == 0) valid = false;
    }
    // dense slot for the record: one atomic per wave
    const uint64_t vm = __ballot(valid);
    if (vm == 0) return;
    uint32_t slot0 = 0;
    if (lane == 0) slot0 = atomicAdd(&pp.counters->n_final, (uint32_t)__popcll(vm));
    slot0 = __builtin_amdgcn_readfirstlane(slot0);
    const uint32_t slot = slot0 + (uint32_t)__popcll(vm & lanemask_lt());
    // side-array space for pattern ids: wave exclusive scan of nid, one atomic per wave
    uint32_t scan = valid ? nid : 0u;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)scan, off);
        if ((int)lane >= off) scan += t;
    }
    const uint32_t total = (uint32_t)__shfl((int)scan, 63);
    uint32_t ids0 = 0;
    if (total) {
        if (lane == 0) ids0 = atomicAdd(&pp.counters->n_final_ids, total);
        ids0 = __builtin_amdgcn_readfirstlane(ids0);
    }
    if (valid) {
        const uint32_t my_ids = ids0 + scan - nid;
        FinalHit f{};
        f.start = h.start;
        f.len_type = h.len_type;
        f.kind = h.kind;
        f.prefix_len = h.prefix_len;
        if (h.kind == 2) f.value = h.a;
        else {
            f.n_ids = (uint16_t)nid;
            f.value = my_ids;
            uint32_t w = my_ids;
            if (lit_off != 0xFFFFFFFFu) {
                if (w < pp.out_ids_cap) { pp.out_ids[w] = h.a; pp.out_offs[w] = (long long)lit_off; }
                if (w < pp.host_ids_cap) { pp.host_ids[w] = h.a; pp.host_offs[w] = (long long)lit_off; }
                ++w;
            }
            for (uint32_t k = 0; k < h.n_globs; ++k, ++w) {
                const uint32_t pid = globs ? globs[k] : ((h.ids_off + k < pp.ids_cap) ? pp.ids[h.ids_off + k] : 0u);
                const long long go = pid < pp.n_glob ? (long long)pp.glob_offsets[pid] : -1ll;
                if (w < pp.out_ids_cap) { pp.out_ids[w] = pid; pp.out_offs[w] = go; }
                if (w < pp.host_ids_cap) { pp.host_ids[w] = pid; pp.host_offs[w] = go; }
            }
        }
        if (slot < pp.out_cap) pp.out[slot] = f;
        if (slot < pp.host_cap) pp.host_out[slot] = f;
    }
}

// Records without glob ids (IP hits, literal hits) collect in a per-lane LDS buffer and leave PEND_RECS per lane at a time:
// the dense slots of the record array come from ONE returning atomic per flush instead of one per loop iteration. With a
// database that nearly every candidate hits (CIDR-heavy: 17 M candidates, 11 M hits per pass) the per-iteration atomics on
// that one counter — served one after the other — were 2/3 of the lookup kernel.
constexpr uint32_t PEND_RECS = 8;
struct PendRec { uint32_t start, len_type, a, kp; };   // kp: kind | prefix_len << 8
// LDS budget of k_lookup<false> with an automaton: 32 KiB of pending records + 1 KiB work-list stage (static) + 32 KiB of
// transition rows (dynamic) — more than the 64 KiB a workgroup gets on earlier CDNA parts; gfx950 has 160 KiB per CU, and the
// launch wrappers check the launch status (check_launch) so that a rejected launch is an error, not a scan without hits.
static_assert(256 * PEND_RECS * sizeof(PendRec) + 4 * 64 * 4 + DFA_LDS_ENTRIES * 4 + 1024 <= 160 * 1024, "k_lookup<false>: LDS of one gfx950 CU");
// WG = true: the call is made by every thread of a 256-thread workgroup (the flush at the end of a kernel); the four waves then
// reserve their slots with ONE pair of atomics per workgroup through `wg` (12 words of LDS). All waves finish their lists at about
// the same time, and 2048 waves queueing on the one counter line for their last flush were half of the string-lookup pass.
template <uint32_t NREC = PEND_RECS, bool WG = false>
__device__ __forceinline__ void pack_pending(const PackParams& pp, uint32_t cnt, const PendRec* mine, uint32_t* wg = nullptr) {
    // literal hits count only if the literal has a data mapping (database.rs:911-981)
    // keep: records that leave as FinalHit, lit: those of them with a pattern id, c4: IPv4 results that leave as compact records
    uint32_t lit_off[NREC], keep = 0, lit = 0, c4 = 0;
    const bool compact = pp.c4_out != nullptr;
#pragma unroll
    for (uint32_t j = 0; j < NREC; ++j) {
        lit_off[j] = 0xFFFFFFFFu;
        if (j < cnt) {
            const PendRec r = mine[j];
            if ((r.kp & 0xFF) == 3) {
                if (r.a < pp.n_lit) lit_off[j] = pp.lit_offsets[r.a];
                if (lit_off[j] != 0xFFFFFFFFu) { keep |= 1u << j; lit |= 1u << j; }
            } else if (compact && (r.len_type >> 24) == IT_IPV4) c4 |= 1u << j;
            else keep |= 1u << j;
        }
    }
    // Slots are handed out RECORD-INDEX major: record j of every lane leaves with one store instruction, consecutive lanes to
    // consecutive slots — whole lines of the pinned host mirror per instruction instead of 64 scattered 16-byte pieces (the
    // mirror is written across PCIe: 5.2 -> 4.6 ms for the 11 M hits of a CIDR-heavy batch). Totals first (one ballot per
    // record index), then the same ballots give the ranks.
    uint32_t total = 0, total_ids = 0, total_c4 = 0;   // wave-uniform
#pragma unroll
    for (uint32_t j = 0; j < NREC; ++j) {
        total += (uint32_t)__popcll(__ballot((keep >> j) & 1u));
        total_ids += (uint32_t)__popcll(__ballot((lit >> j) & 1u));
        if (compact) total_c4 += (uint32_t)__popcll(__ballot((c4 >> j) & 1u));
    }
    uint32_t slot0 = 0, ids0 = 0, c40 = 0;
    if constexpr (WG) {
        const uint32_t wave = threadIdx.x >> 6;
        if (lane_id() == 0) { wg[wave] = total | (total_ids << 16); wg[8 + wave] = total_c4; }   // at most 64 * NREC of each per wave
        __syncthreads();
        uint32_t before = 0, before_ids = 0, before_c4 = 0, all = 0, all_ids = 0, all_c4 = 0;
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t t = wg[k], tc = wg[8 + k];
            all += t & 0xFFFFu; all_ids += t >> 16; all_c4 += tc;
            if (k < wave) { before += t & 0xFFFFu; before_ids += t >> 16; before_c4 += tc; }
        }
        if (threadIdx.x == 0) {
            if (all) wg[4] = atomicAdd(&pp.counters->n_final, all);
            if (all_ids) wg[5] = atomicAdd(&pp.counters->n_final_ids, all_ids);
            if (all_c4) wg[6] = atomicAdd(&pp.counters->n_c4, all_c4);
        }
        __syncthreads();
        if (total + total_c4 == 0) return;
        slot0 = wg[4] + before;
        ids0 = wg[5] + before_ids;
        c40 = wg[6] + before_c4;
    } else {
        if (total + total_c4 == 0) return;
        if (lane_id() == 0) {
            if (total) slot0 = atomicAdd(&pp.counters->n_final, total);
            if (total_ids) ids0 = atomicAdd(&pp.counters->n_final_ids, total_ids);
            if (total_c4) c40 = atomicAdd(&pp.counters->n_c4, total_c4);
        }
        slot0 = __builtin_amdgcn_readfirstlane(slot0);
        ids0 = __builtin_amdgcn_readfirstlane(ids0);
        c40 = __builtin_amdgcn_readfirstlane(c40);
    }
#pragma unroll
    for (uint32_t j = 0; j < NREC; ++j) {
        const bool k = (keep >> j) & 1u, l = (lit >> j) & 1u, c = (c4 >> j) & 1u;
        const uint64_t mk = __ballot(k), ml = __ballot(l), mc = compact ? __ballot(c) : 0ull;
        if (k) {
            const uint32_t slot = slot0 + mbcnt64(mk);
            const PendRec r = mine[j];
            FinalHit f{};
            f.start = r.start; f.len_type = r.len_type;
            f.kind = (uint8_t)(r.kp & 0xFF); f.prefix_len = (uint8_t)(r.kp >> 8);
            if (f.kind == 2) f.value = r.a;
            else {
                const uint32_t w = ids0 + mbcnt64(ml);
                f.n_ids = 1; f.value = w;
                if (w < pp.out_ids_cap) { pp.out_ids[w] = r.a; pp.out_offs[w] = (long long)lit_off[j]; }
                if (w < pp.host_ids_cap) { pp.host_ids[w] = r.a; pp.host_offs[w] = (long long)lit_off[j]; }
            }
            if (slot < pp.out_cap) pp.out[slot] = f;
            if (slot < pp.host_cap) pp.host_out[slot] = f;
        }
        if (c) {
            const uint32_t slot = c40 + mbcnt64(mc);
            const PendRec r = mine[j];
            const uint2 rec = c4_pack(r.start, r.len_type & 0xFFFFFFu, r.a, r.kp >> 8);
            if (slot < pp.c4_cap) pp.c4_out[slot] = rec;
            if (slot < pp.host_c4_cap) pp.host_c4[slot] = rec;
        }
        slot0 += (uint32_t)__popcll(mk);
        ids0 += (uint32_t)__popcll(ml);
        c40 += (uint32_t)__popcll(mc);
    }
}

// GLOB=false carries no glob state and stays register-lean: databases without a PARAGLOB section, and the first pass of
// the two-pass lookup (p.ac_filter): IP and literal lookups plus one DFA walk per string candidate; candidates that
// touch an AC output state are deferred to the GLOB=true pass through p.glob_work. GLOB=true does the full
// Paraglob::find_all, over all candidates or (p.from_work) over the work list.
template <bool GLOB>
__global__ __launch_bounds__(256) void k_lookup(LookupParams p, DevDb db) {
    __shared__ uint8_t cls[256];  // byte -> DFA class
    // the glob pass sees texts that go deep into the automaton anyway: a small shallow part, more resident waves
    constexpr uint32_t ROWS = GLOB ? DFA_LDS_ENTRIES_GLOB : DFA_LDS_ENTRIES;
    // the transition rows of the lean pass are dynamic LDS: a database without a glob section has no automaton, and without the
    // 32 KiB twice as many workgroups are resident (lookup_dyn_lds(); the string lookups are chains of dependent loads)
    extern __shared__ __attribute__((aligned(16))) uint32_t rows_dyn[];
    __shared__ uint32_t rows_glob[GLOB ? ROWS : 1];
    uint32_t* rows = GLOB ? rows_glob : rows_dyn;
    __shared__ uint64_t twin[GLOB ? 256 * GLOB_WIN_WORDS : 1];   // per-lane text window of the glob pass
    __shared__ uint32_t outq[GLOB ? 256 * GLOB_OUTQ : 1];
    const uint32_t n = p.from_work ? min(p.n_work ? *p.n_work : p.counters->n_glob_work, p.glob_work_cap) : min(p.n_in ? *p.n_in : p.counters->n_cand, p.cand_cap);
    uint32_t stride = gridDim.x * blockDim.x;
    // Lean pass, string candidates that come without a verdict (k_validate_dom flags its own: what is left are the long tokens, e-mail
    // addresses and undecided domains of a batch — some ten thousand in a log scan): the automaton walk that decides whether a candidate
    // needs the glob pass is a chain of dependent row loads as long as the text — a 64-byte hash: 64 round trips — and the glob pass walks
    // the text again anyway. While there are few of them they go to the glob pass unwalked; when the anchor lists they come from are long
    // (hash-dense input) the walk here, in the cheaper kernel, thins them out first.
    const bool defer_unwalked = !GLOB && p.ac_filter && p.counters->n_tok + p.counters->n_rare + p.counters->n_rare_dom <= 262144u;
    // ... and a lean pass that walks nothing does not stage the automaton's rows either (32 KiB per workgroup from HBM: most of what such a pass costs)
    DfaView dv{cls, rows, 0};
    if (GLOB || !defer_unwalked) dv = dfa_stage<ROWS>(db, cls, rows);
    __syncthreads();
    ChunkWriter<Hit, HIT_CHUNK> cw;
    __shared__ uint32_t wb_work[4][64];
    BufferedWriter<uint32_t> ww(wb_work[threadIdx.x >> 6]);   // glob work list: sparse, dense output
    __shared__ __attribute__((aligned(16))) PendRec pend_lds[GLOB ? 1 : 256 * PEND_RECS];   // per-lane pending records (pack_pending)
    PendRec* pend = pend_lds + (GLOB ? 0 : threadIdx.x * PEND_RECS);
    uint32_t pn = 0;
    Hit SH{};
    SH.kind = 0xFF;
    // loop bound is wave-uniform so that the chunk writers see converged waves
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += stride) {
        uint32_t i = base + threadIdx.x;
        Hit h{};
        bool emit = false, defer = false, spill = false;
        uint32_t globs[GLOB ? MAX_GLOB_RESULTS : 1];
        uint32_t ng = 0;
        Candidate c{0, 0xFFFFFFFFu, 0, 0};
        if (p.from_work) {
            const uint32_t idx = i < n ? p.glob_work[i] : 0xFFFFFFFFu;
            if (idx != 0xFFFFFFFFu && idx < p.cand_cap) { c = p.cands[idx]; i = idx; }
        } else if (i < n) {
            c = p.cands[i];
        }
        if (c.len_type != 0xFFFFFFFFu) {
            uint32_t type = c.len_type >> 24, tl = c.len_type & 0xFFFFFF;
            const uint8_t* text = p.log + c.start;
            h.cand = i; h.start = c.start; h.len_type = c.len_type;
            if (type == IT_IPV4) {
                uint32_t off, pfx;
                if (db.has_ip && trie_v4(db, c.v4, off, pfx)) { h.kind = 2; h.a = off; h.prefix_len = (uint8_t)pfx; emit = true; }
            } else if (type == IT_IPV6) {
                uint16_t seg[8];
                uint32_t off, pfx;
                if (db.has_ip && d_parse_ipv6(text, tl, seg) && trie_v6(db, seg, off, pfx)) { h.kind = 2; h.a = off; h.prefix_len = (uint8_t)pfx; emit = true; }
            } else {
                // producers that know already (k_validate_dom's suffix filter) say so in the candidate: no automaton walk then
                // a token of letters and digits in a database whose automaton has no literal made of them: the literal table alone can hold it
                const bool alnum_tok = type == IT_MD5 || type == IT_SHA1 || type == IT_SHA256 || type == IT_SHA384 || type == IT_SHA512 ||
                                       type == IT_BITCOIN || type == IT_ETHEREUM || type == IT_MONERO;
                if (!db.ac_alnum && alnum_tok) c.pad = CAND_NO_GLOB;
                if (!GLOB && p.ac_filter && p.early_glob && c.pad == CAND_GLOB) {
                    // queued for the glob pass by its producer (TokParams::glob_work_d): that pass runs beside this one
                } else if (!GLOB && p.ac_filter && (c.pad == CAND_GLOB || (c.pad != CAND_NO_GLOB && (defer_unwalked || ac_touches_output(db, dv, text, tl))))) defer = true;
                else {
                    uint32_t pid = 0xFFFFFFFFu;
                    if (db.has_literal) { uint32_t q; if (lit_lookup(db, text, tl, q)) pid = q; }
                    if constexpr (GLOB) {
                        GlobSetList rs{globs};
                        StarStackFixed stk;
                        glob_find_all(db, dv, text_stage(p.log, p.len, c.start, tl, twin + threadIdx.x * GLOB_WIN_WORDS), outq + threadIdx.x * GLOB_OUTQ, rs, stk);
                        ng = rs.n;
                        spill = rs.over;   // more results / deeper star nesting than this pass holds: the spill pass answers
                    }
                    if (!spill && (pid != 0xFFFFFFFFu || ng)) { h.kind = 3; h.a = pid; h.n_globs = (uint16_t)ng; emit = true; }
                }
            }
        }
        if (p.direct && !GLOB) {
            // bulk scans, records without glob ids: into the lane's pending buffer; all lanes flush when one buffer is full
            if (emit) { pend[pn] = PendRec{h.start, h.len_type, h.a, (uint32_t)h.kind | ((uint32_t)h.prefix_len << 8)}; ++pn; }
            if (__ballot(pn == PEND_RECS)) { pack_pending(p.pk, pn, pend); pn = 0; }
        } else if (p.direct) {
            // bulk scans: the record goes out right here (device copy + pinned host mirror), no hit list, no k_pack
            pack_record(p.pk, emit, h, GLOB ? globs : nullptr);
        } else {
            if (GLOB && emit && ng) {
                uint32_t io = atomicAdd(&p.counters->n_ids, ng);
                h.ids_off = io;
                for (uint32_t k = 0; k < ng; ++k) if (io + k < p.ids_cap) p.ids[io + k] = globs[k];
            }
            cw.append(emit, h, p.hits, p.hit_cap, &p.counters->n_hits, SH);
        }
        if (!GLOB && p.ac_filter) ww.append(defer, i, p.glob_work, p.glob_work_cap, &p.counters->n_glob_work);
        if (GLOB && spill) {   // rare: one atomic per candidate
            const uint32_t q = atomicAdd(&p.counters->n_spill, 1u);
            if (q < p.spill_cap) p.spill[q] = i | (p.spill_tag << 31);
        }
    }
    __shared__ uint32_t wg_slots[12];
    if (p.direct && !GLOB) pack_pending<PEND_RECS, true>(p.pk, pn, pend, wg_slots);
    cw.pad_rest(p.hits, p.hit_cap, SH);
    if (!GLOB && p.ac_filter) ww.flush(p.glob_work, p.glob_work_cap, &p.counters->n_glob_work);
    if (lane_id() == 0 && cw.total) atomicAdd(&p.counters->hits_true, cw.total);
    if (p.arrive_chain) {
        // last kernel of a side-stream chain: everything this workgroup wrote (records in pinned host memory included) is visible
        // before its arrival counts; the workgroup that completes the grid reports the chain
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            if (atomicAdd(&p.arrive->arrive_wgs[p.arrive_chain - 1], 1u) == gridDim.x - 1) atomicAdd(&p.arrive->chains_done, 1u);
        }
    }
}

// k_lookup_spill — the candidates the glob pass could not hold (GlobSetList / StarStackFixed overflow): one lane per candidate,
// result set = one bit per pattern id, star stack as deep as the longest pattern, both in global scratch owned by the thread
// (LookupParams::spill_scratch). The ids leave in ascending order through the id list, so the record has the same form as
// one written by k_lookup.
constexpr int SPILL_THREADS = 64;
__global__ __launch_bounds__(SPILL_THREADS) void k_lookup_spill(LookupParams p, DevDb db) {
    __shared__ uint8_t cls[256];
    __shared__ uint32_t rows[DFA_LDS_ENTRIES_GLOB];
    __shared__ uint64_t twin[SPILL_THREADS * GLOB_WIN_WORDS];
    __shared__ uint32_t outq[SPILL_THREADS * GLOB_OUTQ];
    const DfaView dv = dfa_stage<DFA_LDS_ENTRIES_GLOB>(db, cls, rows);
    __syncthreads();
    const uint32_t n = min(p.counters->n_spill, p.spill_cap);
    const uint32_t stride = gridDim.x * blockDim.x, tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t bit_words = (db.pattern_count + 31) / 32;
    uint32_t* bits = p.spill_scratch + (size_t)tid * p.spill_words;
    ChunkWriter<Hit, HIT_CHUNK> cw;
    Hit SH{};
    SH.kind = 0xFF;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += stride) {
        const uint32_t k = base + threadIdx.x;
        Hit h{};
        bool emit = false;
        if (k < n) {
            const uint32_t tagged = p.spill[k], i = tagged & 0x7FFFFFFFu;
            const bool alt = (tagged >> 31) != 0;   // a candidate of the undecided domains' own list (LookupParams::cands_alt)
            const Candidate c = alt ? (p.cands_alt && i < p.cand_alt_cap ? p.cands_alt[i] : Candidate{0, 0xFFFFFFFFu, 0, 0})
                                    : (i < p.cand_cap ? p.cands[i] : Candidate{0, 0xFFFFFFFFu, 0, 0});
            if (c.len_type != 0xFFFFFFFFu) {
                const uint32_t tl = c.len_type & 0xFFFFFF;
                h.cand = i; h.start = c.start; h.len_type = c.len_type;
                uint32_t pid = 0xFFFFFFFFu;
                if (db.has_literal) { uint32_t q; if (lit_lookup(db, p.log + c.start, tl, q)) pid = q; }
                for (uint32_t w = 0; w < bit_words; ++w) bits[w] = 0;
                GlobSetBitmap rs{bits, db.pattern_count};
                StarStackMem stk{bits + bit_words, (p.spill_words - bit_words) / 2};
                glob_find_all(db, dv, text_stage(p.log, p.len, c.start, tl, twin + threadIdx.x * GLOB_WIN_WORDS), outq + threadIdx.x * GLOB_OUTQ, rs, stk);
                uint32_t ng = rs.n;
                if (rs.over || ng > 0xFFFFu) { atomicOr(&p.counters->error, 1u); ng = 0; }   // the record counts ids in 16 bits
                if (ng) {
                    const uint32_t io = atomicAdd(&p.counters->n_ids, ng);
                    h.ids_off = io;
                    uint32_t o = io;
                    for (uint32_t w = 0; w < bit_words; ++w)
                        for (uint32_t m = bits[w]; m; m &= m - 1, ++o)
                            if (o < p.ids_cap) p.ids[o] = 32 * w + (uint32_t)__builtin_ctz(m);
                }
                if (pid != 0xFFFFFFFFu || ng) { h.kind = 3; h.a = pid; h.n_globs = (uint16_t)ng; emit = true; }
            }
        }
        if (p.direct) pack_record(p.pk, emit, h, nullptr);   // ids from the id list (h.ids_off)
        else cw.append(emit, h, p.hits, p.hit_cap, &p.counters->n_hits, SH);
    }
    cw.pad_rest(p.hits, p.hit_cap, SH);
    if (lane_id() == 0 && cw.total) atomicAdd(&p.counters->hits_true, cw.total);
}

// ------------------------------------------------------------------------------------------------ launch wrapper
// IPv4 candidates of k_anchor (LookupParams::cands / n_in name that list): trie lookups only. A kernel of its own because it
// runs on a second stream BESIDE the validation kernels (engine.cpp) and must leave them their LDS and registers: no automaton
// tables, NREC pending records per lane (2 -> 8 KiB per workgroup when the /24 bitmap has thinned the list, 8 when nearly every
// candidate hits and the slot atomics would otherwise queue up: see pack_pending).
template <uint32_t NREC>
__global__ __launch_bounds__(256) void k_lookup_ip(LookupParams p, DevDb db) {
    __shared__ __attribute__((aligned(16))) PendRec pend_lds[256 * NREC];
    PendRec* pend = pend_lds + threadIdx.x * NREC;
    uint32_t pn = 0;
    const uint32_t n = min(*p.n_in, p.cand_cap);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += stride) {   // wave-uniform bound (pack_pending is a wave operation)
        const uint32_t i = base + threadIdx.x;
        Candidate c{0, 0xFFFFFFFFu, 0, 0};
        if (i < n) c = p.cands[i];
        if (c.len_type != 0xFFFFFFFFu && (c.len_type >> 24) == IT_IPV4) {
            uint32_t off, pfx;
            if (db.has_ip && trie_v4(db, c.v4, off, pfx)) { pend[pn] = PendRec{c.start, c.len_type, off, 2u | (pfx << 8)}; ++pn; }
        }
        if (__ballot(pn == NREC)) { pack_pending<NREC>(p.pk, pn, pend); pn = 0; }
    }
    __shared__ uint32_t wg_slots[12];
    pack_pending<NREC, true>(p.pk, pn, pend, wg_slots);
}
void launch_lookup_ip(const LookupParams& p, const DevDb& db, int grid, bool dense, hipStream_t stream) {
    if (dense) hipLaunchKernelGGL(k_lookup_ip<8>, dim3(grid), dim3(256), 0, stream, p, db);
    else hipLaunchKernelGGL(k_lookup_ip<2>, dim3(grid), dim3(256), 0, stream, p, db);
    check_launch("launch_lookup_ip");
}
void launch_lookup(const LookupParams& p_in, const DevDb& db, int grid, hipStream_t stream) {
    LookupParams p = p_in;
    p.ac_filter = 0; p.from_work = 0;
    const size_t lean_lds = db.dfa ? (size_t)DFA_LDS_ENTRIES * 4 : 0;   // rows_dyn of k_lookup<false>
    if (!db.has_glob) {
        hipLaunchKernelGGL(k_lookup<false>, dim3(grid), dim3(256), lean_lds, stream, p, db);
    } else if (db.dfa && db.wild_count == 0 && p.glob_work && grid > 1) {
        // two passes: lean lookup + AC prefilter for everything, the register-heavy glob matcher only for the few
        // candidates that reach an AC output state (without literal hits no glob can match: pure wildcards aside)
        p.ac_filter = 1;
        const uint32_t arrive_chain = p.arrive_chain;
        p.arrive_chain = 0;   // the glob pass below is the last kernel of the chain
        hipLaunchKernelGGL(k_lookup<false>, dim3(grid), dim3(256), lean_lds, stream, p, db);
        p.arrive_chain = arrive_chain;
        p.ac_filter = 0; p.from_work = 1;
        hipLaunchKernelGGL(k_lookup<true>, dim3(grid), dim3(256), 0, stream, p, db);
    } else {
        hipLaunchKernelGGL(k_lookup<true>, dim3(grid), dim3(256), 0, stream, p, db);
    }
    // candidates beyond the glob pass's per-lane storage are listed in p.spill: Scanner::fetch launches k_lookup_spill when the
    // counters show that there are any (normally none)
    check_launch("launch_lookup");
}
// The glob pass over the work list k_validate_dom filled itself (LookupParams::glob_work / n_work point at it): launched when that
// kernel ends, beside the lean pass over the other candidates (launch_lookup with early_glob = 1)
void launch_lookup_early_glob(const LookupParams& p_in, const DevDb& db, int grid, hipStream_t stream) {
    LookupParams p = p_in;
    p.ac_filter = 0; p.from_work = 1;
    hipLaunchKernelGGL(k_lookup<true>, dim3(grid), dim3(256), 0, stream, p, db);
    check_launch("launch_lookup_early_glob");
}
// see launch_finish (scan_types.h)
__global__ __launch_bounds__(256) void k_finish(ScanCounters* dev, ScanCounters* host, uint32_t n_words, uint32_t expect_chains, unsigned long long poll_ticks,
                                                uint32_t next_n_dom) {
    uint32_t* d = reinterpret_cast<uint32_t*>(dev);
    uint32_t* h = reinterpret_cast<uint32_t*>(host);
    __shared__ uint32_t gave_up;
    if (threadIdx.x == 0) gave_up = 0;
    if (expect_chains) {
        // the side-stream chains of this scan report their ends in chains_done: poll it (one lane; the others wait at the barrier).
        // Bounded: after ~2 s of wall clock (100 MHz counter) the poll gives up and hands the join back to the host — a DIAGNOSTIC, not a
        // result: the counters are copied out as they stand but NOT cleared (the chains are still appending to the lists they describe), the
        // host copy carries error bit 8, and Scanner::fetch then waits for the side streams itself and runs k_finish again without a poll.
        // A slow but legitimate chain (the longest seen: 0.7 s, a glob pass over a few hundred adversarial names whose patterns all run
        // into the reference's 100 000-step budget; a large adversarial batch can take longer) gets slow, as in the reference, never wrong.
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            if (poll_ticks == 0) gave_up = 1;   // tests: the give-up path whatever the chains are doing
            else while (__hip_atomic_load(&dev->chains_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < expect_chains) {
                if (wall_clock64() - t0 > poll_ticks) { gave_up = 1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
        __threadfence();
    } else {
        __syncthreads();
    }
    const bool keep = gave_up != 0;
    for (uint32_t i = threadIdx.x; i < n_words; i += blockDim.x) {
        h[i] = d[i];
        if (!keep) d[i] = 0;
    }
    __syncthreads();
    // the next scan's k_anchor starts with one chunk of the domain list per wave already handed out (TokParams::dom_static)
    if (!keep && threadIdx.x == 0 && next_n_dom) dev->n_dom = next_n_dom;
    if (keep && threadIdx.x == 0) host->error |= 8u;
    __threadfence_system();
}
void launch_finish(ScanCounters* dev, ScanCounters* host_pinned, int n_blocks, uint32_t expect_chains, uint32_t next_n_dom, hipStream_t stream) {
    // how long k_finish polls for the side chains before it hands the join back to the host (100 MHz ticks; MATCHY_AMD_FINISH_POLL_US for tests)
    static const unsigned long long poll_ticks = getenv("MATCHY_AMD_FINISH_POLL_US") ? strtoull(getenv("MATCHY_AMD_FINISH_POLL_US"), nullptr, 10) * 100ull : 200000000ull;
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, stream, dev, host_pinned, (uint32_t)(n_blocks * sizeof(ScanCounters) / 4), expect_chains, poll_ticks, next_n_dom);
    check_launch("launch_finish");
}
void launch_lookup_spill(const LookupParams& p, const DevDb& db, hipStream_t stream) {
    hipLaunchKernelGGL(k_lookup_spill, dim3(p.spill_blocks), dim3(SPILL_THREADS), 0, stream, p, db);
    check_launch("launch_lookup_spill");
}
uint32_t spill_threads() { return SPILL_THREADS; }

}  // namespace mxy
