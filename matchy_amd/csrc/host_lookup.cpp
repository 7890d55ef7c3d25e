// Single-query lookup on the host: see host_lookup.h. Reads the sections where DbImage found them; every offset that is followed
// is bounds-checked here too (queries may run against files that only passed open()'s structural checks).
#include "host_lookup.h"

#include <algorithm>
#include <cstring>

#include "hashes.h"
#include "unicode_lower.h"

namespace mxy {
namespace {

inline uint32_t le32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t le64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

// ---- IP tree (tree.rs:127-245): records are big-endian, 24 / 28 / 32 bits
inline uint32_t tree_record(const DbImage& img, uint32_t node, uint32_t side) {
    const uint8_t* t = img.bytes.data();
    if (img.record_size == 24) {
        const uint8_t* b = t + (size_t)node * 6 + side * 3;
        return ((uint32_t)b[0] << 16) | ((uint32_t)b[1] << 8) | b[2];
    }
    if (img.record_size == 28) {
        const uint8_t* b = t + (size_t)node * 7;
        return side == 0 ? ((uint32_t)(b[3] >> 4) << 24) | ((uint32_t)b[0] << 16) | ((uint32_t)b[1] << 8) | b[2]
                         : ((uint32_t)(b[3] & 0xF) << 24) | ((uint32_t)b[4] << 16) | ((uint32_t)b[5] << 8) | b[6];
    }
    const uint8_t* b = t + (size_t)node * 8 + side * 4;
    return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
}

// walks `nbits` address bits (most significant first) from `node`; depth0 = levels already behind the start node.
// prefix_len = depth at which the data record was met + 1 (tree.rs:76-84, 113-119), data offset = record - node_count - 16.
bool tree_walk(const DbImage& img, uint32_t node, const uint8_t* addr, uint32_t nbits, uint32_t depth0, HostHit& out) {
    uint32_t depth = depth0;
    for (uint32_t i = 0; i < nbits; ++i) {
        const uint32_t bit = (addr[i >> 3] >> (7 - (i & 7))) & 1u;
        const uint32_t rec = tree_record(img, node, bit);
        if (rec == img.node_count) return false;
        if (rec < img.node_count) { node = rec; ++depth; continue; }
        if (rec < img.node_count + 16) return false;   // calculate_data_offset would fail (record inside the separator): an error, i.e. no answer
        out.kind = 2;
        out.a = rec - img.node_count - 16;
        out.prefix_len = (uint8_t)(depth + 1 - depth0);
        return true;
    }
    return false;
}

// ---- literal hash (matchy-literal-hash/src/lib.rs:467-543)
bool literal_lookup(const DbImage& img, const std::string& query, uint32_t& pattern_id) {
    if (!img.has_literal || img.lh_num_shards == 0) return false;
    const std::string* key = &query;
    std::string lowered;
    if (img.match_mode == 1) { lowered = LowerTable::get().to_lowercase(query); key = &lowered; }
    const uint64_t h = xxh64<false>((const uint8_t*)key->data(), key->size(), 0);
    const uint8_t* lh = img.bytes.data() + img.lh_off;
    const uint32_t shard = (uint32_t)(h % img.lh_num_shards);
    const uint32_t s0 = le32(lh + 32 + (size_t)shard * 4), s1 = le32(lh + 32 + (size_t)(shard + 1) * 4);
    if (s1 <= s0 || s1 > img.lh_table_size) return false;
    const uint32_t cap = s1 - s0, mask = cap - 1;
    uint32_t slot = s0 + ((uint32_t)h & mask);
    for (uint32_t n = 0; n < cap; ++n) {
        const uint8_t* e = lh + img.lh_table_start + (size_t)slot * 16;
        const uint32_t so = le32(e + 8);
        if (so == 0xFFFFFFFFu) return false;
        if (le64(e) == h && (uint64_t)so + 2 <= img.lh_strings_size) {
            const uint8_t* sp = lh + img.lh_strings_offset + so;
            const uint32_t sl = (uint32_t)sp[0] | ((uint32_t)sp[1] << 8);
            if ((uint64_t)so + 2 + sl <= img.lh_strings_size && sl == key->size() && memcmp(sp + 2, key->data(), sl) == 0) {
                pattern_id = le32(e + 12);
                return true;
            }
        }
        slot = s0 + ((slot + 1 - s0) & mask);
    }
    return false;
}

// ---- paraglob
struct Pg {
    const uint8_t* b;      // PARAGLOB buffer
    size_t n;
    const uint8_t* ac;     // Aho-Corasick section
    size_t ac_n;
    bool ci;
};

inline size_t utf8_len(uint8_t lead) { return lead < 0x80 ? 1 : lead < 0xE0 ? 2 : lead < 0xF0 ? 3 : 4; }
inline uint32_t utf8_cp(const uint8_t* p, size_t len) {
    if (len == 1) return p[0];
    if (len == 2) return ((uint32_t)(p[0] & 0x1F) << 6) | (p[1] & 0x3F);
    if (len == 3) return ((uint32_t)(p[0] & 0x0F) << 12) | ((uint32_t)(p[1] & 0x3F) << 6) | (p[2] & 0x3F);
    return ((uint32_t)(p[0] & 0x07) << 18) | ((uint32_t)(p[1] & 0x3F) << 12) | ((uint32_t)(p[2] & 0x3F) << 6) | (p[3] & 0x3F);
}
inline bool is_scalar(uint32_t c) { return c < 0xD800 || (c > 0xDFFF && c <= 0x10FFFF); }   // char::from_u32
inline uint32_t lower_ascii(uint32_t c) { return c - 'A' < 26u ? c + 32 : c; }
bool valid_utf8(const uint8_t* s, size_t n) {
    size_t i = 0;
    while (i < n) {
        const uint8_t c = s[i];
        if (c < 0x80) { ++i; continue; }
        size_t len; uint32_t minv;
        if (c >= 0xC2 && c <= 0xDF) { len = 2; minv = 0x80; }
        else if (c >= 0xE0 && c <= 0xEF) { len = 3; minv = 0x800; }
        else if (c >= 0xF0 && c <= 0xF4) { len = 4; minv = 0x10000; }
        else return false;
        if (i + len > n) return false;
        for (size_t k = 1; k < len; ++k) if ((s[i + k] & 0xC0) != 0x80) return false;
        const uint32_t cp = utf8_cp(s + i, len);
        if (cp < minv || !is_scalar(cp)) return false;
        i += len;
    }
    return true;
}

// find_ac_transition (paraglob_offset.rs:1271-1353); `next` only valid when true is returned
bool ac_transition(const Pg& g, size_t node, uint8_t ch, size_t& next) {
    if (node + 20 > g.ac_n) return false;
    const uint8_t* nd = g.ac + node;
    const uint32_t edges = le32(nd + 12);
    switch (nd[0]) {
        case 1:   // ONE
            if (nd[1] != ch) return false;
            next = edges;
            return true;
        case 2: { // SPARSE: sorted 8-byte edges
            const size_t count = nd[2];
            if ((size_t)edges + count * 8 > g.ac_n) return false;
            for (size_t i = 0; i < count; ++i) {
                const uint8_t* e = g.ac + edges + i * 8;
                if (e[0] == ch) { next = le32(e + 4); return true; }
                if (e[0] > ch) return false;
            }
            return false;
        }
        case 3: { // DENSE: 256 targets, 0 = none
            const size_t at = (size_t)edges + (size_t)ch * 4;
            if (at + 4 > g.ac_n) return false;
            const uint32_t t = le32(g.ac + at);
            if (!t) return false;
            next = t;
            return true;
        }
        default: return false;   // EMPTY, unknown kind
    }
}

// run_ac_matching_into_static (:1186-1266): literal ids at every state the text visits (a set)
void ac_literals(const Pg& g, const std::string& text, std::vector<uint32_t>& lits) {
    if (!g.ac_n || text.empty()) return;
    size_t cur = 0;
    for (const char tc : text) {
        uint8_t ch = (uint8_t)tc;
        if (g.ci && ch - 'A' < 26u) ch += 32;   // ASCII lower-casing of the text bytes
        // (a well-formed automaton's failure chain gets one level shallower per step; the bound keeps a file whose links form a cycle from
        // hanging the query — the reference would spin on it)
        for (size_t hops = 0, max_hops = g.ac_n / 20 + 1;; ++hops) {
            size_t nx;
            if (ac_transition(g, cur, ch, nx)) { cur = nx; break; }
            if (cur == 0) break;
            if (cur + 20 > g.ac_n) break;
            if (hops > max_hops) { cur = 0; break; }
            cur = le32(g.ac + cur + 8);   // failure link, then try again
        }
        if (cur + 20 > g.ac_n) continue;
        const uint8_t* nd = g.ac + cur;
        const size_t pc = nd[3];
        if (pc) {
            const size_t po = le32(nd + 16);
            if (po + pc * 4 <= g.ac_n)
                for (size_t i = 0; i < pc; ++i) lits.push_back(le32(g.ac + po + i * 4));
        }
    }
    std::sort(lits.begin(), lits.end());
    lits.erase(std::unique(lits.begin(), lits.end()), lits.end());
}

// match_segments_impl (:1402-1639). 1 = match, 0 = no match, -1 = error (aborts the whole match: the caller's `if let Ok(true)`).
// The reference recurses once per segment; here only a star recurses (once per position it tries) and the segments between stars are a loop —
// the same steps in the same order, one budget step per segment visited, but a stack as deep as the pattern has stars, not as long as it is.
int match_segments(const Pg& g, const std::string& text, size_t first_seg, size_t seg_count, size_t pos, size_t seg, size_t& steps) {
    const uint8_t* t = (const uint8_t*)text.data();
    const size_t tn = text.size();
    for (;;) {
        if (steps == 0) return 0;
        --steps;
        if (seg >= seg_count) return pos >= tn ? 1 : 0;
        const size_t so = first_seg + seg * 12;
        if (so + 12 > g.n) return 0;
        const uint8_t* sh = g.b + so;
        const uint32_t data_len = le32(sh + 4), data_off = le32(sh + 8);
        switch (sh[0]) {
            case 0: {   // literal
                if ((size_t)data_off + data_len > g.n) return 0;
                const uint8_t* lit = g.b + data_off;
                if (!valid_utf8(lit, data_len)) return -1;
                if (!g.ci) {
                    if (tn - pos < data_len || memcmp(t + pos, lit, data_len) != 0) return 0;
                    pos += data_len;
                } else {
                    // character by character, ASCII case folded; a text that ends early does not match
                    size_t tp = pos, lp = 0;
                    while (tp < tn && lp < data_len) {
                        const size_t tl = utf8_len(t[tp]), ll = utf8_len(lit[lp]);
                        const uint32_t tc = utf8_cp(t + tp, tl), lc = utf8_cp(lit + lp, ll);
                        if (lower_ascii(tc) != lower_ascii(lc)) return 0;
                        tp += tl; lp += ll;
                    }
                    if (lp < data_len) return 0;
                    pos = tp;
                }
                ++seg;
                continue;
            }
            case 1: {   // star
                if (seg + 1 >= seg_count) return 1;
                size_t p = pos;
                for (;;) {
                    const int r = match_segments(g, text, first_seg, seg_count, p, seg + 1, steps);
                    if (r != 0) return r;
                    if (p >= tn) break;
                    p += utf8_len(t[p]);
                }
                return 0;
            }
            case 2:     // question mark: one character
                if (pos >= tn) return 0;
                pos += utf8_len(t[pos]);
                ++seg;
                continue;
            case 3: {   // character class
                if (pos >= tn) return 0;
                const size_t cl = utf8_len(t[pos]);
                uint32_t ch = utf8_cp(t + pos, cl);
                if (g.ci) ch = lower_ascii(ch);
                if ((size_t)data_off + data_len > g.n) return 0;
                const size_t items = data_len / 12;
                bool in_class = false;
                for (size_t i = 0; i < items && !in_class; ++i) {
                    const uint8_t* it = g.b + data_off + i * 12;
                    uint32_t c1 = le32(it + 4), c2 = le32(it + 8);
                    if (it[0] == 0) {
                        if (!is_scalar(c1)) continue;
                        if (g.ci) c1 = lower_ascii(c1);
                        in_class = ch == c1;
                    } else if (it[0] == 1) {
                        if (!is_scalar(c1) || !is_scalar(c2)) continue;
                        if (g.ci) { c1 = lower_ascii(c1); c2 = lower_ascii(c2); }
                        in_class = ch >= c1 && ch <= c2;
                    }
                }
                if (((sh[1] & 1) != 0) == in_class) return 0;
                pos += cl;
                ++seg;
                continue;
            }
            default: return 0;
        }
    }
}

// match_glob_from_buffer (:1364-1398)
bool glob_matches(const Pg& g, uint32_t pid, const std::string& text, size_t gso) {
    const size_t io = gso + (size_t)pid * 8;
    if (io + 8 > g.n) return false;
    const size_t first = le32(g.b + io), count = (size_t)g.b[io + 4] | ((size_t)g.b[io + 5] << 8);
    size_t steps = 100000;
    return match_segments(g, text, first, count, 0, 0, steps) == 1;
}

// Paraglob::find_all (:1028-1182)
void find_all(const DbImage& img, const HostTables& tb, const std::string& text, std::vector<uint32_t>& result) {
    Pg g;
    g.b = img.bytes.data() + img.pg_off;
    g.n = img.pg_len;
    if (g.n < 112) return;
    const size_t ac_start = le32(g.b + 20), ac_size = le32(g.b + 24);
    g.ac = g.b + ac_start;
    g.ac_n = ac_start + ac_size <= g.n ? ac_size : 0;
    g.ci = img.match_mode == 1;
    const size_t patterns_off = le32(g.b + 36), gso = le32(g.b + 104);
    std::vector<uint32_t> lits;
    ac_literals(g, text, lits);
    // pure wildcards: always verified
    const uint64_t unaligned = (uint64_t)le32(g.b + 40) + le32(g.b + 44);
    const uint64_t wild_off = unaligned + (8 - unaligned % 8) % 8, wild_count = le32(g.b + 60);
    for (uint64_t i = 0; i < wild_count; ++i) {
        const uint64_t wo = wild_off + i * 8;
        if (wo + 8 > g.n) continue;
        const uint32_t pid = le32(g.b + wo);
        if (patterns_off + ((size_t)pid + 1) * 16 > g.n) continue;
        if (glob_matches(g, pid, text, gso)) result.push_back(pid);
    }
    // candidates of the literals the automaton reported
    for (const uint32_t lit : lits) {
        if ((size_t)lit + 1 >= tb.lit2pat_off.size()) continue;
        for (uint32_t k = tb.lit2pat_off[lit]; k < tb.lit2pat_off[lit + 1]; ++k) {
            const uint32_t pid = tb.lit2pat[k];
            const size_t eo = patterns_off + (size_t)pid * 16;
            if (eo + 16 > g.n) continue;
            const uint8_t* e = g.b + eo;
            if (e[4] == 0) result.push_back(le32(e));   // literal pattern: the automaton has confirmed it (substring semantics, Q9)
            else if (glob_matches(g, le32(e), text, gso)) result.push_back(le32(e));
        }
    }
    std::sort(result.begin(), result.end());
    result.erase(std::unique(result.begin(), result.end()), result.end());
}

}  // namespace

HostTables::HostTables(const DbImage& img) {
    // find_ipv4_start_node (tree.rs:258-277): 96 left links; the walk stops where a record leaves the node range and IPv4 bits are
    // then read from THAT node
    if (img.has_ip && img.ip_version == 6) {
        uint32_t node = 0;
        for (int k = 0; k < 96; ++k) {
            const uint32_t rec = tree_record(img, node, 0);
            if (rec < img.node_count) node = rec; else break;
        }
        v4_start_node = node;
    }
    img.build_lit2pat(lit2pat_off, lit2pat);
}

void host_lookup(const DbImage& img, const HostTables& t, const std::string& query, const IpAddr* ip, HostHit& out) {
    out = HostHit{};
    if (ip) {
        if (!img.has_ip) return;
        // lookup_v4 starts behind the 96 zero bits of an IPv6 tree; lookup_v6 walks from the root whatever the tree's version
        if (!ip->v6) tree_walk(img, img.ip_version == 6 ? t.v4_start_node : 0u, ip->b, 32, 0, out);
        else tree_walk(img, 0u, ip->b, 128, 0, out);
        return;
    }
    uint32_t pid;
    const bool lit = literal_lookup(img, query, pid);
    if (img.has_glob) find_all(img, t, query, out.globs);
    if (lit || !out.globs.empty()) { out.kind = 3; out.a = lit ? pid : 0xFFFFFFFFu; }
}

}  // namespace mxy
