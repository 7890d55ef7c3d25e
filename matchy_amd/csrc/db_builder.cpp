#include "db_builder.h"
#include "unicode_lower.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <unordered_map>
#include <unordered_set>

#include "hashes.h"

namespace mxy {

namespace {

void put32(std::vector<uint8_t>& b, uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
void put16(std::vector<uint8_t>& b, uint16_t v) { b.push_back((uint8_t)v); b.push_back((uint8_t)(v >> 8)); }
void put64(std::vector<uint8_t>& b, uint64_t v) { for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
void set32(std::vector<uint8_t>& b, size_t off, uint32_t v) { for (int i = 0; i < 4; ++i) b[off + i] = (uint8_t)(v >> (8 * i)); }

// ------------------------------------------------------------------------------------------ UTF-8 helpers
std::vector<uint32_t> decode_utf8(const std::string& s) {
    std::vector<uint32_t> out;
    size_t i = 0, n = s.size();
    while (i < n) {
        uint8_t c = (uint8_t)s[i];
        size_t l = c < 0x80 ? 1 : c < 0xE0 ? 2 : c < 0xF0 ? 3 : 4;
        if (i + l > n) l = n - i;
        uint32_t cp = l == 1 ? c : l == 2 ? (c & 0x1F) : l == 3 ? (c & 0x0F) : (c & 0x07);
        for (size_t k = 1; k < l; ++k) cp = (cp << 6) | ((uint8_t)s[i + k] & 0x3F);
        out.push_back(cp);
        i += l;
    }
    return out;
}
void append_utf8(uint32_t cp, std::string& s) {
    if (cp < 0x80) s.push_back((char)cp);
    else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else { s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
}

// ------------------------------------------------------------------------------------------ glob parsing
struct ClassItem { bool range; uint32_t a, b; };
struct GlobSeg {
    int type;  // 0 literal, 1 star, 2 question, 3 class
    std::string lit;
    std::vector<ClassItem> items;
    bool negated = false;
};

// GlobPattern::parse + optimize_segments (glob.rs:307-451)
bool parse_glob(const std::string& pattern, std::vector<GlobSeg>& segs, std::string& err) {
    std::vector<uint32_t> cs = decode_utf8(pattern);
    std::vector<GlobSeg> raw;
    std::string lit;
    auto flush = [&]() {
        if (!lit.empty()) { GlobSeg s; s.type = 0; s.lit.swap(lit); raw.push_back(std::move(s)); lit.clear(); }
    };
    size_t i = 0, n = cs.size();
    while (i < n) {
        uint32_t ch = cs[i++];
        if (ch == '*') { flush(); GlobSeg s; s.type = 1; raw.push_back(s); }
        else if (ch == '?') { flush(); GlobSeg s; s.type = 2; raw.push_back(s); }
        else if (ch == '[') {
            flush();
            GlobSeg s;
            s.type = 3;
            if (i < n && (cs[i] == '!' || cs[i] == '^')) { s.negated = true; ++i; }
            bool have_prev = false, expect_range_end = false;
            uint32_t prev = 0;
            for (;;) {
                if (i >= n) { err = "Unclosed character class"; return false; }
                uint32_t cc = cs[i++];
                if (cc == ']' && (!s.items.empty() || have_prev)) {
                    if (have_prev) s.items.push_back({false, prev, 0});
                    break;
                }
                if (cc == '-' && have_prev && i < n && cs[i] != ']') {
                    expect_range_end = true;
                } else if (expect_range_end) {
                    if (prev > cc) { err = "Invalid character range"; return false; }
                    s.items.push_back({true, prev, cc});
                    have_prev = false;
                    expect_range_end = false;
                } else {
                    if (have_prev) s.items.push_back({false, prev, 0});
                    prev = cc;
                    have_prev = true;
                }
            }
            if (s.items.empty()) { err = "Empty character class"; return false; }
            raw.push_back(std::move(s));
        } else if (ch == '\\') {
            if (i >= n) { err = "Trailing backslash in pattern"; return false; }
            append_utf8(cs[i++], lit);
        } else {
            append_utf8(ch, lit);
        }
    }
    flush();
    segs.clear();
    std::string buf;
    for (auto& s : raw) {
        if (s.type == 0) buf += s.lit;
        else {
            if (!buf.empty()) { GlobSeg l; l.type = 0; l.lit.swap(buf); segs.push_back(std::move(l)); buf.clear(); }
            segs.push_back(std::move(s));
        }
    }
    if (!buf.empty()) { GlobSeg l; l.type = 0; l.lit.swap(buf); segs.push_back(std::move(l)); }
    return true;
}

// PatternType::is_glob / extract_literals (paraglob_offset.rs:93-159)
bool pattern_is_glob(const std::string& p) {
    bool esc = false;
    for (uint32_t ch : decode_utf8(p)) {
        if (esc) { esc = false; continue; }
        if (ch == '\\') esc = true;
        else if (ch == '*' || ch == '?' || ch == '[') return true;
    }
    return false;
}
std::vector<std::string> extract_literals(const std::string& p) {
    std::vector<std::string> lits;
    std::string cur;
    std::vector<uint32_t> cs = decode_utf8(p);
    bool esc = false;
    size_t i = 0, n = cs.size();
    while (i < n) {
        uint32_t ch = cs[i++];
        if (esc) { append_utf8(ch, cur); esc = false; continue; }
        if (ch == '\\') esc = true;
        else if (ch == '*' || ch == '?') { if (!cur.empty()) { lits.push_back(cur); cur.clear(); } }
        else if (ch == '[') {
            if (!cur.empty()) { lits.push_back(cur); cur.clear(); }
            int depth = 1;
            while (i < n) {
                uint32_t c = cs[i++];
                if (c == '\\') { if (i < n) ++i; }
                else if (c == '[') ++depth;
                else if (c == ']') { if (--depth == 0) break; }
            }
        } else append_utf8(ch, cur);
    }
    if (!cur.empty()) lits.push_back(cur);
    return lits;
}

// ------------------------------------------------------------------------------------------ AC automaton
// ACBuilder (crates/matchy-ac/src/lib.rs:139-517), case-sensitive.
struct AcState {
    std::map<uint8_t, uint32_t> tr;
    uint32_t fail = 0;
    std::vector<uint32_t> out;
};
struct AcResult { std::vector<uint8_t> buf; size_t node_count = 0; };

// `ci`: the bytes that go into the trie are the lower-cased literal (matchy-ac/src/lib.rs:204-210: `to_lowercase()`), the
// text is ASCII-lower-cased at search time instead of doubling the transitions.
AcResult build_ac(const std::vector<std::string>& pats, bool ci) {
    std::vector<AcState> st(1);
    for (size_t pid = 0; pid < pats.size(); ++pid) {
        uint32_t cur = 0;
        const std::string bytes = ci ? LowerTable::get().to_lowercase(pats[pid]) : pats[pid];
        for (unsigned char ch : bytes) {
            auto it = st[cur].tr.find(ch);
            if (it != st[cur].tr.end()) cur = it->second;
            else {
                uint32_t id = (uint32_t)st.size();
                st.emplace_back();
                st[cur].tr[ch] = id;
                cur = id;
            }
        }
        st[cur].out.push_back((uint32_t)pid);
    }
    std::deque<uint32_t> q;
    for (auto& kv : st[0].tr) { st[kv.second].fail = 0; q.push_back(kv.second); }
    while (!q.empty()) {
        uint32_t s = q.front();
        q.pop_front();
        std::vector<std::pair<uint8_t, uint32_t>> trs(st[s].tr.begin(), st[s].tr.end());
        for (auto& t : trs) {
            uint8_t ch = t.first;
            uint32_t nx = t.second;
            q.push_back(nx);
            uint32_t f = st[s].fail;
            bool found = false;
            while (f != 0) {
                auto it = st[f].tr.find(ch);
                if (it != st[f].tr.end()) { st[nx].fail = it->second; found = true; break; }
                f = st[f].fail;
            }
            if (!found) {
                auto it = st[0].tr.find(ch);
                if (it != st[0].tr.end() && it->second != nx) st[nx].fail = it->second;
                else st[nx].fail = 0;
            }
            // merge outputs along the whole failure chain (duplicates are kept, as the reference does)
            uint32_t suf = st[nx].fail;
            while (suf != 0) {
                if (!st[suf].out.empty()) {
                    std::vector<uint32_t> so = st[suf].out;
                    st[nx].out.insert(st[nx].out.end(), so.begin(), so.end());
                }
                suf = st[suf].fail;
            }
        }
    }
    // serialize
    const size_t NODE = 20, EDGE = 8, DENSE = 1024;
    size_t nodes_size = st.size() * NODE;
    size_t dense_count = 0, sparse_edges = 0, total_pat = 0;
    for (auto& s : st) {
        size_t k = s.tr.size();
        if (k >= 9) ++dense_count;
        else if (k >= 2) sparse_edges += k;
        total_pat += s.out.size();
    }
    size_t edges_start = nodes_size, edges_size = sparse_edges * EDGE;
    size_t unaligned = edges_start + edges_size;
    size_t dense_align = 4;  // matchy-ac's DenseLookup is repr(C) over [u32;256]
    size_t pad = dense_count ? (dense_align - unaligned % dense_align) % dense_align : 0;
    size_t dense_start = unaligned + pad;
    size_t patterns_start = dense_start + dense_count * DENSE;
    size_t total = patterns_start + total_pat * 4;
    AcResult r;
    r.node_count = st.size();
    r.buf.assign(total, 0);
    size_t eo = edges_start, dof = dense_start, po = patterns_start;
    for (size_t i = 0; i < st.size(); ++i) {
        AcState& s = st[i];
        size_t k = s.tr.size();
        uint8_t kind = k == 0 ? 0 : k == 1 ? 1 : k <= 8 ? 2 : 3;
        uint32_t edges_off = 0, one_target = 0;
        uint8_t one_char = 0;
        if (kind == 1) {
            one_char = s.tr.begin()->first;
            one_target = (uint32_t)(s.tr.begin()->second * NODE);
            edges_off = one_target;
        } else if (kind == 2) {
            edges_off = (uint32_t)eo;
            for (auto& kv : s.tr) {
                r.buf[eo] = kv.first;
                set32(r.buf, eo + 4, (uint32_t)(kv.second * NODE));
                eo += EDGE;
            }
        } else if (kind == 3) {
            edges_off = (uint32_t)dof;
            for (auto& kv : s.tr) set32(r.buf, dof + (size_t)kv.first * 4, (uint32_t)(kv.second * NODE));
            dof += DENSE;
        }
        uint32_t pat_off = s.out.empty() ? 0 : (uint32_t)po;
        for (uint32_t id : s.out) { set32(r.buf, po, id); po += 4; }
        size_t no = i * NODE;
        r.buf[no] = kind;
        r.buf[no + 1] = one_char;
        r.buf[no + 2] = kind == 1 ? 0 : (uint8_t)std::min<size_t>(k, 255);
        r.buf[no + 3] = (uint8_t)std::min<size_t>(s.out.size(), 255);
        set32(r.buf, no + 4, one_target);
        set32(r.buf, no + 8, s.fail == 0 ? 0 : (uint32_t)(s.fail * NODE));
        set32(r.buf, no + 12, edges_off);
        set32(r.buf, no + 16, pat_off);
    }
    return r;
}

// ------------------------------------------------------------------------------------------ PARAGLOB section
// ParaglobBuilder::build_internal_v3 (paraglob_offset.rs:524-888), no per-pattern data (combined databases
// keep data in the MMDB data section).
struct ParaglobBuilder {
    struct Pat { std::string text; int cls; std::vector<std::string> lits; };  // cls: 0 literal, 1 glob, 2 pure wildcard
    std::vector<Pat> pats;
    std::unordered_map<std::string, uint32_t> index;
    bool case_insensitive = false;

    bool add(const std::string& p, uint32_t& id, std::string& err) {
        auto it = index.find(p);
        if (it != index.end()) { id = it->second; return true; }
        if (p.empty()) { err = "Empty pattern"; return false; }
        Pat pt;
        pt.text = p;
        if (pattern_is_glob(p)) {
            pt.lits = extract_literals(p);
            pt.cls = pt.lits.empty() ? 2 : 1;
        } else pt.cls = 0;
        id = (uint32_t)pats.size();
        index.emplace(p, id);
        pats.push_back(std::move(pt));
        return true;
    }

    bool build(std::vector<uint8_t>& buffer, size_t& ac_nodes, std::string& err) {
        std::vector<std::string> ac_lits;
        std::unordered_map<std::string, uint32_t> lit_id;
        std::vector<std::vector<uint32_t>> lit_pats;
        auto add_lit = [&](const std::string& l, uint32_t pid) {
            auto it = lit_id.find(l);
            uint32_t id;
            if (it == lit_id.end()) { id = (uint32_t)ac_lits.size(); lit_id.emplace(l, id); ac_lits.push_back(l); lit_pats.emplace_back(); }
            else id = it->second;
            lit_pats[id].push_back(pid);
        };
        for (uint32_t pid = 0; pid < pats.size(); ++pid) {
            const Pat& p = pats[pid];
            if (p.cls == 0) add_lit(p.text, pid);
            else if (p.cls == 1) for (const auto& l : p.lits) { if (l.size() < 3) continue; add_lit(l, pid); }
        }
        AcResult ac;
        if (!ac_lits.empty()) ac = build_ac(ac_lits, case_insensitive);
        ac_nodes = ac.node_count;

        // ACLH (literal_hash.rs:121-200)
        std::vector<uint8_t> aclh;
        if (!ac_lits.empty()) {
            size_t n = ac_lits.size();
            // Sized and filled like ACLiteralHashBuilder::build (literal_hash.rs:120-160): max(ceil(1.25 n), 16) slots, entries
            // placed by linear probing from FxHasher(literal_id) % table_size. rustc-hash 2.1.1's FxHasher cannot be verified in
            // this image (believed: (id * 0xf1357aea2e62a9c5).rotate_left(26), hashes.h fx_u32), and the reference
            // reader gives up at the first EMPTY slot (literal_hash.rs:263-299) — so the slots the reference would leave empty
            // carry FILLER entries here: a literal id no automaton ever reports (0x80000000 | slot; real ids are dense from 0),
            // no patterns. With the believed function right, every real id is found exactly as in a reference-built table (its
            // probe path only crosses real entries: fillers are placed last, into slots no insertion used); with it wrong, the
            // probe still reaches the entry because nothing stops it (bounded by table_size, literal_hash.rs:272). Lookups of
            // ids that are absent never happen (paraglob_offset.rs:1070-1073 looks up ids the automaton reported). The
            // reference's validator skips nothing but 0xFFFFFFFF and accepts entries without patterns (validation.rs:231-283).
            // This repository's readers do not use the slot function at all (they enumerate the table once at open) and skip
            // entries without patterns.
            size_t table_size = std::max<size_t>((n * 5 + 3) / 4, 16);
            std::vector<uint8_t> lists;
            std::vector<uint32_t> offs(n);
            for (size_t l = 0; l < n; ++l) { offs[l] = (uint32_t)lists.size(); for (uint32_t pid : lit_pats[l]) put32(lists, pid); }
            std::vector<uint32_t> slot_lit(table_size, 0xFFFFFFFFu);
            for (size_t l = 0; l < n; ++l) {
                size_t slot = (size_t)(fx_u32((uint32_t)l) % table_size);
                while (slot_lit[slot] != 0xFFFFFFFFu) slot = (slot + 1) % table_size;
                slot_lit[slot] = (uint32_t)l;
            }
            aclh.insert(aclh.end(), {'A', 'C', 'L', 'H'});
            put32(aclh, 1);
            put32(aclh, (uint32_t)n);
            put32(aclh, (uint32_t)table_size);
            put32(aclh, (uint32_t)(24 + table_size * 16));
            put32(aclh, (uint32_t)lists.size());
            for (size_t s = 0; s < table_size; ++s) {
                uint32_t l = slot_lit[s];
                if (l == 0xFFFFFFFFu) { put32(aclh, 0x80000000u | (uint32_t)s); put32(aclh, 0); put32(aclh, 0); put32(aclh, 0); }   // filler
                else { put32(aclh, l); put32(aclh, offs[l]); put32(aclh, (uint32_t)lit_pats[l].size()); put32(aclh, 0); }
            }
            aclh.insert(aclh.end(), lists.begin(), lists.end());
        }

        // glob segment section (paraglob_offset.rs:368-522)
        struct SegHdr { uint8_t type, flags; uint32_t len, off; };
        std::vector<std::pair<uint32_t, uint16_t>> seg_index;  // (first header index, count)
        std::vector<SegHdr> hdrs;
        std::vector<uint8_t> strdata, ccdata;
        for (const Pat& p : pats) {
            std::vector<GlobSeg> segs;
            if (!parse_glob(p.text, segs, err)) return false;
            seg_index.emplace_back((uint32_t)hdrs.size(), (uint16_t)segs.size());
            for (const GlobSeg& s : segs) {
                if (s.type == 0) {
                    hdrs.push_back({0, 0, (uint32_t)s.lit.size(), (uint32_t)strdata.size()});
                    strdata.insert(strdata.end(), s.lit.begin(), s.lit.end());
                } else if (s.type == 1 || s.type == 2) hdrs.push_back({(uint8_t)s.type, 0, 0, 0});
                else {
                    hdrs.push_back({3, (uint8_t)(s.negated ? 1 : 0), (uint32_t)(s.items.size() * 12), (uint32_t)ccdata.size()});
                    for (const ClassItem& it : s.items) {
                        ccdata.push_back(it.range ? 1 : 0); ccdata.push_back(0); ccdata.push_back(0); ccdata.push_back(0);
                        put32(ccdata, it.a);
                        put32(ccdata, it.range ? it.b : 0);
                    }
                }
            }
        }

        // layout
        auto align_up = [](size_t x, size_t a) { return x + (a - x % a) % a; };
        const size_t header_size = 112;
        size_t ac_start = align_up(header_size, 64);
        size_t ac_size = ac.buf.size();
        size_t patterns_start = align_up(ac_start + ac_size, 8);
        size_t pattern_entries_size = pats.size() * 16;
        size_t strings_start = patterns_start + pattern_entries_size;
        std::vector<uint8_t> strings;
        std::vector<size_t> str_off;
        for (const Pat& p : pats) { str_off.push_back(strings.size()); strings.insert(strings.end(), p.text.begin(), p.text.end()); strings.push_back(0); }
        size_t wild_start = align_up(strings_start + strings.size(), 8);
        std::vector<uint32_t> wild_ids;
        for (uint32_t pid = 0; pid < pats.size(); ++pid) if (pats[pid].cls == 2) wild_ids.push_back(pid);
        size_t data_section_start = wild_start + wild_ids.size() * 8;
        size_t mappings_start = align_up(data_section_start, 4);
        size_t aclh_start = mappings_start;
        size_t glob_start = align_up(aclh_start + aclh.size(), 8);
        size_t index_size = pats.size() * 8;
        size_t hdrs_size = hdrs.size() * 12;
        size_t glob_size = index_size + hdrs_size + strdata.size() + ccdata.size();
        size_t total = glob_start + glob_size;
        buffer.assign(total, 0);

        memcpy(buffer.data(), "PARAGLOB", 8);
        set32(buffer, 8, 5);
        set32(buffer, 12, case_insensitive ? 1 : 0);  // match_mode (paraglob_offset.rs:720-723)
        set32(buffer, 16, (uint32_t)ac.node_count);
        set32(buffer, 20, (uint32_t)ac_start);
        set32(buffer, 24, (uint32_t)ac_size);
        set32(buffer, 32, (uint32_t)pats.size());
        set32(buffer, 36, (uint32_t)patterns_start);
        set32(buffer, 40, (uint32_t)strings_start);
        set32(buffer, 44, (uint32_t)strings.size());
        set32(buffer, 60, (uint32_t)wild_ids.size());
        set32(buffer, 64, (uint32_t)total);
        buffer[68] = 0x01;
        set32(buffer, 96, (uint32_t)aclh_start);
        set32(buffer, 100, (uint32_t)ac_lits.size());
        set32(buffer, 104, (uint32_t)glob_start);
        set32(buffer, 108, (uint32_t)glob_size);

        if (ac_size) memcpy(&buffer[ac_start], ac.buf.data(), ac_size);
        for (uint32_t pid = 0; pid < pats.size(); ++pid) {
            size_t eo = patterns_start + (size_t)pid * 16;
            set32(buffer, eo, pid);
            buffer[eo + 4] = pats[pid].cls == 0 ? 0 : 1;
            set32(buffer, eo + 8, (uint32_t)(strings_start + str_off[pid]));
            set32(buffer, eo + 12, (uint32_t)pats[pid].text.size());
        }
        if (!strings.empty()) memcpy(&buffer[strings_start], strings.data(), strings.size());
        for (size_t i = 0; i < wild_ids.size(); ++i) {
            set32(buffer, wild_start + i * 8, wild_ids[i]);
            set32(buffer, wild_start + i * 8 + 4, (uint32_t)(strings_start + str_off[wild_ids[i]]));
        }
        if (!aclh.empty()) memcpy(&buffer[aclh_start], aclh.data(), aclh.size());
        size_t hdr_base = glob_start + index_size;
        size_t str_base = hdr_base + hdrs_size;
        size_t cc_base = str_base + strdata.size();
        for (size_t i = 0; i < seg_index.size(); ++i) {
            size_t io = glob_start + i * 8;
            set32(buffer, io, (uint32_t)(hdr_base + (size_t)seg_index[i].first * 12));
            buffer[io + 4] = (uint8_t)seg_index[i].second;
            buffer[io + 5] = (uint8_t)(seg_index[i].second >> 8);
        }
        for (size_t i = 0; i < hdrs.size(); ++i) {
            size_t ho = hdr_base + i * 12;
            buffer[ho] = hdrs[i].type;
            buffer[ho + 1] = hdrs[i].flags;
            set32(buffer, ho + 4, hdrs[i].len);
            uint32_t off = 0;
            if (hdrs[i].len > 0) off = (uint32_t)((hdrs[i].type == 0 ? str_base : cc_base) + hdrs[i].off);
            set32(buffer, ho + 8, off);
        }
        if (!strdata.empty()) memcpy(&buffer[str_base], strdata.data(), strdata.size());
        if (!ccdata.empty()) memcpy(&buffer[cc_base], ccdata.data(), ccdata.size());
        return true;
    }
};

// ------------------------------------------------------------------------------------------ LHSH section
// LiteralHashBuilder::build (crates/matchy-literal-hash/src/lib.rs:173-354), case-sensitive.
std::vector<uint8_t> build_literal_hash(const std::vector<std::pair<const std::string*, uint32_t>>& literals,
                                        const std::vector<std::pair<uint32_t, uint32_t>>& pattern_data) {
    std::vector<uint8_t> out;
    if (literals.empty()) return out;
    size_t n = literals.size();
    uint32_t shard_bits = n < 10000 ? 4 : n < 100000 ? 5 : 6;
    size_t num_shards = (size_t)1 << shard_bits;
    struct Ent { const std::string* s; uint32_t id; uint64_t h; };
    std::vector<std::vector<Ent>> buckets(num_shards);
    for (auto& l : literals) {
        uint64_t h = xxh64((const uint8_t*)l.first->data(), l.first->size(), 0);
        buckets[(size_t)(h % num_shards)].push_back({l.first, l.second, h});
    }
    struct HEnt { uint64_t h; uint32_t so, id; };
    std::vector<std::vector<HEnt>> tables(num_shards);
    std::vector<std::vector<uint8_t>> pools(num_shards);
    for (size_t sh = 0; sh < num_shards; ++sh) {
        auto& es = buckets[sh];
        if (es.empty()) continue;
        size_t needed = (size_t)std::ceil((double)es.size() / 0.60);
        size_t cap = 16;
        while (cap < needed) cap <<= 1;
        size_t mask = cap - 1;
        std::vector<uint32_t> soffs;
        auto& pool = pools[sh];
        for (auto& e : es) {
            soffs.push_back((uint32_t)pool.size());
            put16(pool, (uint16_t)e.s->size());
            pool.insert(pool.end(), e.s->begin(), e.s->end());
            pool.push_back(0);
        }
        // one table entry per distinct hash, the last occurrence wins (FxHashMap::insert in the reference)
        std::unordered_map<uint64_t, std::pair<uint32_t, uint32_t>> m;
        std::vector<uint64_t> order;
        for (size_t i = 0; i < es.size(); ++i) {
            auto it = m.find(es[i].h);
            if (it == m.end()) { m.emplace(es[i].h, std::make_pair(soffs[i], es[i].id)); order.push_back(es[i].h); }
            else it->second = std::make_pair(soffs[i], es[i].id);
        }
        auto& tab = tables[sh];
        tab.assign(cap, HEnt{0, 0xFFFFFFFFu, 0});
        for (uint64_t h : order) {
            size_t pos = (size_t)h & mask;
            while (tab[pos].so != 0xFFFFFFFFu) pos = (pos + 1) & mask;
            tab[pos] = HEnt{h, m[h].first, m[h].second};
        }
    }
    size_t table_size = 0;
    std::vector<uint32_t> shard_off(num_shards + 1);
    for (size_t sh = 0; sh < num_shards; ++sh) { shard_off[sh] = (uint32_t)table_size; table_size += tables[sh].size(); }
    shard_off[num_shards] = (uint32_t)table_size;
    size_t pool_total = 0;
    for (auto& p : pools) pool_total += p.size();
    size_t strings_offset = 32 + (num_shards + 1) * 4 + table_size * 16;
    size_t entry_count = 0;
    for (auto& t : tables) for (auto& e : t) entry_count += e.so != 0xFFFFFFFFu;
    out.reserve(strings_offset + pool_total + 4 + pattern_data.size() * 8);
    out.insert(out.end(), {'L', 'H', 'S', 'H'});
    put32(out, 1);
    put32(out, (uint32_t)entry_count);
    put32(out, (uint32_t)table_size);
    put32(out, (uint32_t)strings_offset);
    put32(out, (uint32_t)pool_total);
    put32(out, (uint32_t)num_shards);
    put32(out, shard_bits);
    for (uint32_t o : shard_off) put32(out, o);
    uint32_t pool_base = 0;
    for (size_t sh = 0; sh < num_shards; ++sh) {
        for (auto& e : tables[sh]) {
            put64(out, e.h);
            put32(out, e.so == 0xFFFFFFFFu ? 0xFFFFFFFFu : e.so + pool_base);
            put32(out, e.id);
        }
        pool_base += (uint32_t)pools[sh].size();
    }
    for (auto& p : pools) out.insert(out.end(), p.begin(), p.end());
    put32(out, (uint32_t)pattern_data.size());
    for (auto& pd : pattern_data) { put32(out, pd.first); put32(out, pd.second); }
    return out;
}

// ------------------------------------------------------------------------------------------ IP tree
// IpTreeBuilder (crates/matchy-ip-trie/src/lib.rs:62-449)
struct Ptr { uint32_t val = 0; uint8_t kind = 0, pfx = 0; };  // kind: 0 empty, 1 node, 2 data
struct TNode { Ptr c[2]; };
struct IpTree {
    std::vector<TNode> nodes;
    bool v6;
    explicit IpTree(bool is_v6) : nodes(1), v6(is_v6) {}

    void backfill(uint32_t node_id, uint32_t off, uint8_t pfx) {
        std::vector<uint32_t> stack{node_id};
        while (!stack.empty()) {
            uint32_t id = stack.back();
            stack.pop_back();
            for (int side = 0; side < 2; ++side) {
                Ptr p = nodes[id].c[side];
                if (p.kind == 0) nodes[id].c[side] = Ptr{off, 2, pfx};
                else if (p.kind == 2) { if (pfx > p.pfx) nodes[id].c[side] = Ptr{off, 2, pfx}; }
                else stack.push_back(p.val);
            }
        }
    }
    // bits: 128-bit big-endian address image (v4 trees: address in the top 32 bits)
    void insert_bits(const uint8_t bits[16], unsigned prefix_len, uint32_t off) {
        uint32_t node = 0;
        for (unsigned depth = 0; depth < prefix_len; ++depth) {
            int bit = (bits[depth / 8] >> (7 - depth % 8)) & 1;
            Ptr child = nodes[node].c[bit];
            if (depth + 1 == prefix_len) {
                if (child.kind == 0) nodes[node].c[bit] = Ptr{off, 2, (uint8_t)prefix_len};
                else if (child.kind == 2) { if (prefix_len >= child.pfx) nodes[node].c[bit] = Ptr{off, 2, (uint8_t)prefix_len}; }
                else backfill(child.val, off, (uint8_t)prefix_len);
                return;
            }
            if (child.kind == 0) {
                uint32_t id = (uint32_t)nodes.size();
                nodes.emplace_back();
                nodes[node].c[bit] = Ptr{id, 1, 0};
                node = id;
            } else if (child.kind == 1) {
                node = child.val;
            } else {
                uint32_t id = (uint32_t)nodes.size();
                nodes.emplace_back();
                nodes[id].c[0] = child;
                nodes[id].c[1] = child;
                nodes[node].c[bit] = Ptr{id, 1, 0};
                node = id;
            }
        }
    }
    void insert(const IpAddr& a, uint8_t prefix_len, uint32_t off) {
        uint8_t bits[16] = {0};
        if (!a.v6) {
            if (v6) { memcpy(bits + 12, a.b, 4); insert_bits(bits, 96u + prefix_len, off); }
            else { memcpy(bits, a.b, 4); insert_bits(bits, prefix_len, off); }
        } else {
            memcpy(bits, a.b, 16);
            insert_bits(bits, prefix_len, off);
        }
    }
    uint64_t max_record() const {
        uint64_t nc = nodes.size(), mx = nc;
        for (auto& n : nodes) for (int s = 0; s < 2; ++s) if (n.c[s].kind == 2) mx = std::max<uint64_t>(mx, nc + 16 + n.c[s].val);
        return mx;
    }
    std::vector<uint8_t> serialize(int record_size) const {
        uint32_t nc = (uint32_t)nodes.size();
        size_t nb = (size_t)record_size * 2 / 8;
        std::vector<uint8_t> t(nodes.size() * nb, 0);
        for (size_t i = 0; i < nodes.size(); ++i) {
            uint32_t r[2];
            for (int s = 0; s < 2; ++s) {
                const Ptr& p = nodes[i].c[s];
                r[s] = p.kind == 0 ? nc : p.kind == 1 ? p.val : nc + 16 + p.val;
            }
            uint8_t* o = &t[i * nb];
            if (record_size == 24) {
                o[0] = (uint8_t)(r[0] >> 16); o[1] = (uint8_t)(r[0] >> 8); o[2] = (uint8_t)r[0];
                o[3] = (uint8_t)(r[1] >> 16); o[4] = (uint8_t)(r[1] >> 8); o[5] = (uint8_t)r[1];
            } else if (record_size == 28) {
                o[0] = (uint8_t)(r[0] >> 16); o[1] = (uint8_t)(r[0] >> 8); o[2] = (uint8_t)r[0];
                o[3] = (uint8_t)((((r[0] >> 24) & 0xF) << 4) | ((r[1] >> 24) & 0xF));
                o[4] = (uint8_t)(r[1] >> 16); o[5] = (uint8_t)(r[1] >> 8); o[6] = (uint8_t)r[1];
            } else {
                for (int s = 0; s < 2; ++s) { o[4 * s] = (uint8_t)(r[s] >> 24); o[4 * s + 1] = (uint8_t)(r[s] >> 16); o[4 * s + 2] = (uint8_t)(r[s] >> 8); o[4 * s + 3] = (uint8_t)r[s]; }
            }
        }
        return t;
    }
};

bool parse_u8(const std::string& s, uint8_t& out) {  // Rust `str::parse::<u8>()`: optional '+', decimal digits, <= 255
    size_t i = 0;
    if (!s.empty() && s[0] == '+') i = 1;
    if (i >= s.size()) return false;
    unsigned v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (s[i] - '0');
        if (v > 255) return false;
    }
    out = (uint8_t)v;
    return true;
}

}  // namespace

bool validate_glob_pattern(const std::string& pattern, std::string& err) {
    std::vector<GlobSeg> segs;
    return parse_glob(pattern, segs, err);
}

bool DatabaseBuilder::parse_ip_entry(const std::string& key, IpAddr& addr, uint8_t& prefix_len) {
    if (parse_ip(key.data(), key.size(), addr)) { prefix_len = addr.v6 ? 128 : 32; return true; }
    size_t slash = key.find('/');
    if (slash != std::string::npos) {
        uint8_t p;
        if (parse_ip(key.data(), slash, addr) && parse_u8(key.substr(slash + 1), p)) {
            if (p <= (addr.v6 ? 128 : 32)) { prefix_len = p; return true; }
        }
    }
    return false;
}

bool DatabaseBuilder::detect_entry_type(const std::string& key, EntryKind& kind, std::string& stripped, IpAddr& addr,
                                        uint8_t& prefix_len, std::string& err) {
    if (key.rfind("literal:", 0) == 0) { kind = EntryKind::LITERAL; stripped = key.substr(8); return true; }
    if (key.rfind("glob:", 0) == 0) {
        stripped = key.substr(5);
        std::string gerr;
        if (!validate_glob_pattern(stripped, gerr)) { err = "Invalid glob pattern syntax: " + gerr; return false; }
        kind = EntryKind::GLOB;
        return true;
    }
    if (key.rfind("ip:", 0) == 0) {
        stripped = key.substr(3);
        if (!parse_ip_entry(stripped, addr, prefix_len)) { err = "Invalid IP address or CIDR: " + stripped; return false; }
        kind = EntryKind::IP;
        return true;
    }
    stripped = key;
    if (parse_ip_entry(key, addr, prefix_len)) { kind = EntryKind::IP; return true; }
    if (key.find_first_of("*?[") != std::string::npos) {
        std::string gerr;
        if (validate_glob_pattern(key, gerr)) { kind = EntryKind::GLOB; return true; }
    }
    kind = EntryKind::LITERAL;
    return true;
}

bool DatabaseBuilder::add_entry(const std::string& key, const DataValue& data_map) {
    Entry e;
    std::string stripped;
    if (!detect_entry_type(key, e.kind, stripped, e.addr, e.prefix_len, error_)) return false;
    if (e.kind != EntryKind::IP) e.text = stripped;
    e.data_offset = encode_data(data_map);
    entries_.push_back(std::move(e));
    return true;
}
bool DatabaseBuilder::add_ip(const std::string& s, const DataValue& data_map) {
    Entry e;
    e.kind = EntryKind::IP;
    if (!parse_ip_entry(s, e.addr, e.prefix_len)) { error_ = "Invalid IP address or CIDR: " + s; return false; }
    e.data_offset = encode_data(data_map);
    entries_.push_back(std::move(e));
    return true;
}
bool DatabaseBuilder::add_literal(const std::string& s, const DataValue& data_map) {
    Entry e;
    e.kind = EntryKind::LITERAL;
    e.text = s;
    e.data_offset = encode_data(data_map);
    entries_.push_back(std::move(e));
    return true;
}
bool DatabaseBuilder::add_glob(const std::string& s, const DataValue& data_map) {
    Entry e;
    e.kind = EntryKind::GLOB;
    e.text = s;
    e.data_offset = encode_data(data_map);
    entries_.push_back(std::move(e));
    return true;
}

// DatabaseBuilder::build (mmdb_builder.rs:432-760)
bool DatabaseBuilder::build(std::vector<uint8_t>& db) {
    std::vector<uint8_t> data_section = encoder_.bytes();
    std::vector<const Entry*> ips, lits, globs;
    for (const Entry& e : entries_) (e.kind == EntryKind::IP ? ips : e.kind == EntryKind::LITERAL ? lits : globs).push_back(&e);
    stats_ = BuildStats();
    stats_.ip_entries = ips.size(); stats_.literal_entries = lits.size(); stats_.glob_entries = globs.size();
    stats_.data_section_bytes = data_section.size();

    std::vector<uint8_t> tree_bytes;
    uint32_t node_count;
    int record_size = 24, ip_version = 4;
    if (!ips.empty()) {
        bool needs_v6 = false;
        for (auto* e : ips) needs_v6 |= e->addr.v6;
        size_t est = ips.size();
        record_size = est > 200000000 ? 32 : est > 15000000 ? 28 : 24;
        // test hook: a larger minimum record size, so that the 28- and 32-bit readers are exercised by small databases
        if (const char* f = getenv("MATCHY_AMD_MIN_RECORD_SIZE")) {
            const int want = atoi(f);
            if ((want == 28 || want == 32) && want > record_size) record_size = want;
        }
        // (prefix desc, addr asc) with IpAddr ordering V4 < V6 (mmdb_builder.rs:485-487)
        std::stable_sort(ips.begin(), ips.end(), [](const Entry* a, const Entry* b) {
            if (a->prefix_len != b->prefix_len) return a->prefix_len > b->prefix_len;
            if (a->addr.v6 != b->addr.v6) return !a->addr.v6;
            return memcmp(a->addr.b, b->addr.b, a->addr.v6 ? 16 : 4) < 0;
        });
        IpTree tree(needs_v6);
        tree.nodes.reserve(est + est / 2);
        for (auto* e : ips) tree.insert(e->addr, e->prefix_len, e->data_offset);
        // DEVIATION: the reference keeps the record size chosen from the entry count and silently
        // truncates records that do not fit; we widen the record instead so the tree stays valid.
        uint64_t mx = tree.max_record();
        while (record_size < 32 && mx >= (1ull << record_size)) { record_size += 4; stats_.record_size_bumped = true; }
        if (mx > 0xFFFFFFFFull) { error_ = "IP tree too large for 32-bit records"; return false; }
        tree_bytes = tree.serialize(record_size);
        node_count = (uint32_t)tree.nodes.size();
        ip_version = needs_v6 ? 6 : 4;
    } else {
        IpTree tree(false);
        tree_bytes = tree.serialize(24);
        node_count = 1;
    }
    stats_.node_count = node_count; stats_.record_size = record_size; stats_.ip_version = ip_version;

    bool has_globs = !globs.empty();
    std::vector<uint8_t> glob_section;
    if (has_globs) {
        ParaglobBuilder pb;
        pb.case_insensitive = case_insensitive_;
        std::vector<uint32_t> offsets;
        for (auto* e : globs) {
            uint32_t id;
            if (!pb.add(e->text, id, error_)) return false;
            offsets.push_back(e->data_offset);  // positional, one per glob ENTRY (Q10)
        }
        std::vector<uint8_t> pg;
        if (!pb.build(pg, stats_.ac_nodes, error_)) return false;
        put32(glob_section, 0);
        put32(glob_section, 0);
        glob_section.insert(glob_section.end(), pg.begin(), pg.end());
        put32(glob_section, (uint32_t)offsets.size());
        for (uint32_t o : offsets) put32(glob_section, o);
        set32(glob_section, 0, (uint32_t)glob_section.size());
        set32(glob_section, 4, (uint32_t)pg.size());
    }
    bool has_literals = !lits.empty();
    std::vector<uint8_t> literal_section;
    if (has_literals) {
        std::vector<std::pair<const std::string*, uint32_t>> l;
        std::vector<std::pair<uint32_t, uint32_t>> pd;
        // case-insensitive: LiteralHashBuilder::add_pattern stores and hashes the lower-cased key
        // (matchy-literal-hash/src/lib.rs:158-167)
        std::vector<std::string> lowered;
        if (case_insensitive_) {
            lowered.reserve(lits.size());
            for (auto* e : lits) lowered.push_back(LowerTable::get().to_lowercase(e->text));
        }
        for (size_t i = 0; i < lits.size(); ++i) {
            l.emplace_back(case_insensitive_ ? &lowered[i] : &lits[i]->text, (uint32_t)i);
            pd.emplace_back((uint32_t)i, lits[i]->data_offset);
        }
        literal_section = build_literal_hash(l, pd);
    }

    db.clear();
    db.insert(db.end(), tree_bytes.begin(), tree_bytes.end());
    db.insert(db.end(), 16, 0);
    db.insert(db.end(), data_section.begin(), data_section.end());
    size_t pad = 0;
    if (has_globs) {
        size_t cur = db.size() + 16;
        pad = (4 - cur % 4) % 4;
        db.insert(db.end(), pad, 0);
    }
    size_t tree_and_sep = tree_bytes.size() + 16;
    size_t pattern_offset = has_globs ? tree_and_sep + data_section.size() + pad + 16 : 0;
    size_t literal_offset = 0;
    if (has_literals) literal_offset = has_globs ? pattern_offset + glob_section.size() + 16 : tree_and_sep + data_section.size() + 16;

    DataValue meta = DataValue::Map();
    meta.map["binary_format_major_version"] = DataValue::Uint16(2);
    meta.map["binary_format_minor_version"] = DataValue::Uint16(0);
    meta.map["build_epoch"] = DataValue::Uint64(has_epoch_ ? build_epoch_ : (uint64_t)time(nullptr));
    std::string db_type = database_type_;
    if (db_type.empty()) {
        if (has_globs || !lits.empty()) db_type = !ips.empty() ? "Paraglob-Combined-IP-Pattern" : "Paraglob-Pattern";
        else db_type = "Paraglob-IP";
    }
    meta.map["database_type"] = DataValue::String(db_type);
    DataValue desc = DataValue::Map();
    if (description_.empty()) desc.map["en"] = DataValue::String("Paraglob unified database with IP and pattern matching");
    else for (auto& kv : description_) desc.map[kv.first] = DataValue::String(kv.second);
    meta.map["description"] = desc;
    DataValue langs = DataValue::Array();
    langs.arr.push_back(DataValue::String("en"));
    meta.map["languages"] = langs;
    meta.map["ip_version"] = DataValue::Uint16((uint16_t)ip_version);
    meta.map["node_count"] = DataValue::Uint32(node_count);
    meta.map["record_size"] = DataValue::Uint16((uint16_t)record_size);
    meta.map["ip_entry_count"] = DataValue::Uint32((uint32_t)ips.size());
    meta.map["literal_entry_count"] = DataValue::Uint32((uint32_t)lits.size());
    meta.map["glob_entry_count"] = DataValue::Uint32((uint32_t)globs.size());
    meta.map["match_mode"] = DataValue::Uint16(case_insensitive_ ? 1 : 0);   // mmdb_builder.rs:676-680
    meta.map["pattern_section_offset"] = DataValue::Uint32((uint32_t)pattern_offset);
    meta.map["literal_section_offset"] = DataValue::Uint32((uint32_t)literal_offset);
    DataEncoder menc;
    menc.encode(meta);

    static const char PAT_SEP[17] = "MMDB_PATTERN\0\0\0\0";
    static const char LIT_SEP[17] = "MMDB_LITERAL\0\0\0\0";
    if (has_globs) { db.insert(db.end(), PAT_SEP, PAT_SEP + 16); db.insert(db.end(), glob_section.begin(), glob_section.end()); }
    if (has_literals) { db.insert(db.end(), LIT_SEP, LIT_SEP + 16); db.insert(db.end(), literal_section.begin(), literal_section.end()); }
    static const uint8_t MARK[14] = {0xAB, 0xCD, 0xEF, 'M', 'a', 'x', 'M', 'i', 'n', 'd', '.', 'c', 'o', 'm'};
    db.insert(db.end(), MARK, MARK + 14);
    db.insert(db.end(), menc.bytes().begin(), menc.bytes().end());
    if (db.size() > 0xFFFFFFFFull) { error_ = "database exceeds 4 GiB section-offset range"; return false; }
    return true;
}

}  // namespace mxy
