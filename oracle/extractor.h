// oracle/extractor.h — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product library.
//
// Literal, pass-by-pass restatement of the reference tokenizer
//   /root/reference/crates/matchy-extractor/src/lib.rs
// (extract_from_chunk :409-488 and the functions it calls). Sequential state such as `last_end`
// and `last_domain_end` is kept exactly as in the reference; the HIP path uses a stateless
// per-run formulation and is differential-tested against this file.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>

#include "crypto.h"

namespace orc {

// matchy.h MATCHY_ITEM_TYPE_* (crates/matchy/include/matchy/matchy.h:233-288)
enum ItemType : uint8_t {
    IT_DOMAIN = 0, IT_EMAIL = 1, IT_IPV4 = 2, IT_IPV6 = 3, IT_MD5 = 4, IT_SHA1 = 5, IT_SHA256 = 6,
    IT_SHA384 = 7, IT_SHA512 = 8, IT_BITCOIN = 9, IT_ETHEREUM = 10, IT_MONERO = 11
};
// matchy.h MATCHY_EXTRACT_* flag bits (:188-228)
enum ExtractFlags : uint32_t {
    EX_DOMAINS = 1, EX_EMAILS = 2, EX_IPV4 = 4, EX_IPV6 = 8, EX_HASHES = 16, EX_BITCOIN = 32,
    EX_ETHEREUM = 64, EX_MONERO = 128, EX_ALL = 255
};

struct Match {
    uint8_t type;
    size_t start, end;   // exclusive end, chunk-relative
    uint8_t ip[16];      // IPv4: ip[0..4] big-endian; IPv6: 16 bytes big-endian
};

// ---- lookup tables (lib.rs:1568-1629, 1696-1717)
inline bool is_boundary(uint8_t b) {
    switch (b) {
        case ' ': case '\t': case '\n': case '\r': case '/': case ',': case ';': case ':': case '(': case ')':
        case '[': case ']': case '{': case '}': case '<': case '>': case '"': case '\'': case '@': case '=':
            return true;
        default: return false;
    }
}
inline bool is_digit(uint8_t b) { return b >= '0' && b <= '9'; }
inline bool is_alpha(uint8_t b) { return (b >= 'a' && b <= 'z') || (b >= 'A' && b <= 'Z'); }
inline bool is_alnum(uint8_t b) { return is_digit(b) || is_alpha(b); }
inline bool is_hex(uint8_t b) { return is_digit(b) || (b >= 'a' && b <= 'f') || (b >= 'A' && b <= 'F'); }
inline bool is_domain_char_fast(uint8_t b) { return is_alnum(b) || b == '-' || b == '.' || b >= 0x80; }  // :1597-1629
inline bool is_domain_char(uint8_t b) { return is_alnum(b) || b == '-' || b == '.'; }                       // :1639-1641
inline bool is_email_local_char(uint8_t b) { return is_alnum(b) || b == '.' || b == '-' || b == '_' || b == '+'; }  // :1644-1647

// Rust core::str::from_utf8 acceptance (strict UTF-8: no overlongs, no surrogates, <= U+10FFFF)
inline bool valid_utf8(const uint8_t* s, size_t n) {
    size_t i = 0;
    while (i < n) {
        uint8_t c = s[i];
        if (c < 0x80) { ++i; continue; }
        if (c >= 0xC2 && c <= 0xDF) {
            if (i + 1 >= n || (s[i + 1] & 0xC0) != 0x80) return false;
            i += 2;
        } else if (c >= 0xE0 && c <= 0xEF) {
            if (i + 2 >= n) return false;
            uint8_t c1 = s[i + 1], c2 = s[i + 2];
            uint8_t lo = 0x80, hi = 0xBF;
            if (c == 0xE0) lo = 0xA0;
            if (c == 0xED) hi = 0x9F;
            if (c1 < lo || c1 > hi || (c2 & 0xC0) != 0x80) return false;
            i += 3;
        } else if (c >= 0xF0 && c <= 0xF4) {
            if (i + 3 >= n) return false;
            uint8_t c1 = s[i + 1], c2 = s[i + 2], c3 = s[i + 3];
            uint8_t lo = 0x80, hi = 0xBF;
            if (c == 0xF0) lo = 0x90;
            if (c == 0xF4) hi = 0x8F;
            if (c1 < lo || c1 > hi || (c2 & 0xC0) != 0x80 || (c3 & 0xC0) != 0x80) return false;
            i += 4;
        } else {
            return false;
        }
    }
    return true;
}

// ---- Public Suffix List set (lib.rs:1546-1563). Loaded from matchy_amd/data/psl.bin (tools/gen_psl.py).
struct Psl {
    std::unordered_set<std::string> set;
    bool load(const char* path) {
        FILE* f = fopen(path, "rb");
        if (!f) return false;
        std::vector<uint8_t> buf;
        uint8_t tmp[65536];
        size_t n;
        while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
        fclose(f);
        if (buf.size() < 16 || memcmp(buf.data(), "PSLB", 4) != 0) return false;
        uint32_t count, bytes;
        memcpy(&count, &buf[8], 4);
        memcpy(&bytes, &buf[12], 4);
        if (16 + (size_t)bytes > buf.size()) return false;
        std::string prev;
        size_t p = 16;
        for (uint32_t i = 0; i < count; ++i) {
            if (p + 2 > buf.size()) return false;
            uint8_t shared = buf[p], rest = buf[p + 1];
            p += 2;
            if (shared > prev.size() || p + rest > buf.size()) return false;
            std::string s = prev.substr(0, shared) + std::string((const char*)&buf[p], rest);
            p += rest;
            set.insert(s);
            prev.swap(s);
        }
        return true;
    }
    bool contains(const uint8_t* s, size_t n) const { return set.count(std::string((const char*)s, n)) != 0; }
};

// find_valid_tld_suffix_bytes (lib.rs:1671-1692): walk dots right-to-left; first suffix in PSL wins.
// Returns index of that dot, or -1.
inline long find_valid_tld_suffix(const Psl& psl, const uint8_t* d, size_t n) {
    for (size_t i = n; i-- > 0;) {
        if (d[i] == '.') {
            if (psl.contains(d + i + 1, n - i - 1)) return (long)i;
        }
    }
    return -1;
}

// find_word_boundaries_into (lib.rs:1742-1782): [start0,end0,start1,end1,...]
inline void find_word_boundaries(const uint8_t* c, size_t n, std::vector<size_t>& out) {
    out.clear();
    if (n == 0) return;
    bool in_token = !is_boundary(c[0]);
    if (in_token) out.push_back(0);
    for (size_t i = 1; i < n; ++i) {
        bool b = is_boundary(c[i]);
        if (in_token && b) { out.push_back(i); in_token = false; }
        else if (!in_token && !b) { out.push_back(i); in_token = true; }
    }
    if (in_token) out.push_back(n);
}

// Rust std `<Ipv6Addr as FromStr>` (library/core/src/net/parser.rs read_ipv6_addr), restricted to what
// can occur here: the candidate only contains [0-9A-Fa-f:], so the embedded-IPv4 branch never fires.
inline bool parse_ipv6_rust(const uint8_t* s, size_t n, uint16_t seg[8]) {
    size_t pos = 0;
    auto read_group = [&](uint16_t& g) -> bool {  // read_number(16, Some(4), true) — atomically
        size_t p = pos;
        uint32_t v = 0;
        int digits = 0;
        while (p < n && is_hex(s[p])) {
            uint8_t ch = s[p];
            uint32_t dv = is_digit(ch) ? ch - '0' : (uint32_t)((ch | 0x20) - 'a' + 10);
            v = v * 16 + dv;
            ++digits;
            ++p;
            if (digits > 4) return false;
        }
        if (digits == 0) return false;
        g = (uint16_t)v;
        pos = p;
        return true;
    };
    auto read_groups = [&](uint16_t* groups, size_t limit) -> size_t {
        for (size_t i = 0; i < limit; ++i) {
            size_t save = pos;
            if (i > 0) {  // read_separator(':', i, ..)
                if (pos < n && s[pos] == ':') ++pos;
                else { pos = save; return i; }
            }
            uint16_t g;
            if (!read_group(g)) { pos = save; return i; }
            groups[i] = g;
        }
        return limit;
    };
    uint16_t head[8] = {0};
    size_t head_size = read_groups(head, 8);
    if (head_size == 8) {
        if (pos != n) return false;
        memcpy(seg, head, sizeof(head));
        return true;
    }
    if (!(pos < n && s[pos] == ':')) return false;
    ++pos;
    if (!(pos < n && s[pos] == ':')) return false;
    ++pos;
    uint16_t tail[7] = {0};
    size_t limit = 8 - (head_size + 1);
    size_t tail_size = read_groups(tail, limit);
    if (pos != n) return false;
    for (size_t i = 0; i < tail_size; ++i) head[8 - tail_size + i] = tail[i];
    memcpy(seg, head, sizeof(head));
    return true;
}

inline bool eq_ignore_case(const uint8_t* a, const char* b, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        uint8_t x = a[i], y = (uint8_t)b[i];
        if (x >= 'A' && x <= 'Z') x += 32;
        if (y >= 'A' && y <= 'Z') y += 32;
        if (x != y) return false;
    }
    return true;
}
// is_ipv6_loopback_or_linklocal (lib.rs:1425-1456)
inline bool ipv6_loopback_or_linklocal(const uint8_t* c, size_t n) {
    if (n == 3 && memcmp(c, "::1", 3) == 0) return true;
    if (n >= 4) {
        if (eq_ignore_case(c, "fe80", 4)) return true;
        if (eq_ignore_case(c, "fe8", 3) || eq_ignore_case(c, "fe9", 3) || eq_ignore_case(c, "fea", 3) ||
            eq_ignore_case(c, "feb", 3))
            return true;
    }
    return false;
}

struct Extractor {
    const Psl* psl;
    uint32_t flags = EX_ALL;
    size_t min_domain_labels = 2;
    bool require_word_boundaries = true;

    // memchr::memmem::Finder::find_iter — NON-overlapping leftmost matches of a 2-byte needle
    static void find_iter2(const uint8_t* c, size_t n, uint8_t a, uint8_t b, std::vector<size_t>& out) {
        out.clear();
        size_t i = 0;
        while (i + 1 < n) {
            if (c[i] == a && c[i + 1] == b) { out.push_back(i); i += 2; }
            else ++i;
        }
    }

    // try_parse_ipv4 (lib.rs:813-869)
    bool try_parse_ipv4(const uint8_t* line, size_t len, size_t start, uint8_t oct[4], size_t& end) const {
        size_t pos = start;
        if (require_word_boundaries && start > 0 && !is_boundary(line[start - 1])) return false;
        for (int idx = 0; idx < 4; ++idx) {
            uint32_t v = 0;
            int digits = 0;
            size_t octet_start = pos;
            while (pos < len && is_digit(line[pos]) && digits < 3) {
                v = v * 10 + (line[pos] - '0');
                ++pos;
                ++digits;
            }
            if (digits == 0) return false;
            if (v > 255) return false;
            if (digits > 1 && line[octet_start] == '0') return false;
            oct[idx] = (uint8_t)v;
            if (idx < 3) {
                if (pos >= len || line[pos] != '.') return false;
                ++pos;
            }
        }
        if (require_word_boundaries && pos < len && !is_boundary(line[pos])) return false;
        end = pos;
        return true;
    }

    // extract_ipv6_chunk (lib.rs:1044-1116)
    void extract_ipv6(const uint8_t* c, size_t n, std::vector<Match>& out) const {
        size_t last_end = 0;
        std::vector<size_t> dcs;
        find_iter2(c, n, ':', ':', dcs);
        for (size_t dc : dcs) {
            if (dc < last_end) continue;
            bool hex_before = dc > 0 && is_hex(c[dc - 1]);
            bool hex_after = dc + 2 < n && is_hex(c[dc + 2]);
            if (!hex_before && !hex_after) { last_end = dc + 2; continue; }
            size_t start = dc;
            while (start > 0) {
                uint8_t ch = c[start - 1];
                if (!is_hex(ch) && ch != ':') break;
                --start;
            }
            size_t end = dc + 2;
            while (end < n) {
                uint8_t ch = c[end];
                if (!is_hex(ch) && ch != ':') break;
                ++end;
            }
            const uint8_t* cand = c + start;
            size_t clen = end - start;
            if (clen < 8) { last_end = end; continue; }
            if ((cand[0] == ':' && cand[1] == ':') || (cand[clen - 2] == ':' && cand[clen - 1] == ':')) { last_end = end; continue; }
            if (ipv6_loopback_or_linklocal(cand, clen)) { last_end = end; continue; }
            uint16_t seg[8];
            if (parse_ipv6_rust(cand, clen, seg)) {  // candidate is ASCII, from_utf8 always Ok
                Match m{};
                m.type = IT_IPV6; m.start = start; m.end = end;
                for (int i = 0; i < 8; ++i) { m.ip[2 * i] = (uint8_t)(seg[i] >> 8); m.ip[2 * i + 1] = (uint8_t)seg[i]; }
                out.push_back(m);
                last_end = end;
                continue;
            }
            last_end = dc + 2;
        }
    }

    // extract_ipv4_chunk_with_dots (lib.rs:1120-1179)
    void extract_ipv4(const uint8_t* c, size_t n, const std::vector<size_t>& dots, std::vector<Match>& out) const {
        size_t last_end = 0;
        for (size_t i = 0; i < dots.size(); ++i) {
            size_t dp = dots[i];
            if (dp == 0 || dp + 6 > n) continue;
            if (!is_digit(c[dp - 1]) || !is_digit(c[dp + 1])) continue;
            size_t start = dp;
            while (start > 0 && (is_digit(c[start - 1]) || c[start - 1] == '.')) --start;
            if (start < last_end) continue;
            size_t end_search = start + 15 < n ? start + 15 : n;
            size_t in_range = 0;
            for (size_t k = i; k < dots.size() && dots[k] < end_search; ++k) ++in_range;
            if (in_range < 3) continue;
            uint8_t oct[4];
            size_t end;
            if (try_parse_ipv4(c, n, start, oct, end)) {
                Match m{};
                m.type = IT_IPV4; m.start = start; m.end = end;
                memcpy(m.ip, oct, 4);
                out.push_back(m);
                last_end = end;
            }
        }
    }

    // extract_email_at (lib.rs:891-950)
    bool extract_email_at(const uint8_t* line, size_t len, size_t at, size_t& s, size_t& e) const {
        size_t start = at;
        while (start > 0 && is_email_local_char(line[start - 1])) --start;
        if (start == at) return false;
        if (require_word_boundaries && start > 0 && !is_boundary(line[start - 1])) return false;
        size_t end = at + 1;
        while (end < len && is_domain_char(line[end])) ++end;
        if (end == at + 1) return false;
        if (require_word_boundaries && end < len && !is_boundary(line[end])) return false;
        const uint8_t* local = line + start;
        size_t llen = at - start;
        const uint8_t* dom = line + at + 1;
        size_t dlen = end - at - 1;
        for (size_t i = 0; i + 1 < llen; ++i)
            if (local[i] == '.' && local[i + 1] == '.') return false;
        bool has_letter = false;
        for (size_t i = 0; i < llen; ++i) has_letter |= is_alpha(local[i]);
        if (!has_letter) return false;
        if (!memchr(dom, '.', dlen)) return false;
        if (find_valid_tld_suffix(*psl, dom, dlen) < 0) return false;
        s = start; e = end;
        return true;
    }
    void extract_emails(const uint8_t* c, size_t n, std::vector<Match>& out) const {  // :1182-1196
        for (size_t at = 0; at < n; ++at) {
            if (c[at] != '@') continue;
            size_t s, e;
            if (extract_email_at(c, n, at, s, e) && valid_utf8(c + s, e - s)) {
                Match m{}; m.type = IT_EMAIL; m.start = s; m.end = e;
                out.push_back(m);
            }
        }
    }

    // is_valid_domain / is_valid_label (lib.rs:637-689)
    bool is_valid_label(const uint8_t* l, size_t n) const {
        if (n == 0) return false;
        if (l[0] == '-' || l[n - 1] == '-') return false;
        return true;
    }
    bool is_valid_domain(const uint8_t* d, size_t n) const {
        size_t label_count = 0, label_start = 0;
        for (size_t i = 0; i < n; ++i) {
            if (d[i] == '.') {
                if (!is_valid_label(d + label_start, i - label_start)) return false;
                ++label_count;
                label_start = i + 1;
            }
        }
        if (!is_valid_label(d + label_start, n - label_start)) return false;
        ++label_count;
        return label_count >= min_domain_labels;
    }
    // extract_domains_chunk_with_dots (lib.rs:537-628)
    void extract_domains(const uint8_t* c, size_t n, const std::vector<size_t>& dots, std::vector<Match>& out) const {
        size_t last_domain_end = 0;
        for (size_t dp : dots) {
            if (dp < last_domain_end) continue;
            size_t start = dp;
            while (start > 0 && is_domain_char_fast(c[start - 1])) --start;
            size_t end = dp + 1;
            while (end < n && is_domain_char_fast(c[end])) ++end;
            if (start >= dp || end <= dp + 1) continue;
            const uint8_t* cand = c + start;
            size_t clen = end - start;
            long tld = find_valid_tld_suffix(*psl, cand, clen);
            if (tld < 0) continue;
            if (tld == 0) continue;
            if (require_word_boundaries) {
                if (start > 0 && !is_boundary(c[start - 1])) continue;
                if (end < n && !is_boundary(c[end])) continue;
            }
            if (is_valid_domain(cand, clen)) {
                if (!valid_utf8(cand, clen)) continue;
                Match m{}; m.type = IT_DOMAIN; m.start = start; m.end = end;
                out.push_back(m);
                last_domain_end = end;
            }
        }
    }

    static bool all_hex(const uint8_t* s, size_t n) {
        for (size_t i = 0; i < n; ++i) if (!is_hex(s[i])) return false;
        return true;
    }
    // extract_hashes_chunk_with_boundaries (lib.rs:1212-1250)
    void extract_hashes(const uint8_t* c, const std::vector<size_t>& b, std::vector<Match>& out) const {
        for (size_t i = 0; i + 1 < b.size(); i += 2) {
            size_t s = b[i], e = b[i + 1], len = e - s;
            int t = len == 32 ? IT_MD5 : len == 40 ? IT_SHA1 : len == 64 ? IT_SHA256 : len == 96 ? IT_SHA384 : len == 128 ? IT_SHA512 : -1;
            if (t < 0) continue;
            if (all_hex(c + s, len)) { Match m{}; m.type = (uint8_t)t; m.start = s; m.end = e; out.push_back(m); }
        }
    }
    // validate_bitcoin_base58 (lib.rs:1799-1822)
    static bool validate_bitcoin_base58(const uint8_t* s, size_t n) {
        std::vector<uint8_t> dec;
        if (!base58_decode(s, n, dec)) return false;
        if (dec.size() < 5) return false;
        uint8_t h1[32], h2[32];
        sha256(dec.data(), dec.size() - 4, h1);
        sha256(h1, 32, h2);
        return memcmp(h2, dec.data() + dec.size() - 4, 4) == 0;
    }
    // extract_bitcoin_chunk_with_boundaries (lib.rs:1269-1319)
    void extract_bitcoin(const uint8_t* c, const std::vector<size_t>& b, std::vector<Match>& out) const {
        for (size_t i = 0; i + 1 < b.size(); i += 2) {
            size_t s = b[i], e = b[i + 1], len = e - s;
            if (len < 26 || len > 62) continue;
            const uint8_t* cand = c + s;
            bool ok = false;
            if (len >= 3 && memcmp(cand, "bc1", 3) == 0) {
                ok = valid_utf8(cand, len) && bech32_decode_is_bc(cand, len);
            } else if (cand[0] == '1' || cand[0] == '3') {
                ok = valid_utf8(cand, len) && validate_bitcoin_base58(cand, len);
            }
            if (ok) { Match m{}; m.type = IT_BITCOIN; m.start = s; m.end = e; out.push_back(m); }
        }
    }
    // validate_ethereum_checksum (lib.rs:1840-1892); input already "0x" + 40 hex
    static bool validate_ethereum_checksum(const uint8_t* a) {
        const uint8_t* hex = a + 2;
        bool all_lower = true, all_upper = true;
        for (int i = 0; i < 40; ++i) {
            if (is_alpha(hex[i])) {
                if (!(hex[i] >= 'a' && hex[i] <= 'z')) all_lower = false;
                if (!(hex[i] >= 'A' && hex[i] <= 'Z')) all_upper = false;
            }
        }
        if (all_lower || all_upper) return true;
        uint8_t lower[40], hash[32];
        for (int i = 0; i < 40; ++i) lower[i] = (hex[i] >= 'A' && hex[i] <= 'Z') ? hex[i] + 32 : hex[i];
        keccak256(lower, 40, hash);
        for (int i = 0; i < 40; ++i) {
            if (is_alpha(hex[i])) {
                uint8_t hb = hash[i / 2];
                uint8_t nib = (i % 2 == 0) ? (hb >> 4) : (hb & 0x0f);
                bool should_upper = nib >= 8;
                bool is_upper = hex[i] >= 'A' && hex[i] <= 'Z';
                if (is_upper != should_upper) return false;
            }
        }
        return true;
    }
    // extract_ethereum_chunk (lib.rs:1328-1361)
    void extract_ethereum(const uint8_t* c, size_t n, std::vector<Match>& out) const {
        std::vector<size_t> oxs;
        find_iter2(c, n, '0', 'x', oxs);
        for (size_t start : oxs) {
            if (start + 42 > n) continue;
            if (require_word_boundaries && start > 0 && !is_boundary(c[start - 1])) continue;
            size_t end = start + 42;
            if (require_word_boundaries && end < n && !is_boundary(c[end])) continue;
            if (!all_hex(c + start + 2, 40)) continue;
            if (validate_ethereum_checksum(c + start)) {
                Match m{}; m.type = IT_ETHEREUM; m.start = start; m.end = end; out.push_back(m);
            }
        }
    }
    // validate_monero_address (lib.rs:1895-1920)
    static bool validate_monero(const uint8_t* s, size_t n) {
        std::vector<uint8_t> dec;
        if (!base58_decode(s, n, dec)) return false;
        if (dec.size() < 5) return false;
        uint8_t h[32];
        keccak256(dec.data(), dec.size() - 4, h);
        return memcmp(h, dec.data() + dec.size() - 4, 4) == 0;
    }
    // extract_monero_chunk_with_boundaries (lib.rs:1367-1409)
    void extract_monero(const uint8_t* c, const std::vector<size_t>& b, std::vector<Match>& out) const {
        for (size_t i = 0; i + 1 < b.size(); i += 2) {
            size_t s = b[i], e = b[i + 1], len = e - s;
            if (len < 90 || len > 110) continue;
            const uint8_t* cand = c + s;
            if (cand[0] != '4' && cand[0] != '8') continue;
            if (valid_utf8(cand, len) && validate_monero(cand, len)) {
                Match m{}; m.type = IT_MONERO; m.start = s; m.end = e; out.push_back(m);
            }
        }
    }

    // extract_from_chunk (lib.rs:409-488) — fixed order IPv6, IPv4, email, domain, hash, BTC, ETH, XMR
    void extract_from_chunk(const uint8_t* c, size_t n, std::vector<Match>& out) const {
        std::vector<size_t> boundaries, dots;
        if (flags & (EX_HASHES | EX_BITCOIN | EX_MONERO)) find_word_boundaries(c, n, boundaries);
        if (flags & (EX_IPV4 | EX_DOMAINS))
            for (size_t i = 0; i < n; ++i) if (c[i] == '.') dots.push_back(i);
        if (flags & EX_IPV6) extract_ipv6(c, n, out);
        if (flags & EX_IPV4) extract_ipv4(c, n, dots, out);
        if (flags & EX_EMAILS) extract_emails(c, n, out);
        if (flags & EX_DOMAINS) extract_domains(c, n, dots, out);
        if (flags & EX_HASHES) extract_hashes(c, boundaries, out);
        if (flags & EX_BITCOIN) extract_bitcoin(c, boundaries, out);
        if (flags & EX_ETHEREUM) extract_ethereum(c, n, out);
        if (flags & EX_MONERO) extract_monero(c, boundaries, out);
    }
};

}  // namespace orc
