// Host-side IP address text <-> binary helpers with Rust std semantics (the reference parses keys with
// `str::parse::<IpAddr>()`, crates/matchy-format/src/mmdb_builder.rs:338-365, and prints with Display).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace mxy {

struct IpAddr {
    bool v6 = false;
    uint8_t b[16] = {0};  // v4: b[0..4]
};

namespace detail {
struct AddrParser {
    const char* s;
    size_t n, pos = 0;
    // read_number(radix, max_digits, allow_zero_prefix) — atomic
    bool number(int radix, int max_digits, bool allow_zero_prefix, uint32_t maxval, uint32_t& out) {
        size_t p = pos;
        uint32_t v = 0;
        int digits = 0;
        bool leading_zero = p < n && s[p] == '0';
        while (p < n) {
            char c = s[p];
            int d;
            if (c >= '0' && c <= '9') d = c - '0';
            else if (radix == 16 && c >= 'a' && c <= 'f') d = c - 'a' + 10;
            else if (radix == 16 && c >= 'A' && c <= 'F') d = c - 'A' + 10;
            else break;
            v = v * radix + d;
            ++digits;
            ++p;
            if (digits > max_digits) return false;
        }
        if (digits == 0) return false;
        if (!allow_zero_prefix && leading_zero && digits > 1) return false;
        if (v > maxval) return false;
        out = v;
        pos = p;
        return true;
    }
    bool ipv4(uint8_t o[4]) {
        size_t save = pos;
        for (int i = 0; i < 4; ++i) {
            if (i > 0) {
                if (pos < n && s[pos] == '.') ++pos;
                else { pos = save; return false; }
            }
            uint32_t v;
            if (!number(10, 3, false, 255, v)) { pos = save; return false; }
            o[i] = (uint8_t)v;
        }
        return true;
    }
    // read_groups: returns (count, ends_with_ipv4)
    size_t groups(uint16_t* g, size_t limit, bool& v4) {
        v4 = false;
        for (size_t i = 0; i < limit; ++i) {
            if (i + 1 < limit) {  // try a trailing embedded IPv4 (needs two groups)
                size_t save = pos;
                bool ok = true;
                if (i > 0) { if (pos < n && s[pos] == ':') ++pos; else ok = false; }
                uint8_t o[4];
                if (ok && ipv4(o)) {
                    g[i] = (uint16_t)((o[0] << 8) | o[1]);
                    g[i + 1] = (uint16_t)((o[2] << 8) | o[3]);
                    v4 = true;
                    return i + 2;
                }
                pos = save;
            }
            size_t save = pos;
            if (i > 0) { if (pos < n && s[pos] == ':') ++pos; else { pos = save; return i; } }
            uint32_t v;
            if (!number(16, 4, true, 0xFFFF, v)) { pos = save; return i; }
            g[i] = (uint16_t)v;
        }
        return limit;
    }
    bool ipv6(uint16_t seg[8]) {
        uint16_t head[8] = {0};
        bool head_v4;
        size_t hs = groups(head, 8, head_v4);
        if (hs == 8) { memcpy(seg, head, 16); return true; }
        if (head_v4) return false;
        if (!(pos < n && s[pos] == ':')) return false;
        ++pos;
        if (!(pos < n && s[pos] == ':')) return false;
        ++pos;
        uint16_t tail[7] = {0};
        bool tv4;
        size_t limit = 8 - (hs + 1);
        size_t ts = groups(tail, limit, tv4);
        for (size_t i = 0; i < ts; ++i) head[8 - ts + i] = tail[i];
        memcpy(seg, head, 16);
        return true;
    }
};
}  // namespace detail

inline bool parse_ipv4(const char* s, size_t n, uint8_t out[4]) {
    detail::AddrParser p{s, n};
    return p.ipv4(out) && p.pos == n;
}
inline bool parse_ipv6(const char* s, size_t n, uint8_t out[16]) {
    detail::AddrParser p{s, n};
    uint16_t seg[8];
    if (!p.ipv6(seg) || p.pos != n) return false;
    for (int i = 0; i < 8; ++i) { out[2 * i] = (uint8_t)(seg[i] >> 8); out[2 * i + 1] = (uint8_t)seg[i]; }
    return true;
}
inline bool parse_ip(const char* s, size_t n, IpAddr& a) {
    a = IpAddr();
    if (parse_ipv4(s, n, a.b)) { a.v6 = false; return true; }
    if (parse_ipv6(s, n, a.b)) { a.v6 = true; return true; }
    return false;
}

inline std::string format_ipv4(const uint8_t a[4]) {
    char b[20];
    snprintf(b, sizeof(b), "%u.%u.%u.%u", a[0], a[1], a[2], a[3]);
    return b;
}
// <Ipv6Addr as Display>: "::ffff:a.b.c.d" for IPv4-mapped, otherwise RFC 5952 (longest zero run >= 2 compressed, first wins)
inline std::string format_ipv6(const uint8_t a[16]) {
    uint16_t seg[8];
    for (int i = 0; i < 8; ++i) seg[i] = (uint16_t)((a[2 * i] << 8) | a[2 * i + 1]);
    if (!seg[0] && !seg[1] && !seg[2] && !seg[3] && !seg[4] && seg[5] == 0xffff) return "::ffff:" + format_ipv4(a + 12);
    int bs = 0, bl = 0, cs = 0, cl = 0;
    for (int i = 0; i < 8; ++i) {
        if (seg[i] == 0) { if (cl == 0) cs = i; ++cl; if (cl > bl) { bl = cl; bs = cs; } }
        else cl = 0;
    }
    auto sub = [&](int lo, int hi) {
        std::string r;
        char b[8];
        for (int i = lo; i < hi; ++i) { if (i > lo) r.push_back(':'); snprintf(b, sizeof(b), "%x", seg[i]); r += b; }
        return r;
    };
    if (bl > 1) return sub(0, bs) + "::" + sub(bs + bl, 8);
    return sub(0, 8);
}
inline std::string format_ip(const IpAddr& a) { return a.v6 ? format_ipv6(a.b) : format_ipv4(a.b); }

// format_cidr_into (crates/matchy/src/bin/cli_utils.rs:107-141)
inline std::string format_cidr(const IpAddr& a, unsigned prefix) {
    IpAddr net = a;
    int nb = a.v6 ? 16 : 4;
    for (int i = 0; i < nb; ++i) {
        int left = (int)prefix - i * 8;
        uint8_t mask = left >= 8 ? 0xFF : left <= 0 ? 0 : (uint8_t)(0xFF << (8 - left));
        net.b[i] &= mask;
    }
    return format_ip(net) + "/" + std::to_string(prefix);
}

}  // namespace mxy
