// ORACLE (test infrastructure): Rust `str::to_lowercase` restated for the case-insensitive mode of the reference
// (alloc/src/str.rs to_lowercase + map_uppercase_sigma / case_ignorable_then_cased; call sites
// matchy-literal-hash/src/lib.rs:162-165,469-472). The character data is the same file the product loads
// (matchy_amd/data/lowercase.bin, tools/gen_lowercase.py); the code is written independently of matchy_amd/csrc/unicode_lower.cpp.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace orc {

struct Lowercase {
    std::map<uint32_t, std::string> lower;
    std::vector<std::pair<uint32_t, uint32_t>> ign, cased;

    static Lowercase& table() {
        static Lowercase t;
        return t;
    }
    bool loaded = false;
    void load(const char* path) {
        FILE* f = fopen(path, "rb");
        if (!f) throw std::runtime_error(std::string("oracle: cannot open ") + path);
        std::vector<uint8_t> b;
        uint8_t tmp[65536];
        size_t n;
        while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) b.insert(b.end(), tmp, tmp + n);
        fclose(f);
        if (b.size() < 20 || memcmp(b.data(), "LCTB", 4) != 0) throw std::runtime_error("oracle: bad lowercase table");
        auto u32 = [&](size_t o) { uint32_t v; memcpy(&v, b.data() + o, 4); return v; };
        size_t nm = u32(8), ni = u32(12), nc = u32(16), o = 20;
        for (size_t i = 0; i < nm; ++i, o += 12) lower[u32(o)] = std::string((const char*)b.data() + o + 5, b[o + 4]);
        for (size_t i = 0; i < ni; ++i, o += 8) ign.push_back({u32(o), u32(o + 4)});
        for (size_t i = 0; i < nc; ++i, o += 8) cased.push_back({u32(o), u32(o + 4)});
        loaded = true;
    }
    static bool has(const std::vector<std::pair<uint32_t, uint32_t>>& r, uint32_t c) {
        for (auto& p : r) if (c >= p.first && c <= p.second) return true;
        return false;
    }
    // chars of a valid UTF-8 string
    static std::vector<uint32_t> chars(const std::string& s) {
        std::vector<uint32_t> out;
        for (size_t i = 0; i < s.size();) {
            uint8_t c = (uint8_t)s[i];
            size_t n = c < 0x80 ? 1 : c < 0xE0 ? 2 : c < 0xF0 ? 3 : 4;
            uint32_t cp = n == 1 ? c : n == 2 ? (c & 0x1F) : n == 3 ? (c & 0x0F) : (c & 0x07);
            for (size_t k = 1; k < n && i + k < s.size(); ++k) cp = (cp << 6) | ((uint8_t)s[i + k] & 0x3F);
            out.push_back(cp);
            i += n;
        }
        return out;
    }
    static void push_utf8(uint32_t cp, std::string& o) {
        if (cp < 0x80) o.push_back((char)cp);
        else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else { o.push_back((char)(0xF0 | (cp >> 18))); o.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    }
    std::string to_lowercase(const std::string& s) const {
        if (!loaded) throw std::runtime_error("oracle: lowercase table not loaded");
        const std::vector<uint32_t> cs = chars(s);
        std::string out;
        for (size_t i = 0; i < cs.size(); ++i) {
            const uint32_t c = cs[i];
            if (c == 0x3A3) {
                // Final_Sigma: preceded by a cased letter and not followed by one, case-ignorable characters skipped
                bool before = false, after = false;
                for (size_t j = i; j-- > 0;) { if (has(ign, cs[j])) continue; before = has(cased, cs[j]); break; }
                for (size_t j = i + 1; j < cs.size(); ++j) { if (has(ign, cs[j])) continue; after = has(cased, cs[j]); break; }
                push_utf8(before && !after ? 0x3C2 : 0x3C3, out);
                continue;
            }
            auto it = lower.find(c);
            if (it != lower.end()) out += it->second;
            else push_utf8(c, out);
        }
        return out;
    }
};

}  // namespace orc
