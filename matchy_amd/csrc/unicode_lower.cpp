#include "unicode_lower.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace mxy {

namespace {

std::string module_dir() {
    Dl_info info;
    if (dladdr((void*)&module_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t k = p.rfind('/');
        if (k != std::string::npos) return p.substr(0, k);
    }
    return ".";
}
bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    bool ok = n >= 0 && fread(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}
uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

bool in_ranges(const std::vector<CpRange>& r, uint32_t cp) {
    size_t lo = 0, hi = r.size();
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (r[mid].last < cp) lo = mid + 1; else hi = mid; }
    return lo < r.size() && r[lo].first <= cp;
}
// one character of valid UTF-8 at s[i]; returns its length
size_t decode(const std::string& s, size_t i, uint32_t& cp) {
    const uint8_t c = (uint8_t)s[i];
    if (c < 0x80) { cp = c; return 1; }
    if (c < 0xE0) { cp = ((c & 0x1Fu) << 6) | ((uint8_t)s[i + 1] & 0x3Fu); return 2; }
    if (c < 0xF0) { cp = ((c & 0x0Fu) << 12) | (((uint8_t)s[i + 1] & 0x3Fu) << 6) | ((uint8_t)s[i + 2] & 0x3Fu); return 3; }
    cp = ((c & 0x07u) << 18) | (((uint8_t)s[i + 1] & 0x3Fu) << 12) | (((uint8_t)s[i + 2] & 0x3Fu) << 6) | ((uint8_t)s[i + 3] & 0x3Fu);
    return 4;
}

}  // namespace

const LowerTable& LowerTable::get() {
    static const LowerTable* inst = [] {
        std::vector<std::string> candidates;
        if (const char* e = getenv("MATCHY_AMD_LOWERCASE")) candidates.push_back(e);
        const std::string d = module_dir();
        candidates.push_back(d + "/../data/lowercase.bin");
        candidates.push_back(d + "/data/lowercase.bin");
        candidates.push_back(d + "/lowercase.bin");
        std::vector<uint8_t> buf;
        bool ok = false;
        for (auto& c : candidates) if (read_file(c, buf)) { ok = true; break; }
        if (!ok) throw std::runtime_error("matchy_amd: cannot find lowercase.bin (set MATCHY_AMD_LOWERCASE)");
        if (buf.size() < 20 || memcmp(buf.data(), "LCTB", 4) != 0) throw std::runtime_error("matchy_amd: bad lowercase.bin header");
        auto* t = new LowerTable();
        t->unicode_version = rd32(buf.data() + 4);
        const size_t nm = rd32(buf.data() + 8), ni = rd32(buf.data() + 12), nc = rd32(buf.data() + 16);
        if (buf.size() != 20 + nm * 12 + (ni + nc) * 8) throw std::runtime_error("matchy_amd: truncated lowercase.bin");
        t->map.resize(nm);
        if (nm) memcpy(t->map.data(), buf.data() + 20, nm * 12);
        const uint8_t* p = buf.data() + 20 + nm * 12;
        for (size_t i = 0; i < ni; ++i, p += 8) t->ignorable.push_back({rd32(p), rd32(p + 4)});
        for (size_t i = 0; i < nc; ++i, p += 8) t->cased.push_back({rd32(p), rd32(p + 4)});
        return t;
    }();
    return *inst;
}

std::string LowerTable::to_lowercase(const std::string& s) const {
    std::string out;
    out.reserve(s.size());
    // case_ignorable_then_cased (alloc/src/str.rs): first character that is not Case_Ignorable, is it Cased?
    auto cased_behind = [&](size_t i) {   // characters of s[..i], right to left
        while (i > 0) {
            size_t j = i - 1;
            while (j > 0 && ((uint8_t)s[j] & 0xC0) == 0x80) --j;
            uint32_t cp;
            decode(s, j, cp);
            if (!in_ranges(ignorable, cp)) return in_ranges(cased, cp);
            i = j;
        }
        return false;
    };
    auto cased_ahead = [&](size_t i) {
        while (i < s.size()) {
            uint32_t cp;
            const size_t n = decode(s, i, cp);
            if (!in_ranges(ignorable, cp)) return in_ranges(cased, cp);
            i += n;
        }
        return false;
    };
    for (size_t i = 0; i < s.size();) {
        uint32_t cp;
        const size_t n = decode(s, i, cp);
        if (cp < 0x80) {
            out.push_back((char)((cp >= 'A' && cp <= 'Z') ? cp + 32 : cp));
        } else if (cp == 0x3A3) {
            const bool final_sigma = cased_behind(i) && !cased_ahead(i + n);
            out += final_sigma ? "\xCF\x82" : "\xCF\x83";
        } else {
            auto it = std::lower_bound(map.begin(), map.end(), cp, [](const LowerMapEntry& e, uint32_t v) { return e.cp < v; });
            if (it != map.end() && it->cp == cp) out.append((const char*)it->utf8, it->len);
            else out.append(s, i, n);
        }
        i += n;
    }
    return out;
}

}  // namespace mxy
