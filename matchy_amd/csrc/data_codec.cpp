#include "data_codec.h"

#include <algorithm>
#include <cerrno>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace mxy {

// ------------------------------------------------------------------------------------------------ encoder
namespace {

void put_size(uint8_t type_id, size_t size, std::vector<uint8_t>& out) {
    uint8_t tb = (uint8_t)(type_id << 5);
    if (size < 29) {
        out.push_back(tb | (uint8_t)size);
    } else if (size < 29 + 256) {
        out.push_back(tb | 29);
        out.push_back((uint8_t)(size - 29));
    } else if (size < 29 + 256 + 65536) {
        out.push_back(tb | 30);
        size_t a = size - 29 - 256;
        out.push_back((uint8_t)(a >> 8));
        out.push_back((uint8_t)a);
    } else {
        out.push_back(tb | 31);
        size_t a = size - 29 - 256 - 65536;
        out.push_back((uint8_t)(a >> 16));
        out.push_back((uint8_t)(a >> 8));
        out.push_back((uint8_t)a);
    }
}

// Arrays are an extended type: the control byte carries type 0 + size, then the extended-type byte.
void put_array_header(size_t size, std::vector<uint8_t>& out) {
    put_size(0, size, out);
    out.push_back(0x04);
}

void put_be(uint64_t v, int nbytes, std::vector<uint8_t>& out) {
    for (int i = nbytes - 1; i >= 0; --i) out.push_back((uint8_t)(v >> (8 * i)));
}

void put_pointer(uint32_t off, std::vector<uint8_t>& out) {
    if (off < 2048) {
        out.push_back((uint8_t)(0x20 | ((off >> 8) & 7)));
        out.push_back((uint8_t)off);
    } else if (off < 2048 + 524288) {
        uint32_t a = off - 2048;
        out.push_back((uint8_t)(0x20 | (1 << 3) | ((a >> 16) & 7)));
        out.push_back((uint8_t)(a >> 8));
        out.push_back((uint8_t)a);
    } else if (off < 2048u + 524288u + 134217728u) {
        uint32_t a = off - 526336;
        out.push_back((uint8_t)(0x20 | (2 << 3) | ((a >> 24) & 7)));
        out.push_back((uint8_t)(a >> 16));
        out.push_back((uint8_t)(a >> 8));
        out.push_back((uint8_t)a);
    } else {
        out.push_back((uint8_t)(0x20 | (3 << 3)));
        put_be(off, 4, out);
    }
}

void put_string(const std::string& s, std::vector<uint8_t>& out) {
    put_size(2, s.size(), out);
    out.insert(out.end(), s.begin(), s.end());
}

void put_scalar(const DataValue& v, std::vector<uint8_t>& out) {
    switch (v.type) {
        case DataValue::POINTER: put_pointer((uint32_t)v.u, out); break;
        case DataValue::STRING: put_string(v.str, out); break;
        case DataValue::DOUBLE: {
            uint64_t bits;
            memcpy(&bits, &v.f64, 8);
            out.push_back(0x68);
            put_be(bits, 8, out);
            break;
        }
        case DataValue::BYTES:
            put_size(4, v.str.size(), out);
            out.insert(out.end(), v.str.begin(), v.str.end());
            break;
        case DataValue::UINT16: out.push_back(0xA2); put_be(v.u, 2, out); break;
        case DataValue::UINT32: out.push_back(0xC4); put_be(v.u, 4, out); break;
        case DataValue::INT32: out.push_back(0x04); out.push_back(0x01); put_be((uint32_t)v.i32, 4, out); break;
        case DataValue::UINT64: out.push_back(0x08); out.push_back(0x02); put_be(v.u, 8, out); break;
        case DataValue::UINT128: out.push_back(0x10); out.push_back(0x03); put_be(v.uhi, 8, out); put_be(v.u, 8, out); break;
        case DataValue::BOOL: out.push_back(v.u ? 0x01 : 0x00); out.push_back(0x07); break;
        case DataValue::FLOAT: {
            uint32_t bits;
            memcpy(&bits, &v.f32, 4);
            out.push_back(0x04);
            out.push_back(0x08);
            put_be(bits, 4, out);
            break;
        }
        default: break;
    }
}

}  // namespace

void DataEncoder::encode_plain(const DataValue& v, std::vector<uint8_t>& out) {
    if (v.type == DataValue::MAP) {
        put_size(7, v.map.size(), out);
        for (const auto& kv : v.map) {  // std::map iterates in key order, as the reference sorts pairs by key
            put_string(kv.first, out);
            encode_plain(kv.second, out);
        }
    } else if (v.type == DataValue::ARRAY) {
        put_array_header(v.arr.size(), out);
        for (const auto& e : v.arr) encode_plain(e, out);
    } else {
        put_scalar(v, out);
    }
}

void DataEncoder::encode_interned(const DataValue& v) {
    if (v.type == DataValue::STRING) {
        auto it = strings_.find(v.str);
        if (it != strings_.end()) {
            put_pointer(it->second, buf_);
        } else {
            uint32_t off = (uint32_t)buf_.size();
            put_string(v.str, buf_);
            strings_.emplace(v.str, off);
        }
    } else if (v.type == DataValue::MAP) {
        put_size(7, v.map.size(), buf_);
        for (const auto& kv : v.map) {
            auto it = strings_.find(kv.first);
            if (it != strings_.end()) {
                put_pointer(it->second, buf_);
            } else {
                uint32_t off = (uint32_t)buf_.size();
                put_string(kv.first, buf_);
                strings_.emplace(kv.first, off);
            }
            encode_interned(kv.second);
        }
    } else if (v.type == DataValue::ARRAY) {
        put_array_header(v.arr.size(), buf_);
        for (const auto& e : v.arr) encode_interned(e);
    } else {
        put_scalar(v, buf_);
    }
}

uint32_t DataEncoder::encode(const DataValue& v) {
    std::vector<uint8_t> tmp;
    encode_plain(v, tmp);
    std::string key((const char*)tmp.data(), tmp.size());
    auto it = dedup_.find(key);
    if (it != dedup_.end()) return it->second;
    uint32_t off = (uint32_t)buf_.size();
    encode_interned(v);
    dedup_.emplace(std::move(key), off);
    return off;
}

// ------------------------------------------------------------------------------------------------ decoder
namespace {

struct Dec {
    const uint8_t* b;
    size_t n;

    bool size(size_t& cur, uint8_t bits, size_t& out) const {
        if (bits <= 28) { out = bits; return true; }
        if (bits == 29) { if (cur >= n) return false; out = 29 + b[cur]; cur += 1; return true; }
        if (bits == 30) { if (cur + 2 > n) return false; out = 29 + 256 + (((size_t)b[cur] << 8) | b[cur + 1]); cur += 2; return true; }
        if (cur + 3 > n) return false;
        out = 29 + 256 + 65536 + (((size_t)b[cur] << 16) | ((size_t)b[cur + 1] << 8) | b[cur + 2]);
        cur += 3;
        return true;
    }
    bool uint(size_t& cur, uint8_t bits, size_t maxsz, uint64_t& hi, uint64_t& lo) const {
        size_t sz;
        if (!size(cur, bits, sz) || sz > maxsz || cur + sz > n) return false;
        hi = lo = 0;
        for (size_t k = 0; k < sz; ++k) { hi = (hi << 8) | (lo >> 56); lo = (lo << 8) | b[cur + k]; }
        cur += sz;
        return true;
    }
    bool at(size_t& cur, DataValue& v, int depth) const {
        if (depth > 64 || cur >= n) return false;
        uint8_t ctrl = b[cur++];
        uint8_t type = ctrl >> 5, payload = ctrl & 0x1f;
        if (type == 0) {
            if (cur >= n) return false;
            int t = 7 + b[cur++];
            if (t == 8) {
                size_t sz;
                if (!size(cur, payload, sz) || sz > 4 || cur + sz > n) return false;
                uint32_t val = (sz > 0 && (b[cur] & 0x80)) ? 0xFFFFFFFFu : 0;
                for (size_t k = 0; k < sz; ++k) val = (val << 8) | b[cur + k];
                cur += sz;
                v.type = DataValue::INT32; v.i32 = (int32_t)val;
                return true;
            }
            if (t == 9) { v.type = DataValue::UINT64; return uint(cur, payload, 8, v.uhi, v.u); }
            if (t == 10) { v.type = DataValue::UINT128; return uint(cur, payload, 16, v.uhi, v.u); }
            if (t == 11) {
                size_t cnt;
                if (!size(cur, payload, cnt)) return false;
                v.type = DataValue::ARRAY;
                for (size_t k = 0; k < cnt; ++k) {
                    DataValue e;
                    if (!at(cur, e, depth + 1)) return false;
                    v.arr.push_back(std::move(e));
                }
                return true;
            }
            if (t == 14) { v.type = DataValue::BOOL; v.u = payload != 0; return true; }
            if (t == 15) {
                if (payload != 4 || cur + 4 > n) return false;
                uint32_t bits = ((uint32_t)b[cur] << 24) | ((uint32_t)b[cur + 1] << 16) | ((uint32_t)b[cur + 2] << 8) | b[cur + 3];
                cur += 4;
                memcpy(&v.f32, &bits, 4);
                v.type = DataValue::FLOAT;
                return true;
            }
            return false;
        }
        if (type == 1) {
            uint8_t sb = (payload >> 3) & 3;
            uint32_t low3 = payload & 7, off;
            if (sb == 0) { if (cur >= n) return false; off = (low3 << 8) | b[cur]; cur += 1; }
            else if (sb == 1) { if (cur + 1 >= n) return false; off = 2048 + ((low3 << 16) | ((uint32_t)b[cur] << 8) | b[cur + 1]); cur += 2; }
            else if (sb == 2) { if (cur + 2 >= n) return false; off = 526336 + ((low3 << 24) | ((uint32_t)b[cur] << 16) | ((uint32_t)b[cur + 1] << 8) | b[cur + 2]); cur += 3; }
            else { if (cur + 3 >= n) return false; off = ((uint32_t)b[cur] << 24) | ((uint32_t)b[cur + 1] << 16) | ((uint32_t)b[cur + 2] << 8) | b[cur + 3]; cur += 4; }
            v.type = DataValue::POINTER; v.u = off;
            return true;
        }
        if (type == 2 || type == 4) {
            size_t sz;
            if (!size(cur, payload, sz) || cur + sz > n) return false;
            v.type = type == 2 ? DataValue::STRING : DataValue::BYTES;
            v.str.assign((const char*)b + cur, sz);
            cur += sz;
            return true;
        }
        if (type == 3) {
            if (cur + 8 > n) return false;
            uint64_t bits = 0;
            for (int k = 0; k < 8; ++k) bits = (bits << 8) | b[cur + k];
            cur += 8;
            memcpy(&v.f64, &bits, 8);
            v.type = DataValue::DOUBLE;
            return true;
        }
        if (type == 5) { v.type = DataValue::UINT16; return uint(cur, payload, 2, v.uhi, v.u); }
        if (type == 6) { v.type = DataValue::UINT32; return uint(cur, payload, 4, v.uhi, v.u); }
        // type 7: map
        size_t cnt;
        if (!size(cur, payload, cnt)) return false;
        v.type = DataValue::MAP;
        for (size_t k = 0; k < cnt; ++k) {
            DataValue key;
            if (!at(cur, key, depth + 1)) return false;
            std::string ks;
            if (key.type == DataValue::STRING) ks = std::move(key.str);
            else if (key.type == DataValue::POINTER) {
                DataValue kv;
                size_t c2 = (size_t)key.u;
                if (!at(c2, kv, depth + 1) || !resolve(kv, depth + 1) || kv.type != DataValue::STRING) return false;
                ks = std::move(kv.str);
            } else return false;
            DataValue val;
            if (!at(cur, val, depth + 1)) return false;
            v.map[ks] = std::move(val);
        }
        return true;
    }
    bool resolve(DataValue& v, int depth) const {
        if (depth > 64) return false;
        if (v.type == DataValue::POINTER) {
            size_t cur = (size_t)v.u;
            DataValue t;
            if (!at(cur, t, depth + 1) || !resolve(t, depth + 1)) return false;
            v = std::move(t);
            return true;
        }
        if (v.type == DataValue::MAP) for (auto& kv : v.map) if (!resolve(kv.second, depth + 1)) return false;
        if (v.type == DataValue::ARRAY) for (auto& e : v.arr) if (!resolve(e, depth + 1)) return false;
        return true;
    }
};

}  // namespace

bool decode_value(const uint8_t* section, size_t len, uint32_t offset, DataValue& out) {
    Dec d{section, len};
    size_t cur = offset;
    return d.at(cur, out, 0) && d.resolve(out, 0);
}

// ------------------------------------------------------------------------------------------------ JSON out
void json_escape(const std::string& s, std::string& out) {
    static const char* HEX = "0123456789abcdef";
    out.push_back('"');
    for (unsigned char ch : s) {
        switch (ch) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            default:
                if (ch < 0x20) { out += "\\u00"; out.push_back(HEX[ch >> 4]); out.push_back(HEX[ch & 15]); }
                else out.push_back((char)ch);
        }
    }
    out.push_back('"');
}

static std::string fmt_f64(double d) {
    if (!std::isfinite(d)) return "null";
    char b[64];
    for (int p = 1; p <= 17; ++p) {
        snprintf(b, sizeof(b), "%.*g", p, d);
        if (strtod(b, nullptr) == d) break;
    }
    std::string s(b);
    if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
    return s;
}

void to_json(const DataValue& v, std::string& out) {
    switch (v.type) {
        case DataValue::STRING: json_escape(v.str, out); break;
        case DataValue::DOUBLE: out += fmt_f64(v.f64); break;
        case DataValue::FLOAT: out += fmt_f64((double)v.f32); break;
        case DataValue::BYTES:
            out.push_back('[');
            for (size_t i = 0; i < v.str.size(); ++i) { if (i) out.push_back(','); out += std::to_string((unsigned)(uint8_t)v.str[i]); }
            out.push_back(']');
            break;
        case DataValue::UINT16: case DataValue::UINT32: case DataValue::UINT64: out += std::to_string(v.u); break;
        case DataValue::UINT128: {
            unsigned __int128 x = ((unsigned __int128)v.uhi << 64) | v.u;
            std::string r;
            if (x == 0) r = "0";
            while (x) { r.push_back((char)('0' + (int)(x % 10))); x /= 10; }
            std::reverse(r.begin(), r.end());
            out.push_back('"'); out += r; out.push_back('"');
            break;
        }
        case DataValue::INT32: out += std::to_string(v.i32); break;
        case DataValue::BOOL: out += v.u ? "true" : "false"; break;
        case DataValue::MAP: {
            out.push_back('{');
            bool first = true;
            for (const auto& kv : v.map) {
                if (!first) out.push_back(',');
                first = false;
                json_escape(kv.first, out);
                out.push_back(':');
                to_json(kv.second, out);
            }
            out.push_back('}');
            break;
        }
        case DataValue::ARRAY:
            out.push_back('[');
            for (size_t i = 0; i < v.arr.size(); ++i) { if (i) out.push_back(','); to_json(v.arr[i], out); }
            out.push_back(']');
            break;
        case DataValue::POINTER: out += "\"<pointer>\""; break;
    }
}

// ------------------------------------------------------------------------------------------------ JSON in
namespace {

struct JP {
    const char* p;
    const char* e;
    NumberTyping typing;
    std::string err;

    void ws() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }

    static void put_utf8(uint32_t cp, std::string& s) {
        if (cp < 0x80) s.push_back((char)cp);
        else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
        else { s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    }
    bool hex4(uint32_t& v) {
        if (e - p < 4) return false;
        v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0';
            else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else return false;
        }
        return true;
    }
    bool string(std::string& s) {
        if (p >= e || *p != '"') return fail("expected string");
        ++p;
        while (p < e) {
            unsigned char c = (unsigned char)*p++;
            if (c == '"') return true;
            if (c < 0x20) return fail("control character in string");
            if (c != '\\') { s.push_back((char)c); continue; }
            if (p >= e) break;
            char x = *p++;
            switch (x) {
                case '"': s.push_back('"'); break; case '\\': s.push_back('\\'); break; case '/': s.push_back('/'); break;
                case 'b': s.push_back('\b'); break; case 'f': s.push_back('\f'); break; case 'n': s.push_back('\n'); break;
                case 'r': s.push_back('\r'); break; case 't': s.push_back('\t'); break;
                case 'u': {
                    uint32_t cp;
                    if (!hex4(cp)) return fail("bad \\u escape");
                    if (cp >= 0xD800 && cp <= 0xDBFF) {
                        uint32_t lo;
                        if (e - p < 6 || p[0] != '\\' || p[1] != 'u') return fail("lone surrogate");
                        p += 2;
                        if (!hex4(lo) || lo < 0xDC00 || lo > 0xDFFF) return fail("bad surrogate pair");
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    } else if (cp >= 0xDC00 && cp <= 0xDFFF) return fail("lone surrogate");
                    put_utf8(cp, s);
                    break;
                }
                default: return fail("bad escape");
            }
        }
        return fail("unterminated string");
    }
    bool number(DataValue& v) {
        const char* s = p;
        bool neg = false, is_float = false;
        if (p < e && *p == '-') { neg = true; ++p; }
        if (p >= e || !(*p >= '0' && *p <= '9')) return fail("bad number");
        if (*p == '0') ++p; else while (p < e && *p >= '0' && *p <= '9') ++p;
        if (p < e && *p == '.') { is_float = true; ++p; if (p >= e || !(*p >= '0' && *p <= '9')) return fail("bad number"); while (p < e && *p >= '0' && *p <= '9') ++p; }
        if (p < e && (*p == 'e' || *p == 'E')) {
            is_float = true; ++p;
            if (p < e && (*p == '+' || *p == '-')) ++p;
            if (p >= e || !(*p >= '0' && *p <= '9')) return fail("bad number");
            while (p < e && *p >= '0' && *p <= '9') ++p;
        }
        std::string tok(s, p - s);
        if (!is_float) {
            errno = 0;
            if (!neg) {
                unsigned long long u = strtoull(tok.c_str(), nullptr, 10);
                if (errno == 0) {
                    if (typing == NumberTyping::CLI) {
                        if (u <= (unsigned long long)INT64_MAX) v = DataValue::Int32((int32_t)(int64_t)u);  // `i as i32` truncation
                        else v = DataValue::Uint64(u);
                    } else {
                        if (u <= 0xFFFF) v = DataValue::Uint16((uint16_t)u);
                        else if (u <= 0xFFFFFFFFull) v = DataValue::Uint32((uint32_t)u);
                        else v = DataValue::Uint64(u);
                    }
                    return true;
                }
            } else {
                long long i = strtoll(tok.c_str(), nullptr, 10);
                if (errno == 0) {
                    if (typing == NumberTyping::CLI) v = DataValue::Int32((int32_t)i);
                    else if (i >= INT32_MIN) v = DataValue::Int32((int32_t)i);
                    else v = DataValue::Double((double)i);
                    return true;
                }
            }
        }
        v = DataValue::Double(strtod(tok.c_str(), nullptr));
        return true;
    }
    bool value(DataValue& v, int depth) {
        if (depth > 128) return fail("nesting too deep");
        ws();
        if (p >= e) return fail("unexpected end");
        char c = *p;
        if (c == '{') {
            ++p;
            v = DataValue::Map();
            ws();
            if (p < e && *p == '}') { ++p; return true; }
            for (;;) {
                ws();
                std::string k;
                if (!string(k)) return false;
                ws();
                if (p >= e || *p != ':') return fail("expected ':'");
                ++p;
                DataValue x;
                if (!value(x, depth + 1)) return false;
                v.map[k] = std::move(x);
                ws();
                if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == '}') { ++p; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++p;
            v = DataValue::Array();
            ws();
            if (p < e && *p == ']') { ++p; return true; }
            for (;;) {
                DataValue x;
                if (!value(x, depth + 1)) return false;
                v.arr.push_back(std::move(x));
                ws();
                if (p < e && *p == ',') { ++p; continue; }
                if (p < e && *p == ']') { ++p; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.type = DataValue::STRING; v.str.clear(); return string(v.str); }
        if (c == 't' && e - p >= 4 && memcmp(p, "true", 4) == 0) { p += 4; v = DataValue::Bool(true); return true; }
        if (c == 'f' && e - p >= 5 && memcmp(p, "false", 5) == 0) { p += 5; v = DataValue::Bool(false); return true; }
        if (c == 'n' && e - p >= 4 && memcmp(p, "null", 4) == 0) {
            p += 4;
            if (typing == NumberTyping::CLI) { v = DataValue(); v.type = DataValue::BYTES; return true; }  // cli_utils.rs:206
            return fail("null is not a valid data value");
        }
        return number(v);
    }
};

}  // namespace

bool parse_json(const char* text, size_t len, NumberTyping typing, DataValue& out, std::string& err) {
    JP jp{text, text + len, typing, {}};
    if (!jp.value(out, 0)) { err = jp.err; return false; }
    jp.ws();
    if (jp.p != jp.e) { err = "trailing characters"; return false; }
    return true;
}

}  // namespace mxy
