// oracle/database.h — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product library.
//
// Literal restatement of the reference's read path over a .mxy image:
//   section discovery   crates/matchy/src/database.rs:649-713, 1023-1069, 1218-1415
//   metadata / marker   crates/matchy-format/src/mmdb/format.rs:31-171
//   data decoder        crates/matchy-data-format/src/lib.rs:635-1048
//   IP trie walk        crates/matchy-format/src/mmdb/tree.rs:38-277
//   literal hash        crates/matchy-literal-hash/src/lib.rs:380-575
//   paraglob find_all   crates/matchy-paraglob/src/paraglob_offset.rs:1028-1639
//   ACLH lookup         crates/matchy-paraglob/src/literal_hash.rs:263-333
//   dispatch            crates/matchy/src/database.rs:810-981
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "crypto.h"
#include "unicode_lower.h"

namespace orc {

// ------------------------------------------------------------------ DataValue tree
struct Value {
    enum Kind { POINTER = 1, STRING = 2, DOUBLE = 3, BYTES = 4, UINT16 = 5, UINT32 = 6, MAP = 7, INT32 = 8, UINT64 = 9,
                UINT128 = 10, ARRAY = 11, BOOL = 14, FLOAT = 15 } kind = STRING;
    std::string s;           // STRING / BYTES
    uint64_t u = 0, uhi = 0; // integers (uhi for UINT128), BOOL, POINTER
    int32_t i = 0;
    double d = 0;
    float f = 0;
    std::map<std::string, Value> map;  // MAP (sorted: serde_json without preserve_order = BTreeMap)
    std::vector<Value> arr;
};

struct Decoder {
    const uint8_t* buf;
    size_t len;
    bool fail = false;

    bool decode_size(size_t& cur, uint8_t bits, size_t& out) {
        if (bits <= 28) { out = bits; return true; }
        if (bits == 29) { if (cur >= len) return false; out = 29 + buf[cur]; cur += 1; return true; }
        if (bits == 30) { if (cur + 2 > len) return false; out = 29 + 256 + (((size_t)buf[cur] << 8) | buf[cur + 1]); cur += 2; return true; }
        if (cur + 3 > len) return false;
        out = 29 + 256 + 65536 + (((size_t)buf[cur] << 16) | ((size_t)buf[cur + 1] << 8) | buf[cur + 2]);
        cur += 3;
        return true;
    }
    bool decode_uint(size_t& cur, uint8_t bits, size_t maxsz, uint64_t& hi, uint64_t& lo) {
        size_t sz;
        if (!decode_size(cur, bits, sz) || sz > maxsz || cur + sz > len) return false;
        hi = 0; lo = 0;
        for (size_t k = 0; k < sz; ++k) { hi = (hi << 8) | (lo >> 56); lo = (lo << 8) | buf[cur + k]; }
        cur += sz;
        return true;
    }
    // decode_at (lib.rs:665-687)
    bool decode_at(size_t& cur, Value& v, int depth = 0) {
        if (depth > 64 || cur >= len) return false;
        uint8_t ctrl = buf[cur++];
        uint8_t type = ctrl >> 5, payload = ctrl & 0x1f;
        switch (type) {
            case 0: {  // extended (lib.rs:689-719)
                if (cur >= len) return false;
                uint8_t t = 7 + buf[cur++];
                switch (t) {
                    case 8: {
                        size_t sz;
                        if (!decode_size(cur, payload, sz) || sz > 4 || cur + sz > len) return false;
                        int32_t val = 0;
                        if (sz > 0) {
                            if (buf[cur] & 0x80) val = -1;
                            for (size_t k = 0; k < sz; ++k) val = (int32_t)(((uint32_t)val << 8) | buf[cur + k]);
                        }
                        cur += sz;
                        v.kind = Value::INT32; v.i = val; return true;
                    }
                    case 9: v.kind = Value::UINT64; return decode_uint(cur, payload, 8, v.uhi, v.u);
                    case 10: v.kind = Value::UINT128; return decode_uint(cur, payload, 16, v.uhi, v.u);
                    case 11: {
                        size_t cnt;
                        if (!decode_size(cur, payload, cnt)) return false;
                        v.kind = Value::ARRAY;
                        for (size_t k = 0; k < cnt; ++k) { Value e; if (!decode_at(cur, e, depth + 1)) return false; v.arr.push_back(std::move(e)); }
                        return true;
                    }
                    case 14: v.kind = Value::BOOL; v.u = payload != 0; return true;
                    case 15: {
                        if (payload != 4 || cur + 4 > len) return false;
                        uint32_t bits = ((uint32_t)buf[cur] << 24) | ((uint32_t)buf[cur + 1] << 16) | ((uint32_t)buf[cur + 2] << 8) | buf[cur + 3];
                        cur += 4;
                        memcpy(&v.f, &bits, 4); v.kind = Value::FLOAT; return true;
                    }
                    default: return false;
                }
            }
            case 1: {  // pointer (lib.rs:721-771)
                uint8_t sb = (payload >> 3) & 3;
                uint32_t low3 = payload & 7, off;
                if (sb == 0) { if (cur >= len) return false; off = (low3 << 8) | buf[cur]; cur += 1; }
                else if (sb == 1) { if (cur + 1 >= len) return false; off = 2048 + ((low3 << 16) | ((uint32_t)buf[cur] << 8) | buf[cur + 1]); cur += 2; }
                else if (sb == 2) { if (cur + 2 >= len) return false; off = 526336 + ((low3 << 24) | ((uint32_t)buf[cur] << 16) | ((uint32_t)buf[cur + 1] << 8) | buf[cur + 2]); cur += 3; }
                else { if (cur + 3 >= len) return false; off = ((uint32_t)buf[cur] << 24) | ((uint32_t)buf[cur + 1] << 16) | ((uint32_t)buf[cur + 2] << 8) | buf[cur + 3]; cur += 4; }
                v.kind = Value::POINTER; v.u = off; return true;
            }
            case 2: case 4: {
                size_t sz;
                if (!decode_size(cur, payload, sz) || cur + sz > len) return false;
                v.kind = type == 2 ? Value::STRING : Value::BYTES;
                v.s.assign((const char*)buf + cur, sz);
                cur += sz;
                return true;
            }
            case 3: {
                if (cur + 8 > len) return false;
                uint64_t bits = 0;
                for (int k = 0; k < 8; ++k) bits = (bits << 8) | buf[cur + k];
                cur += 8;
                memcpy(&v.d, &bits, 8); v.kind = Value::DOUBLE; return true;
            }
            case 5: v.kind = Value::UINT16; return decode_uint(cur, payload, 2, v.uhi, v.u);
            case 6: v.kind = Value::UINT32; return decode_uint(cur, payload, 4, v.uhi, v.u);
            case 7: {  // map (lib.rs:854-878)
                size_t cnt;
                if (!decode_size(cur, payload, cnt)) return false;
                v.kind = Value::MAP;
                for (size_t k = 0; k < cnt; ++k) {
                    Value key;
                    if (!decode_at(cur, key, depth + 1)) return false;
                    std::string ks;
                    if (key.kind == Value::STRING) ks = key.s;
                    else if (key.kind == Value::POINTER) {
                        Value kv;
                        if (!decode((uint32_t)key.u, kv) || kv.kind != Value::STRING) return false;
                        ks = kv.s;
                    } else return false;
                    Value val;
                    if (!decode_at(cur, val, depth + 1)) return false;
                    v.map[ks] = std::move(val);
                }
                return true;
            }
        }
        return false;
    }
    // resolve_pointers (lib.rs:1016-1047)
    bool resolve(Value& v, int depth = 0) {
        if (depth > 64) return false;
        if (v.kind == Value::POINTER) {
            size_t cur = (size_t)v.u;
            Value t;
            if (!decode_at(cur, t)) return false;
            if (!resolve(t, depth + 1)) return false;
            v = std::move(t);
            return true;
        }
        if (v.kind == Value::MAP) { for (auto& kv : v.map) if (!resolve(kv.second, depth + 1)) return false; }
        if (v.kind == Value::ARRAY) { for (auto& e : v.arr) if (!resolve(e, depth + 1)) return false; }
        return true;
    }
    bool decode(uint32_t offset, Value& v) {
        size_t cur = offset;
        if (!decode_at(cur, v)) return false;
        return resolve(v);
    }
};

// ------------------------------------------------------------------ JSON (serde_json compact, sorted keys)
inline void json_escape(const std::string& s, std::string& out) {
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
inline std::string u128_to_dec(uint64_t hi, uint64_t lo) {
    if (hi == 0) return std::to_string(lo);
    unsigned __int128 v = ((unsigned __int128)hi << 64) | lo;
    std::string r;
    while (v) { r.push_back((char)('0' + (int)(v % 10))); v /= 10; }
    std::reverse(r.begin(), r.end());
    return r;
}
inline std::string fmt_double(double d) {  // shortest round-trip; approximates ryu for the values tests use
    if (!(d == d) || d == 1.0 / 0.0 || d == -1.0 / 0.0) return "null";
    char b[64];
    for (int p = 1; p <= 17; ++p) {
        snprintf(b, sizeof(b), "%.*g", p, d);
        if (strtod(b, nullptr) == d) break;
    }
    std::string s(b);
    if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
    return s;
}
// data_value_to_json (crates/matchy/src/bin/cli_utils.rs:177-201)
inline void value_to_json(const Value& v, std::string& out) {
    switch (v.kind) {
        case Value::STRING: json_escape(v.s, out); break;
        case Value::DOUBLE: out += fmt_double(v.d); break;
        case Value::FLOAT: out += fmt_double((double)v.f); break;
        case Value::BYTES: {
            out.push_back('[');
            for (size_t i = 0; i < v.s.size(); ++i) { if (i) out.push_back(','); out += std::to_string((unsigned)(uint8_t)v.s[i]); }
            out.push_back(']');
            break;
        }
        case Value::UINT16: case Value::UINT32: case Value::UINT64: out += std::to_string(v.u); break;
        case Value::UINT128: out.push_back('"'); out += u128_to_dec(v.uhi, v.u); out.push_back('"'); break;
        case Value::INT32: out += std::to_string(v.i); break;
        case Value::BOOL: out += v.u ? "true" : "false"; break;
        case Value::MAP: {
            out.push_back('{');
            bool first = true;
            for (auto& kv : v.map) {
                if (!first) out.push_back(',');
                first = false;
                json_escape(kv.first, out);
                out.push_back(':');
                value_to_json(kv.second, out);
            }
            out.push_back('}');
            break;
        }
        case Value::ARRAY: {
            out.push_back('[');
            for (size_t i = 0; i < v.arr.size(); ++i) { if (i) out.push_back(','); value_to_json(v.arr[i], out); }
            out.push_back(']');
            break;
        }
        case Value::POINTER: out += "\"<pointer>\""; break;
    }
}

// ------------------------------------------------------------------ IP address Display (Rust std)
inline std::string fmt_ipv4(const uint8_t a[4]) {
    char b[20];
    snprintf(b, sizeof(b), "%u.%u.%u.%u", a[0], a[1], a[2], a[3]);
    return b;
}
inline std::string fmt_ipv6(const uint8_t a[16]) {  // <Ipv6Addr as Display>: RFC 5952 + ::ffff:a.b.c.d
    uint16_t seg[8];
    for (int i = 0; i < 8; ++i) seg[i] = (uint16_t)((a[2 * i] << 8) | a[2 * i + 1]);
    if (seg[0] == 0 && seg[1] == 0 && seg[2] == 0 && seg[3] == 0 && seg[4] == 0 && seg[5] == 0xffff)
        return "::ffff:" + fmt_ipv4(a + 12);
    int best_start = 0, best_len = 0, cur_start = 0, cur_len = 0;
    for (int i = 0; i < 8; ++i) {
        if (seg[i] == 0) {
            if (cur_len == 0) cur_start = i;
            ++cur_len;
            if (cur_len > best_len) { best_len = cur_len; best_start = cur_start; }
        } else cur_len = 0;
    }
    auto sub = [&](int lo, int hi) {
        std::string r;
        char b[8];
        for (int i = lo; i < hi; ++i) { if (i > lo) r.push_back(':'); snprintf(b, sizeof(b), "%x", seg[i]); r += b; }
        return r;
    };
    if (best_len > 1) return sub(0, best_start) + "::" + sub(best_start + best_len, 8);
    return sub(0, 8);
}

// ------------------------------------------------------------------ Database
struct QueryResult {
    enum Kind { NONE = 0, NOT_FOUND = 1, IP = 2, PATTERN = 3 } kind = NONE;
    uint8_t prefix_len = 0;
    uint32_t ip_data_offset = 0;
    std::vector<uint32_t> pattern_ids;
    std::vector<int64_t> data_offsets;  // -1 = None
};

static inline uint32_t rd32le(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd64le(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint16_t rd16le(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }

struct Database {
    std::vector<uint8_t> bytes;
    const uint8_t* data = nullptr;
    size_t len = 0;
    std::string error;

    // MmdbHeader (format.rs:20-29)
    uint32_t node_count = 0;
    int record_size = 24;
    int ip_version = 4;
    size_t tree_size = 0;
    bool has_ip = false;

    // literal hash (literal-hash/lib.rs:370-457)
    bool has_literal = false;
    const uint8_t* lh = nullptr; size_t lh_len = 0;
    uint32_t lh_num_shards = 0, lh_strings_offset = 0, lh_strings_size = 0;
    size_t lh_table_start = 0, lh_mappings_start = 0;
    std::vector<uint32_t> lh_shard_offsets;

    // paraglob (paraglob_offset.rs:1713-1760) + PatternDataMappings (database.rs:203-229)
    bool has_glob = false;
    const uint8_t* pg = nullptr; size_t pg_len = 0;
    size_t pdm_offset = 0, pdm_count = 0;
    bool pg_has_aclh = false;
    const uint8_t* aclh = nullptr; size_t aclh_len = 0;
    uint32_t aclh_table_size = 0, aclh_patterns_start = 0;

    int match_mode = 0;
    Value metadata;

    static long find_metadata_marker(const uint8_t* d, size_t n) {  // format.rs:126-150
        static const uint8_t M[14] = {0xAB, 0xCD, 0xEF, 'M', 'a', 'x', 'M', 'i', 'n', 'd', '.', 'c', 'o', 'm'};
        if (n < 14) return -1;
        size_t start = n > 128 * 1024 ? n - 128 * 1024 : 0;
        long last = -1;
        for (size_t i = start; i + 14 <= n; ++i)
            if (memcmp(d + i, M, 14) == 0) last = (long)i;
        return last;
    }
    static bool get_uint(const Value& m, const char* key, uint64_t& out) {  // format.rs:154-171
        auto it = m.map.find(key);
        if (it == m.map.end()) return false;
        const Value& v = it->second;
        if (v.kind != Value::UINT16 && v.kind != Value::UINT32 && v.kind != Value::UINT64) return false;
        out = v.u;
        return true;
    }

    bool open(const uint8_t* d, size_t n) {
        bytes.assign(d, d + n);
        data = bytes.data();
        len = n;
        if (n >= 8 && memcmp(data, "PARAGLOB", 8) == 0) { error = "pattern-only (.pgb) format not supported by oracle"; return false; }
        long marker = find_metadata_marker(data, len);
        if (marker < 0) { error = "Unknown database format (no MMDB or PARAGLOB marker)"; return false; }
        Decoder md{data + marker + 14, len - marker - 14};
        if (!md.decode(0, metadata) || metadata.kind != Value::MAP) { error = "Failed to decode metadata"; return false; }
        uint64_t nc, rs, ipv;
        if (!get_uint(metadata, "node_count", nc) || !get_uint(metadata, "record_size", rs) || !get_uint(metadata, "ip_version", ipv)) {
            error = "Required metadata field missing"; return false;
        }
        if (rs != 24 && rs != 28 && rs != 32) { error = "Invalid record size"; return false; }
        if (ipv != 4 && ipv != 6) { error = "Invalid IP version"; return false; }
        node_count = (uint32_t)nc; record_size = (int)rs; ip_version = (int)ipv;
        tree_size = (size_t)node_count * (record_size * 2 / 8);
        has_ip = true;
        auto mm = metadata.map.find("match_mode");
        if (mm != metadata.map.end() && mm->second.kind == Value::UINT16) match_mode = mm->second.u == 1 ? 1 : 0;

        // find_pattern_section_fast / find_literal_section_fast: metadata offsets must be Uint32 (database.rs:1222,1259)
        uint32_t pat_off = 0, lit_off = 0;
        auto po = metadata.map.find("pattern_section_offset");
        auto lo = metadata.map.find("literal_section_offset");
        if (po == metadata.map.end() || po->second.kind != Value::UINT32 || lo == metadata.map.end() || lo->second.kind != Value::UINT32) {
            error = "legacy databases without section offsets are not supported by the oracle"; return false;
        }
        pat_off = (uint32_t)po->second.u;
        lit_off = (uint32_t)lo->second.u;

        if (pat_off != 0) {  // load_combined_pattern_section (database.rs:1315-1394)
            size_t off = pat_off;
            if (off >= len || off + 8 > len) { error = "Pattern section header truncated"; return false; }
            size_t pg_size = rd32le(data + off + 4);
            size_t pg_start = off + 8, pg_end = pg_start + pg_size;
            if (pg_end > len) { error = "Paraglob section extends beyond file"; return false; }
            pg = data + pg_start; pg_len = pg_size;
            if (pg_len < 112 || memcmp(pg, "PARAGLOB", 8) != 0 || rd32le(pg + 8) != 5) { error = "Invalid paraglob header"; return false; }
            uint32_t map_off = rd32le(pg + 96), map_cnt = rd32le(pg + 100);
            if (map_cnt > 0 && map_off > 0) {
                if (map_off >= pg_len) { error = "AC literal map offset out of bounds"; return false; }
                aclh = pg + map_off; aclh_len = pg_len - map_off;
                if (aclh_len < 24 || memcmp(aclh, "ACLH", 4) != 0 || rd32le(aclh + 4) != 1) { error = "Invalid ACLH header"; return false; }
                aclh_table_size = rd32le(aclh + 12);
                aclh_patterns_start = rd32le(aclh + 16);
                pg_has_aclh = true;
            }
            size_t ms = pg_end;
            if (ms + 4 > len) { error = "Pattern mappings section truncated"; return false; }
            pdm_count = rd32le(data + ms);
            pdm_offset = ms + 4;
            if (pdm_offset + pdm_count * 4 > len) { error = "Pattern mappings section out of bounds"; return false; }
            has_glob = true;
        }
        if (lit_off != 0) {  // LiteralHash::from_buffer (literal-hash/lib.rs:382-457)
            if (lit_off > len) { error = "literal offset out of bounds"; return false; }
            lh = data + lit_off; lh_len = len - lit_off;
            if (lh_len < 32 || memcmp(lh, "LHSH", 4) != 0) { error = "Invalid literal hash magic"; return false; }
            if (rd32le(lh + 4) != 1) { error = "Unsupported literal hash version"; return false; }
            lh_strings_offset = rd32le(lh + 16); lh_strings_size = rd32le(lh + 20); lh_num_shards = rd32le(lh + 24);
            for (uint32_t i = 0; i <= lh_num_shards; ++i) {
                size_t p = 32 + (size_t)i * 4;
                if (p + 4 > lh_len) { error = "Shard offset table truncated"; return false; }
                lh_shard_offsets.push_back(rd32le(lh + p));
            }
            lh_table_start = 32 + ((size_t)lh_num_shards + 1) * 4;
            lh_mappings_start = (size_t)lh_strings_offset + lh_strings_size;
            has_literal = true;
        }
        return true;
    }

    // ---- tree (tree.rs:132-248)
    bool read_record(uint32_t node, int side, uint32_t& rec) const {
        if (node >= node_count) return false;
        if (record_size == 24) {
            size_t o = (size_t)node * 6 + side * 3;
            if (o + 3 > tree_size) return false;
            rec = ((uint32_t)data[o] << 16) | ((uint32_t)data[o + 1] << 8) | data[o + 2];
        } else if (record_size == 28) {
            size_t o = (size_t)node * 7;
            if (o + 7 > tree_size) return false;
            const uint8_t* b = data + o;
            if (side == 0) rec = ((uint32_t)((b[3] >> 4) & 0xF) << 24) | ((uint32_t)b[0] << 16) | ((uint32_t)b[1] << 8) | b[2];
            else rec = ((uint32_t)(b[3] & 0xF) << 24) | ((uint32_t)b[4] << 16) | ((uint32_t)b[5] << 8) | b[6];
        } else {
            size_t o = (size_t)node * 8 + side * 4;
            if (o + 4 > tree_size) return false;
            rec = ((uint32_t)data[o] << 24) | ((uint32_t)data[o + 1] << 16) | ((uint32_t)data[o + 2] << 8) | data[o + 3];
        }
        return true;
    }
    // returns 1 found, 0 not found, -1 error
    int lookup_v4(uint32_t bits, uint32_t& data_offset, uint8_t& prefix) const {  // tree.rs:46-89, 258-277
        uint32_t node = 0;
        uint32_t depth = 0;
        if (ip_version == 6) {
            for (int k = 0; k < 96; ++k) {
                uint32_t rec;
                if (!read_record(node, 0, rec)) return -1;
                if (rec == node_count) break;
                else if (rec < node_count) node = rec;
                else break;
            }
            depth = 96;
        }
        for (int bi = 0; bi < 32; ++bi) {
            int bit = (bits >> (31 - bi)) & 1;
            uint32_t rec;
            if (!read_record(node, bit, rec)) return -1;
            if (rec == node_count) return 0;
            else if (rec < node_count) { node = rec; depth += 1; }
            else {
                uint32_t off = rec - node_count;
                if (off < 16) return -1;
                data_offset = off - 16;
                prefix = (uint8_t)(depth >= 96 ? depth - 96 + 1 : depth + 1);
                return 1;
            }
        }
        return 0;
    }
    int lookup_v6(const uint8_t a[16], uint32_t& data_offset, uint8_t& prefix) const {  // tree.rs:92-125
        uint32_t node = 0;
        uint32_t depth = 0;
        for (int bi = 0; bi < 128; ++bi) {
            int bit = (a[bi / 8] >> (7 - (bi % 8))) & 1;
            uint32_t rec;
            if (!read_record(node, bit, rec)) return -1;
            if (rec == node_count) return 0;
            else if (rec < node_count) { node = rec; depth = bi + 1; }
            else {
                uint32_t off = rec - node_count;
                if (off < 16) return -1;
                data_offset = off - 16;
                prefix = (uint8_t)(depth + 1);
                return 1;
            }
        }
        return 0;
    }
    // lookup_ip_uncached (database.rs:810-832)
    QueryResult lookup_ip(bool v6, const uint8_t* addr) const {
        QueryResult r;
        if (!has_ip) return r;
        uint32_t off = 0; uint8_t pfx = 0;
        int rc;
        if (v6) rc = lookup_v6(addr, off, pfx);
        else rc = lookup_v4(((uint32_t)addr[0] << 24) | ((uint32_t)addr[1] << 16) | ((uint32_t)addr[2] << 8) | addr[3], off, pfx);
        if (rc <= 0) { r.kind = QueryResult::NOT_FOUND; return r; }
        r.kind = QueryResult::IP; r.prefix_len = pfx; r.ip_data_offset = off;
        return r;
    }

    // ---- literal hash (literal-hash/lib.rs:467-575)
    bool lh_lookup(const uint8_t* q0, size_t qn0, uint32_t& pattern_id) const {
        // case-insensitive: the query is lower-cased the way the keys were (lib.rs:469-472: `query.to_lowercase()`)
        std::string lowered;
        const uint8_t* q = q0;
        size_t qn = qn0;
        if (match_mode == 1) {
            lowered = Lowercase::table().to_lowercase(std::string((const char*)q0, qn0));
            q = (const uint8_t*)lowered.data(); qn = lowered.size();
        }
        uint64_t hash = xxh64(q, qn, 0);
        size_t shard = (size_t)(hash % lh_num_shards);
        size_t s0 = lh_shard_offsets[shard], s1 = lh_shard_offsets[shard + 1];
        size_t cap = s1 - s0;
        if (cap == 0) return false;
        size_t mask = cap - 1;
        size_t slot = s0 + ((size_t)hash & mask);
        for (size_t it = 0; it < cap; ++it) {
            size_t eo = lh_table_start + slot * 16;
            if (eo + 16 > lh_len) return false;
            uint64_t eh = rd64le(lh + eo);
            uint32_t so = rd32le(lh + eo + 8), pid = rd32le(lh + eo + 12);
            if (so == 0xFFFFFFFFu) return false;
            if (eh == hash) {
                size_t abs = (size_t)lh_strings_offset + so;
                if (abs + 2 <= lh_len) {
                    size_t sl = rd16le(lh + abs);
                    if (abs + 2 + sl <= lh_len && sl == qn && memcmp(lh + abs + 2, q, qn) == 0) { pattern_id = pid; return true; }
                }
            }
            slot = s0 + ((slot + 1 - s0) & mask);
        }
        return false;
    }
    bool lh_get_data_offset(uint32_t pattern_id, uint32_t& off) const {
        if (lh_mappings_start + 4 > lh_len) return false;
        uint32_t cnt = rd32le(lh + lh_mappings_start);
        size_t base = lh_mappings_start + 4;
        // reference: linear scan (lib.rs:560-572). The builder writes ids densely 0..n-1 in order, so the
        // direct slot is tried first (same answer as the scan whenever ids are unique) and the scan remains
        // as the general path.
        if ((size_t)pattern_id < cnt && base + (size_t)pattern_id * 8 + 8 <= lh_len &&
            rd32le(lh + base + (size_t)pattern_id * 8) == pattern_id) {
            off = rd32le(lh + base + (size_t)pattern_id * 8 + 4);
            return true;
        }
        for (uint32_t i = 0; i < cnt; ++i) {
            size_t o = base + (size_t)i * 8;
            if (o + 8 > lh_len) return false;
            if (rd32le(lh + o) == pattern_id) { off = rd32le(lh + o + 4); return true; }
        }
        return false;
    }

    // ---- paraglob
    // find_ac_transition (paraglob_offset.rs:1271-1353); returns -1 for None
    long ac_transition(const uint8_t* ac, size_t ac_len, size_t node_off, uint8_t ch) const {
        if (node_off + 20 > ac_len) return -1;
        const uint8_t* nd = ac + node_off;
        uint8_t kind = nd[0];
        switch (kind) {
            case 0: return -1;
            case 1: return nd[1] == ch ? (long)rd32le(nd + 12) : -1;
            case 2: {
                size_t eo = rd32le(nd + 12), cnt = nd[2];
                if (eo + cnt * 8 > ac_len) return -1;
                for (size_t i = 0; i < cnt; ++i) {
                    uint8_t ec = ac[eo + i * 8];
                    if (ec == ch) return (long)rd32le(ac + eo + i * 8 + 4);
                    if (ec > ch) return -1;
                }
                return -1;
            }
            case 3: {
                size_t t = (size_t)rd32le(nd + 12) + (size_t)ch * 4;
                if (t + 4 > ac_len) return -1;
                uint32_t target = rd32le(ac + t);
                return target != 0 ? (long)target : -1;
            }
            default: return -1;
        }
    }
    // run_ac_matching_into_static (paraglob_offset.rs:1186-1266); case-insensitive: the text is ASCII-lower-cased (:1198-1206)
    void run_ac(const uint8_t* ac, size_t ac_len, const uint8_t* text, size_t tn, std::set<uint32_t>& out) const {
        if (ac_len == 0 || tn == 0) return;
        size_t cur = 0;
        for (size_t i = 0; i < tn; ++i) {
            uint8_t ch = text[i];
            if (match_mode == 1 && ch >= 'A' && ch <= 'Z') ch = (uint8_t)(ch + 32);
            for (;;) {
                long nx = ac_transition(ac, ac_len, cur, ch);
                if (nx >= 0) { cur = (size_t)nx; break; }
                if (cur == 0) break;
                if (cur + 20 > ac_len) break;
                cur = rd32le(ac + cur + 8);
            }
            if (cur + 20 > ac_len) continue;
            const uint8_t* nd = ac + cur;
            uint8_t pc = nd[3];
            if (pc > 0) {
                size_t po = rd32le(nd + 16);
                if (po + (size_t)pc * 4 <= ac_len)
                    for (size_t k = 0; k < pc; ++k) out.insert(rd32le(ac + po + k * 4));
            }
        }
    }
    // ACLiteralHash::lookup (literal_hash.rs:263-333)
    void aclh_lookup(uint32_t literal_id, std::vector<uint32_t>& out) const {
        uint64_t hash = fxhash_u32(literal_id);
        size_t ts = aclh_table_size;
        if (ts == 0) return;
        size_t slot = (size_t)(hash % ts);
        for (size_t it = 0; it < ts; ++it) {
            size_t eo = 24 + slot * 16;
            if (eo + 16 > aclh_len) return;
            uint32_t lid = rd32le(aclh + eo);
            if (lid == 0xFFFFFFFFu) return;
            if (lid == literal_id) {
                size_t po = (size_t)aclh_patterns_start + rd32le(aclh + eo + 4);
                size_t cnt = rd32le(aclh + eo + 8);
                if (po + cnt * 4 > aclh_len) return;
                for (size_t k = 0; k < cnt; ++k) out.push_back(rd32le(aclh + po + k * 4));
                return;
            }
            slot = (slot + 1) % ts;
        }
    }
    static size_t utf8_len(uint8_t c) { return c < 0x80 ? 1 : c < 0xE0 ? 2 : c < 0xF0 ? 3 : 4; }
    static uint32_t utf8_decode(const uint8_t* s, size_t n, size_t& adv) {
        uint8_t c = s[0];
        adv = utf8_len(c);
        if (adv > n) adv = n;
        if (adv == 1) return c;
        uint32_t cp = adv == 2 ? (c & 0x1F) : adv == 3 ? (c & 0x0F) : (c & 0x07);
        for (size_t k = 1; k < adv; ++k) cp = (cp << 6) | (s[k] & 0x3F);
        return cp;
    }
    static uint32_t ascii_lower_char(uint32_t c) { return (c >= 'A' && c <= 'Z') ? c + 32 : c; }
    // match_segments_impl (paraglob_offset.rs:1402-1639)
    bool match_segments(const uint8_t* text, size_t tn, size_t first_seg, size_t seg_count, size_t pos, size_t seg, size_t& steps) const {
        if (steps == 0) return false;
        --steps;
        if (seg >= seg_count) return pos >= tn;
        size_t so = first_seg + seg * 12;
        if (so + 12 > pg_len) return false;
        const uint8_t* sh = pg + so;
        uint8_t st = sh[0], fl = sh[1];
        size_t dlen = rd32le(sh + 4), doff = rd32le(sh + 8);
        switch (st) {
            case 0: {
                if (doff + dlen > pg_len) return false;
                if (!valid_utf8_local(pg + doff, dlen)) return false;  // Err(..) is treated as no match by callers
                if (match_mode == 1) {
                    // :1456-1478: literal and text are walked char by char with eq_ignore_ascii_case; the text advances by the
                    // bytes of its own characters
                    size_t lp = 0, tp = pos;
                    bool ok = true;
                    while (tp < tn) {
                        if (lp >= dlen) break;
                        size_t la, ta;
                        uint32_t lc = utf8_decode(pg + doff + lp, dlen - lp, la), tc = utf8_decode(text + tp, tn - tp, ta);
                        if (ascii_lower_char(lc) != ascii_lower_char(tc)) { ok = false; break; }
                        lp += la; tp += ta;
                    }
                    if (ok && lp < dlen) ok = false;
                    if (ok) return match_segments(text, tn, first_seg, seg_count, tp, seg + 1, steps);
                    return false;
                }
                if (tn - pos >= dlen && memcmp(text + pos, pg + doff, dlen) == 0)
                    return match_segments(text, tn, first_seg, seg_count, pos + dlen, seg + 1, steps);
                return false;
            }
            case 1: {
                if (seg + 1 >= seg_count) return true;
                size_t p = pos;
                for (;;) {
                    if (match_segments(text, tn, first_seg, seg_count, p, seg + 1, steps)) return true;
                    if (p >= tn) break;
                    p += utf8_len(text[p]);
                }
                return false;
            }
            case 2: {
                if (pos >= tn) return false;
                return match_segments(text, tn, first_seg, seg_count, pos + utf8_len(text[pos]), seg + 1, steps);
            }
            case 3: {
                if (pos >= tn) return false;
                size_t adv;
                uint32_t ch = utf8_decode(text + pos, tn - pos, adv);
                if (match_mode == 1) ch = ascii_lower_char(ch);   // :1552-1555
                size_t items = dlen / 12;
                if (doff + dlen > pg_len) return false;
                bool negated = (fl & 1) != 0, in_class = false;
                for (size_t k = 0; k < items; ++k) {
                    const uint8_t* it = pg + doff + k * 12;
                    uint32_t c1 = rd32le(it + 4), c2 = rd32le(it + 8);
                    auto is_char = [](uint32_t c) { return c < 0xD800 || (c > 0xDFFF && c <= 0x10FFFF); };
                    if (match_mode == 1) { if (is_char(c1)) c1 = ascii_lower_char(c1); if (is_char(c2)) c2 = ascii_lower_char(c2); }   // :1584-1607
                    bool m = false;
                    if (it[0] == 0) m = is_char(c1) && ch == c1;
                    else if (it[0] == 1) m = is_char(c1) && is_char(c2) && ch >= c1 && ch <= c2;
                    if (m) { in_class = true; break; }
                }
                bool ok = negated ? !in_class : in_class;
                if (!ok) return false;
                return match_segments(text, tn, first_seg, seg_count, pos + adv, seg + 1, steps);
            }
            default: return false;
        }
    }
    static bool valid_utf8_local(const uint8_t* s, size_t n);
    bool match_glob(uint32_t pattern_id, const uint8_t* text, size_t tn) const {  // :1364-1398
        size_t gso = rd32le(pg + 104);
        size_t io = gso + (size_t)pattern_id * 8;
        if (io + 8 > pg_len) return false;
        size_t first = rd32le(pg + io), cnt = rd16le(pg + io + 4);
        size_t steps = 100000;
        return match_segments(text, tn, first, cnt, 0, 0, steps);
    }
    // Paraglob::find_all (paraglob_offset.rs:1028-1182)
    std::vector<uint32_t> find_all(const uint8_t* text, size_t tn) const {
        std::vector<uint32_t> result;
        if (pg_len < 112) return result;
        size_t ac_start = rd32le(pg + 20), ac_size = rd32le(pg + 24);
        std::set<uint32_t> candidates;
        if (ac_size > 0 && ac_start + ac_size <= pg_len) {
            std::set<uint32_t> lits;
            run_ac(pg + ac_start, ac_size, text, tn, lits);
            if (!lits.empty() && pg_has_aclh)
                for (uint32_t lid : lits) { std::vector<uint32_t> ids; aclh_lookup(lid, ids); candidates.insert(ids.begin(), ids.end()); }
        }
        size_t unaligned = (size_t)rd32le(pg + 40) + rd32le(pg + 44);
        size_t wild_off = unaligned + (8 - unaligned % 8) % 8;
        size_t wild_cnt = rd32le(pg + 60);
        size_t patterns_off = rd32le(pg + 36);
        for (size_t i = 0; i < wild_cnt; ++i) {
            size_t wo = wild_off + i * 8;
            if (wo + 8 > pg_len) continue;
            uint32_t pid = rd32le(pg + wo);
            if (patterns_off + (size_t)pid * 16 + 16 > pg_len) continue;
            if (match_glob(pid, text, tn)) result.push_back(pid);
        }
        for (uint32_t pid : candidates) {
            size_t eo = patterns_off + (size_t)pid * 16;
            if (eo + 16 > pg_len) continue;
            uint32_t entry_id = rd32le(pg + eo);
            uint8_t ptype = pg[eo + 4];
            if (ptype == 0) result.push_back(entry_id);
            else if (match_glob(entry_id, text, tn)) result.push_back(entry_id);
        }
        std::sort(result.begin(), result.end());
        result.erase(std::unique(result.begin(), result.end()), result.end());
        return result;
    }

    // lookup_string_uncached (database.rs:911-981)
    QueryResult lookup_string(const uint8_t* s, size_t n) const {
        QueryResult r;
        if (has_literal) {
            uint32_t pid;
            if (lh_lookup(s, n, pid)) {
                uint32_t off;
                if (lh_get_data_offset(pid, off)) { r.pattern_ids.push_back(pid); r.data_offsets.push_back(off); }
            }
        }
        if (has_glob) {
            for (uint32_t pid : find_all(s, n)) {
                int64_t off = -1;
                if ((size_t)pid < pdm_count && pdm_offset + (size_t)pid * 4 + 4 <= len) off = rd32le(data + pdm_offset + (size_t)pid * 4);
                r.pattern_ids.push_back(pid);
                r.data_offsets.push_back(off);
            }
        }
        if (r.pattern_ids.empty()) r.kind = (has_literal || has_glob) ? QueryResult::NOT_FOUND : QueryResult::NONE;
        else r.kind = QueryResult::PATTERN;
        return r;
    }

    // decode_ip_data (database.rs:1005-1020) → JSON
    bool data_json(uint32_t offset, std::string& out) const {
        size_t ds = tree_size + 16;
        if (ds > len) return false;
        Decoder dec{data + ds, len - ds};
        Value v;
        if (!dec.decode(offset, v)) return false;
        value_to_json(v, out);
        return true;
    }
};

inline bool Database::valid_utf8_local(const uint8_t* s, size_t n) {
    size_t i = 0;
    while (i < n) {
        uint8_t c = s[i];
        if (c < 0x80) { ++i; continue; }
        size_t l = c >= 0xC2 && c <= 0xDF ? 2 : c >= 0xE0 && c <= 0xEF ? 3 : c >= 0xF0 && c <= 0xF4 ? 4 : 0;
        if (l == 0 || i + l > n) return false;
        for (size_t k = 1; k < l; ++k) if ((s[i + k] & 0xC0) != 0x80) return false;
        if (c == 0xE0 && s[i + 1] < 0xA0) return false;
        if (c == 0xED && s[i + 1] > 0x9F) return false;
        if (c == 0xF0 && s[i + 1] < 0x90) return false;
        if (c == 0xF4 && s[i + 1] > 0x8F) return false;
        i += l;
    }
    return true;
}

}  // namespace orc
