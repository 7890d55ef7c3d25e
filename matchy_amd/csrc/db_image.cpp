#include "db_image.h"

#include <cstring>

namespace mxy {

namespace {
inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

long find_metadata_marker(const uint8_t* d, size_t n) {  // mmdb/format.rs:126-150: last occurrence in the final 128 KiB
    static const uint8_t M[14] = {0xAB, 0xCD, 0xEF, 'M', 'a', 'x', 'M', 'i', 'n', 'd', '.', 'c', 'o', 'm'};
    if (n < 14) return -1;
    size_t start = n > 128 * 1024 ? n - 128 * 1024 : 0;
    long last = -1;
    for (size_t i = start; i + 14 <= n; ++i)
        if (memcmp(d + i, M, 14) == 0) last = (long)i;
    return last;
}
bool meta_uint(const DataValue& m, const char* key, uint64_t& out) {
    auto it = m.map.find(key);
    if (it == m.map.end()) return false;
    const DataValue& v = it->second;
    if (v.type != DataValue::UINT16 && v.type != DataValue::UINT32 && v.type != DataValue::UINT64) return false;
    out = v.u;
    return true;
}
}  // namespace

bool DbImage::open(std::vector<uint8_t>&& data, std::string& err) {
    bytes = std::move(data);
    const uint8_t* d = bytes.data();
    size_t n = bytes.size();
    if (n >= 8 && memcmp(d, "PARAGLOB", 8) == 0) { err = "pattern-only (.pgb) databases are not supported"; return false; }
    long marker = find_metadata_marker(d, n);
    if (marker < 0) { err = "Unknown database format (no MMDB or PARAGLOB marker)"; return false; }
    if (!decode_value(d + marker + 14, n - marker - 14, 0, metadata) || metadata.type != DataValue::MAP) { err = "Failed to decode metadata"; return false; }
    uint64_t nc, rs, ipv;
    if (!meta_uint(metadata, "node_count", nc) || !meta_uint(metadata, "record_size", rs) || !meta_uint(metadata, "ip_version", ipv)) {
        err = "Required metadata field missing"; return false;
    }
    if (rs != 24 && rs != 28 && rs != 32) { err = "Invalid record size"; return false; }
    if (ipv != 4 && ipv != 6) { err = "Invalid IP version"; return false; }
    node_count = (uint32_t)nc; record_size = (int)rs; ip_version = (int)ipv;
    tree_size = (size_t)node_count * (size_t)(record_size * 2 / 8);
    if (node_count == 0 || tree_size + 16 > n) { err = "Search tree extends beyond file"; return false; }
    has_ip = true;
    auto mm = metadata.map.find("match_mode");
    if (mm != metadata.map.end() && mm->second.type == DataValue::UINT16) match_mode = mm->second.u == 1 ? 1 : 0;

    auto po = metadata.map.find("pattern_section_offset");
    auto lo = metadata.map.find("literal_section_offset");
    if (po == metadata.map.end() || po->second.type != DataValue::UINT32 || lo == metadata.map.end() || lo->second.type != DataValue::UINT32) {
        err = "legacy databases without section-offset metadata are not supported"; return false;
    }
    size_t pat_off = (size_t)po->second.u, lit_off = (size_t)lo->second.u;

    if (pat_off) {
        if (pat_off + 8 > n) { err = "Pattern section header truncated"; return false; }
        size_t pg_size = rd32(d + pat_off + 4);
        pg_off = pat_off + 8;
        pg_len = pg_size;
        if (pg_off + pg_len > n || pg_len < 112) { err = "Paraglob section extends beyond file"; return false; }
        const uint8_t* pg = d + pg_off;
        if (memcmp(pg, "PARAGLOB", 8) != 0 || rd32(pg + 8) != 5) { err = "Unsupported paraglob header/version"; return false; }
        // the section's own match_mode field (offset 12) is not consulted: like the reference (database.rs:1297-1304,
        // 1323-1361) the mode comes from the metadata
        pattern_count = rd32(pg + 32);
        uint32_t ac_start = rd32(pg + 20), ac_size = rd32(pg + 24), patterns_off = rd32(pg + 36), gso = rd32(pg + 104);
        if ((size_t)ac_start + ac_size > pg_len || (ac_start & 3)) { err = "AC section out of bounds"; return false; }
        if ((size_t)patterns_off + (size_t)pattern_count * 16 > pg_len) { err = "Pattern entries out of bounds"; return false; }
        if ((size_t)gso + (size_t)pattern_count * 8 > pg_len) { err = "Glob segment index out of bounds"; return false; }
        size_t ms = pg_off + pg_len;
        if (ms + 4 > n) { err = "Pattern mappings section truncated"; return false; }
        pdm_count = rd32(d + ms);
        pdm_off = ms + 4;
        if (pdm_off + pdm_count * 4 > n) { err = "Pattern mappings section out of bounds"; return false; }
        has_glob = true;
    }
    if (lit_off) {
        if (lit_off + 32 > n) { err = "Literal section truncated"; return false; }
        lh_off = lit_off;
        lh_len = n - lit_off;
        const uint8_t* lh = d + lh_off;
        if (memcmp(lh, "LHSH", 4) != 0 || rd32(lh + 4) != 1) { err = "Invalid literal hash header"; return false; }
        lh_table_size = rd32(lh + 12); lh_strings_offset = rd32(lh + 16); lh_strings_size = rd32(lh + 20); lh_num_shards = rd32(lh + 24);
        lh_table_start = 32 + ((size_t)lh_num_shards + 1) * 4;
        if (lh_table_start + (size_t)lh_table_size * 16 > lh_len || (size_t)lh_strings_offset + lh_strings_size + 4 > lh_len) {
            err = "Literal hash table out of bounds"; return false;
        }
        size_t mstart = (size_t)lh_strings_offset + lh_strings_size;
        uint32_t cnt = rd32(lh + mstart);
        if (mstart + 4 + (size_t)cnt * 8 > lh_len) { err = "Literal mappings out of bounds"; return false; }
        bool dense = true;
        for (uint32_t i = 0; i < cnt; ++i) if (rd32(lh + mstart + 4 + (size_t)i * 8) != i) { dense = false; break; }
        if (dense) {
            lit_data_offsets.resize(cnt);
            for (uint32_t i = 0; i < cnt; ++i) lit_data_offsets[i] = rd32(lh + mstart + 8 + (size_t)i * 8);
        } else {
            // not in id order: still ids 0..cnt-1 (mmdb_builder.rs:562-564 numbers the literals by enumerate()); a larger id would
            // size the per-pattern offset table of the upload from a field of the file
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint32_t pid = rd32(lh + mstart + 4 + (size_t)i * 8);
                if (pid >= cnt) { err = "Literal mappings name an implausible pattern id"; return false; }
                lit_data_map.emplace(pid, rd32(lh + mstart + 8 + (size_t)i * 8));
            }
        }
        has_literal = true;
    }
    return check_structure(err);
}

bool DbImage::check_structure(std::string& err) const {
    const uint8_t* d = bytes.data();
    const size_t n = bytes.size();
    // ---- IP tree: a record above node_count points into the data section (tree.rs:93-125)
    {
        const size_t dlen = n - (tree_size + 16);
        const uint64_t limit = (uint64_t)node_count + 16 + dlen;
        const size_t stride = (size_t)record_size * 2 / 8;
        for (uint32_t i = 0; i < node_count; ++i) {
            const uint8_t* b = d + (size_t)i * stride;
            uint32_t l, r;
            if (record_size == 24) { l = ((uint32_t)b[0] << 16) | ((uint32_t)b[1] << 8) | b[2]; r = ((uint32_t)b[3] << 16) | ((uint32_t)b[4] << 8) | b[5]; }
            else if (record_size == 28) { l = ((uint32_t)(b[3] >> 4) << 24) | ((uint32_t)b[0] << 16) | ((uint32_t)b[1] << 8) | b[2]; r = ((uint32_t)(b[3] & 0xF) << 24) | ((uint32_t)b[4] << 16) | ((uint32_t)b[5] << 8) | b[6]; }
            else { l = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]; r = ((uint32_t)b[4] << 24) | ((uint32_t)b[5] << 16) | ((uint32_t)b[6] << 8) | b[7]; }
            if ((uint64_t)l >= limit || (uint64_t)r >= limit) { err = "IP tree record points outside the data section"; return false; }
            max_ip_record = std::max(max_ip_record, std::max(l, r));
        }
    }
    // ---- literal hash: the length-prefixed string of every occupied slot lies inside the pool (lh:467-543)
    if (has_literal) {
        const uint8_t* lh = d + lh_off;
        for (uint32_t i = 0; i < lh_table_size; ++i) {
            const uint32_t so = rd32(lh + lh_table_start + (size_t)i * 16 + 8);
            if (so == 0xFFFFFFFFu) continue;
            if ((uint64_t)so + 2 > lh_strings_size) { err = "Literal hash entry points outside the string pool"; return false; }
            const uint32_t sl = (uint32_t)lh[lh_strings_offset + so] | ((uint32_t)lh[lh_strings_offset + so + 1] << 8);
            if ((uint64_t)so + 2 + sl > lh_strings_size) { err = "Literal hash string extends beyond the string pool"; return false; }
        }
    }
    if (!has_glob) return true;
    const uint8_t* pg = d + pg_off;
    // ---- pure-wildcard list, pattern entries and their strings, glob segments (offset_format.rs:73-476)
    {
        const uint64_t unaligned = (uint64_t)rd32(pg + 40) + rd32(pg + 44);
        const uint64_t wild_off = unaligned + (8 - unaligned % 8) % 8, wild_count = rd32(pg + 60);
        if (wild_count && wild_off + wild_count * 8 > pg_len) { err = "Wildcard list out of bounds"; return false; }
        for (uint64_t i = 0; i < wild_count; ++i)
            if (rd32(pg + wild_off + i * 8) >= pattern_count) { err = "Wildcard entry names an unknown pattern"; return false; }
        const uint32_t patterns_off = rd32(pg + 36), gso = rd32(pg + 104);
        for (uint32_t pid = 0; pid < pattern_count; ++pid) {
            const uint8_t* e = pg + (size_t)patterns_off + (size_t)pid * 16;
            if ((uint64_t)rd32(e + 8) + rd32(e + 12) > pg_len) { err = "Pattern string out of bounds"; return false; }
            const uint64_t first = rd32(pg + (size_t)gso + (size_t)pid * 8), count = rd32(pg + (size_t)gso + (size_t)pid * 8 + 4) & 0xFFFFu;
            if (count == 0) continue;
            if (first + count * 12 > pg_len) { err = "Glob segment list out of bounds"; return false; }
            for (uint64_t k = 0; k < count; ++k) {
                const uint8_t* sh = pg + first + k * 12;
                const uint32_t st = sh[0];
                if ((st == 0 || st == 3) && (uint64_t)rd32(sh + 8) + rd32(sh + 4) > pg_len) { err = "Glob segment data out of bounds"; return false; }
            }
        }
    }
    // ---- ACLH literal -> pattern table (literal_hash.rs:48-77): ids name patterns
    {
        const uint32_t map_off = rd32(pg + 96), map_cnt = rd32(pg + 100);
        if (map_cnt && map_off) {
            if ((uint64_t)map_off + 24 > pg_len || memcmp(pg + map_off, "ACLH", 4) != 0) { err = "AC literal map header invalid"; return false; }
            const uint8_t* a = pg + map_off;
            const uint64_t alen = pg_len - map_off, table_size = rd32(a + 12), pstart = rd32(a + 16);
            if (24 + table_size * 16 > alen) { err = "AC literal map table out of bounds"; return false; }
            for (uint64_t s2 = 0; s2 < table_size; ++s2) {
                const uint8_t* e = a + 24 + s2 * 16;
                if (rd32(e) == 0xFFFFFFFFu) continue;
                if (rd32(e + 8) == 0) continue;   // an entry without patterns maps nothing (db_builder.cpp writes such fillers)
                // literal ids are dense (paraglob_offset.rs:587: enumerate() over the AC literals), so the table has at least as
                // many slots as the largest id + 1; the host tables indexed by literal id are sized from the largest id
                if (rd32(e) >= table_size) { err = "AC literal map names an implausible literal id"; return false; }
                const uint64_t po = rd32(e + 4), pc = rd32(e + 8);
                if (pstart + po + pc * 4 > alen) { err = "AC literal map pattern list out of bounds"; return false; }
                for (uint64_t k = 0; k < pc; ++k)
                    if (rd32(a + pstart + po + k * 4) >= pattern_count) { err = "AC literal map names an unknown pattern"; return false; }
            }
        }
    }
    // ---- Aho-Corasick nodes reachable from the root (matchy-ac/src/lib.rs:118-124, 201-516). Breadth-first over the goto
    // edges, so every node gets its depth (= length of the shortest text that reaches it = length of its string). Failure
    // links must lead to a goto-reachable node of SMALLER depth (a proper suffix): that is what lets the device walk follow
    // failure chains without a bound of its own — a cycle of failure links would never end.
    {
        const uint64_t ac_start = rd32(pg + 20), ac_size = rd32(pg + 24);
        if (ac_size >= 20) {
            const uint8_t* ac = pg + ac_start;
            constexpr uint32_t UNSEEN = 0xFFFFFFFFu;
            std::vector<uint32_t> depth(ac_size / 4 + 1, UNSEEN);
            std::vector<uint32_t> queue{0};
            depth[0] = 0;
            uint32_t cur_depth = 0;
            auto visit = [&](uint64_t target) -> bool {
                if ((target & 3) || target + 20 > ac_size) return false;
                if (depth[target / 4] == UNSEEN) { depth[target / 4] = cur_depth + 1; queue.push_back((uint32_t)target); }
                return true;
            };
            for (size_t qi = 0; qi < queue.size(); ++qi) {
                const uint64_t off = queue[qi];
                cur_depth = depth[off / 4];
                const uint32_t w0 = rd32(ac + off), kind = w0 & 0xFF, pc = ac[off + 3];
                const uint64_t eo = rd32(ac + off + 12), po = rd32(ac + off + 16);
                bool ok = true;
                if (kind == 1) ok = visit(eo);
                else if (kind == 2) {
                    const uint64_t cnt = (w0 >> 16) & 0xFF;
                    ok = eo + cnt * 8 <= ac_size;
                    for (uint64_t i = 0; ok && i < cnt; ++i) ok = visit(rd32(ac + eo + i * 8 + 4));
                } else if (kind == 3) {
                    ok = eo + 1024 <= ac_size;
                    for (uint64_t c = 0; ok && c < 256; ++c) { const uint32_t t = rd32(ac + eo + c * 4); if (t) ok = visit(t); }
                } else if (kind != 0) ok = false;
                if (ok && pc) ok = po + (uint64_t)pc * 4 <= ac_size;   // output literal ids
                if (!ok) { err = "Aho-Corasick automaton malformed (node at offset " + std::to_string(off) + ")"; return false; }
            }
            for (const uint32_t off : queue) {
                const uint64_t fo = rd32(ac + off + 8);   // failure link (0 = root)
                if (fo == 0) continue;
                if ((fo & 3) || fo + 20 > ac_size || depth[fo / 4] == UNSEEN || depth[fo / 4] >= depth[off / 4]) {
                    err = "Aho-Corasick automaton malformed (failure link of the node at offset " + std::to_string(off) + ")"; return false;
                }
            }
        }
    }
    return true;
}

bool DbImage::lit_data_offset(uint32_t pid, uint32_t& off) const {
    if (!lit_data_offsets.empty()) { if (pid >= lit_data_offsets.size()) return false; off = lit_data_offsets[pid]; return true; }
    auto it = lit_data_map.find(pid);
    if (it == lit_data_map.end()) return false;
    off = it->second;
    return true;
}
bool DbImage::glob_data_offset(uint32_t pid, uint32_t& off) const {
    if (pid >= pdm_count) return false;
    off = rd32(bytes.data() + pdm_off + (size_t)pid * 4);
    return true;
}
std::string DbImage::format_name() const {  // Database::format (database.rs:1072-1078)
    return (has_glob || has_literal) ? "Combined IP+Pattern database" : "IP database";
}
std::string DbImage::pattern_string(uint32_t pid) const {
    if (!has_glob || pid >= pattern_count) return {};
    const uint8_t* pg = bytes.data() + pg_off;
    size_t eo = (size_t)rd32(pg + 36) + (size_t)pid * 16;
    uint32_t so = rd32(pg + eo + 8), sl = rd32(pg + eo + 12);
    if ((size_t)so + sl > pg_len) return {};
    return std::string((const char*)pg + so, sl);
}

void DbImage::build_ip_nodes(std::vector<uint2>& out, uint32_t& v4_start) const {
    out.resize(node_count);
    const uint8_t* t = bytes.data();
    for (uint32_t i = 0; i < node_count; ++i) {
        uint32_t l, r;
        if (record_size == 24) {
            const uint8_t* b = t + (size_t)i * 6;
            l = ((uint32_t)b[0] << 16) | ((uint32_t)b[1] << 8) | b[2];
            r = ((uint32_t)b[3] << 16) | ((uint32_t)b[4] << 8) | b[5];
        } else if (record_size == 28) {
            const uint8_t* b = t + (size_t)i * 7;
            l = ((uint32_t)(b[3] >> 4) << 24) | ((uint32_t)b[0] << 16) | ((uint32_t)b[1] << 8) | b[2];
            r = ((uint32_t)(b[3] & 0xF) << 24) | ((uint32_t)b[4] << 16) | ((uint32_t)b[5] << 8) | b[6];
        } else {
            const uint8_t* b = t + (size_t)i * 8;
            l = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
            r = ((uint32_t)b[4] << 24) | ((uint32_t)b[5] << 16) | ((uint32_t)b[6] << 8) | b[7];
        }
        out[i] = make_uint2(l, r);
    }
    // find_ipv4_start_node (tree.rs:258-277): follow 96 left links, stop where the walk leaves the node range
    v4_start = 0;
    if (ip_version == 6) {
        uint32_t node = 0;
        for (int k = 0; k < 96; ++k) {
            uint32_t rec = out[node].x;
            if (rec < node_count) node = rec;
            else break;
        }
        v4_start = node;
    }
}

void DbImage::build_lit_table(std::vector<LitSlot>& slots, uint32_t& mask) const {
    const uint8_t* lh = bytes.data() + lh_off;
    size_t live = 0;
    for (uint32_t i = 0; i < lh_table_size; ++i) live += rd32(lh + lh_table_start + (size_t)i * 16 + 8) != 0xFFFFFFFFu;
    size_t cap = 16;
    while (cap < live * 2) cap <<= 1;
    mask = (uint32_t)(cap - 1);
    slots.assign(cap, LitSlot{0, 0xFFFFFFFFu, 0});
    for (uint32_t i = 0; i < lh_table_size; ++i) {
        const uint8_t* e = lh + lh_table_start + (size_t)i * 16;
        uint32_t so = rd32(e + 8);
        if (so == 0xFFFFFFFFu) continue;
        if ((size_t)so + 2 > lh_strings_size) continue;  // unreachable string: the reference's read_string would fail too
        uint64_t h = rd64(e);
        uint32_t s = (uint32_t)(h ^ (h >> 32)) & mask;
        while (slots[s].str_off != 0xFFFFFFFFu) s = (s + 1) & mask;
        slots[s] = LitSlot{h, so, rd32(e + 12)};
    }
}

void DbImage::build_lit2pat(std::vector<uint32_t>& off, std::vector<uint32_t>& ids) const {
    off.assign(1, 0);
    ids.clear();
    if (!has_glob) return;
    const uint8_t* pg = bytes.data() + pg_off;
    uint32_t map_off = rd32(pg + 96), map_cnt = rd32(pg + 100);
    if (map_cnt == 0 || map_off == 0 || (size_t)map_off + 24 > pg_len) return;
    const uint8_t* a = pg + map_off;
    size_t alen = pg_len - map_off;
    if (memcmp(a, "ACLH", 4) != 0) return;
    uint32_t table_size = rd32(a + 12), pstart = rd32(a + 16);
    if (24 + (size_t)table_size * 16 > alen) return;
    // enumerate every slot (no dependence on the slot hash function), then densify by literal id
    std::vector<std::pair<uint32_t, std::pair<uint32_t, uint32_t>>> ents;  // lit -> (offset, count)
    uint32_t max_lit = 0;
    for (uint32_t s = 0; s < table_size; ++s) {
        const uint8_t* e = a + 24 + (size_t)s * 16;
        uint32_t lit = rd32(e);
        if (lit == 0xFFFFFFFFu) continue;
        uint32_t po = rd32(e + 4), pc = rd32(e + 8);
        if (pc == 0) continue;   // filler / literal without patterns
        if ((size_t)pstart + po + (size_t)pc * 4 > alen) continue;
        ents.push_back({lit, {po, pc}});
        if (lit + 1 > max_lit) max_lit = lit + 1;
    }
    std::vector<std::pair<uint32_t, uint32_t>> by_lit(max_lit, {0, 0});
    std::vector<bool> seen(max_lit, false);
    for (auto& en : ents) if (!seen[en.first]) { by_lit[en.first] = en.second; seen[en.first] = true; }
    off.assign(max_lit + 1, 0);
    for (uint32_t l = 0; l < max_lit; ++l) {
        off[l] = (uint32_t)ids.size();
        for (uint32_t k = 0; k < by_lit[l].second; ++k) ids.push_back(rd32(a + pstart + by_lit[l].first + (size_t)k * 4));
    }
    off[max_lit] = (uint32_t)ids.size();
}

bool DbImage::build_ac_dfa(std::vector<uint32_t>& next, std::vector<uint8_t>& cls, uint32_t& k, std::vector<uint32_t>& node_off,
                           size_t max_bytes, bool* alnum_literal) const {
    next.clear(); cls.assign(256, 0); node_off.clear(); k = 0;
    if (alnum_literal) *alnum_literal = true;   // until the whole trie has been seen
    if (!has_glob) return false;
    const uint8_t* pg = bytes.data() + pg_off;
    const uint32_t ac_start = rd32(pg + 20), ac_size = rd32(pg + 24);
    if (ac_size < 20 || (size_t)ac_start + ac_size > pg_len) return false;
    const uint8_t* ac = pg + ac_start;
    // goto edges of one node, as find_ac_transition reads them (paraglob_offset.rs:1271-1353)
    struct Edge { uint8_t ch; uint32_t target; };
    auto edges_of = [&](uint32_t off, std::vector<Edge>& out) -> bool {
        out.clear();
        if ((size_t)off + 20 > ac_size) return false;
        const uint32_t w0 = rd32(ac + off), kind = w0 & 0xFF, eo = rd32(ac + off + 12);
        if (kind == 1) out.push_back({(uint8_t)((w0 >> 8) & 0xFF), eo});
        else if (kind == 2) {
            const uint32_t cnt = (w0 >> 16) & 0xFF;
            if ((size_t)eo + (size_t)cnt * 8 > ac_size) return false;
            int prev = -1;
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint8_t c = ac[eo + i * 8];
                if ((int)c <= prev) return false;  // the reader's early exit relies on sorted edges; refuse anything else
                prev = c;
                out.push_back({c, rd32(ac + eo + i * 8 + 4)});
            }
        } else if (kind == 3) {
            if ((size_t)eo + 1024 > ac_size) return false;
            for (uint32_t c = 0; c < 256; ++c) { const uint32_t t = rd32(ac + eo + c * 4); if (t) out.push_back({(uint8_t)c, t}); }
        }
        return true;
    };
    // breadth-first numbering of the reachable nodes
    std::vector<uint32_t> state_of(ac_size / 4 + 1, 0xFFFFFFFFu), depth;
    std::vector<uint8_t> alnum_path(1, 1);   // the bytes on the way from the root to this state are ASCII letters and digits
    std::vector<Edge> flat, tmp;        // goto edges of all states, CSR by state
    std::vector<size_t> edge_begin;
    bool used[256] = {false};
    node_off.push_back(0);
    depth.push_back(0);
    state_of[0] = 0;
    for (size_t s = 0; s < node_off.size(); ++s) {
        if (!edges_of(node_off[s], tmp)) return false;
        for (const Edge& e : tmp) {
            if ((e.target & 3) || (size_t)e.target + 20 > ac_size) return false;
            used[e.ch] = true;
            if (state_of[e.target / 4] == 0xFFFFFFFFu) {
                state_of[e.target / 4] = (uint32_t)node_off.size();
                node_off.push_back(e.target);
                depth.push_back(depth[s] + 1);
                const uint8_t c = e.ch;
                alnum_path.push_back(alnum_path[s] && ((c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z')));
            }
        }
        edge_begin.push_back(flat.size());
        flat.insert(flat.end(), tmp.begin(), tmp.end());
    }
    edge_begin.push_back(flat.size());
    // byte classes: every byte that labels an edge is its own class; all the others share class 0 (they lead back to
    // the root from every state). If all 256 values label edges there is no class 0 and classes are the byte values.
    uint32_t n_used = 0;
    for (int c = 0; c < 256; ++c) n_used += used[c];
    if (n_used == 256) {
        for (int c = 0; c < 256; ++c) cls[c] = (uint8_t)c;
        k = 256;
    } else {
        k = 1;
        for (int c = 0; c < 256; ++c) if (used[c]) cls[c] = (uint8_t)k++;
    }
    const size_t n = node_off.size();
    if (n >= 0x7FFFFFFFu || n * (size_t)k * 4 > max_bytes) { next.clear(); return false; }
    next.assign(n * (size_t)k, 0);
    std::vector<uint8_t> has_out(n);
    for (size_t s = 0; s < n; ++s) has_out[s] = ac[node_off[s] + 3] != 0;
    if (alnum_literal) {
        // The state after a text of letters and digits is a node whose path consists of them (the longest suffix of the text that is a prefix of a
        // literal), and what such a node puts out are suffixes of its path: if none of these nodes has output, no such text contains a literal.
        bool any = false;
        for (size_t s = 1; s < n && !any; ++s) any = alnum_path[s] && has_out[s];
        *alnum_literal = any;
    }
    // states are in breadth-first order, so a valid failure link (strictly shallower) is always resolved already
    for (size_t s = 0; s < n; ++s) {
        uint32_t* row = &next[s * k];
        if (s == 0) {
            for (uint32_t c = 0; c < k; ++c) row[c] = 0;
        } else {
            const uint32_t fo = rd32(ac + node_off[s] + 8);
            if ((fo & 3) || fo / 4 >= state_of.size()) return false;
            const uint32_t fs = state_of[fo / 4];
            if (fs == 0xFFFFFFFFu || depth[fs] >= depth[s]) { next.clear(); return false; }
            const uint32_t* frow = &next[(size_t)fs * k];
            for (uint32_t c = 0; c < k; ++c) row[c] = frow[c] & 0x7FFFFFFFu;
        }
        for (size_t q = edge_begin[s]; q < edge_begin[s + 1]; ++q) row[cls[flat[q].ch]] = state_of[flat[q].target / 4];
    }
    for (size_t i = 0; i < next.size(); ++i) if (has_out[next[i]]) next[i] |= 0x80000000u;
    return true;
}

}  // namespace mxy
