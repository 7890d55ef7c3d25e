// oracle/matchy_oracle.cpp — TEST INFRASTRUCTURE ONLY.
//
// CPU oracle for the `matchy match` hot path: a restatement of the reference's Rust algorithm
// (matchylabs/matchy @ 2025-12-12) used ONLY by tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg, as the checker. The shipped library (matchy_amd/csrc) does not include, link or
// call anything in this directory.
//
// Parity status: PINNED BY FIXTURES ONLY. The reference is Rust and cannot be built in this image
// (no cargo/rustc); the restatement is checked against the known-answer vectors transcribed from the
// reference's own tests (tests/golden/extractor_kat.json etc.). See DESIGN.md §Oracle.
//
// Restated here:
//   Worker::process_bytes            crates/matchy/src/processing/mod.rs:353-448
//   FileReader::next_batch chunking  crates/matchy/src/processing/mod.rs:206-251
//   chunk_size_for                   crates/matchy/src/processing/parallel.rs:107-123
//   Database::lookup_extracted       crates/matchy/src/database.rs:889-901 (+ thread-local LRU :32-40)
//   CLI NDJSON record                crates/matchy/src/bin/match_processor/parallel.rs:297-369,
//                                    crates/matchy/src/bin/cli_utils.rs:107-141
#include <atomic>
#include <list>
#include <mutex>
#include <thread>
#include <unordered_map>

#include "database.h"
#include "extractor.h"

using namespace orc;

namespace {

const char* type_name(uint8_t t) {  // ExtractedItem::type_name (matchy-extractor/src/lib.rs:261-272)
    switch (t) {
        case IT_DOMAIN: return "Domain"; case IT_EMAIL: return "Email"; case IT_IPV4: return "IPv4"; case IT_IPV6: return "IPv6";
        case IT_MD5: return "MD5"; case IT_SHA1: return "SHA1"; case IT_SHA256: return "SHA256"; case IT_SHA384: return "SHA384";
        case IT_SHA512: return "SHA512"; case IT_BITCOIN: return "Bitcoin"; case IT_ETHEREUM: return "Ethereum"; case IT_MONERO: return "Monero";
    }
    return "?";
}

struct Stats {  // WorkerStats (processing/mod.rs:86-128), counters only
    uint64_t lines = 0, candidates = 0, matches = 0, bytes = 0;
    uint64_t by_type[12] = {0};
};

// One emitted match, position made absolute for canonical ordering (the reference's byte_offset is chunk-relative).
struct Hit {
    uint64_t abs_start, abs_end;
    uint8_t type;
    uint8_t ip[16];
    QueryResult qr;
};

struct Lru {  // lru crate semantic: get promotes, put evicts least-recent; stores NotFound too
    size_t cap;
    std::list<std::pair<std::string, QueryResult>> items;
    std::unordered_map<std::string, std::list<std::pair<std::string, QueryResult>>::iterator> idx;
    explicit Lru(size_t c) : cap(c) {}
    const QueryResult* get(const std::string& k) {
        auto it = idx.find(k);
        if (it == idx.end()) return nullptr;
        items.splice(items.begin(), items, it->second);
        return &it->second->second;
    }
    void put(const std::string& k, const QueryResult& v) {
        auto it = idx.find(k);
        if (it != idx.end()) { it->second->second = v; items.splice(items.begin(), items, it->second); return; }
        items.emplace_front(k, v);
        idx[k] = items.begin();
        if (items.size() > cap) { idx.erase(items.back().first); items.pop_back(); }
    }
};

struct Oracle {
    Psl psl;
    bool psl_ok = false;
};
Oracle g;

// Worker::process_bytes (processing/mod.rs:353-448) for one database.
void process_bytes(const Database& db, const Extractor& ex, const uint8_t* data, size_t n, uint64_t base, Lru* cache,
                   std::vector<Hit>& out, Stats& st) {
    for (size_t i = 0; i < n; ++i) st.lines += data[i] == '\n';
    st.bytes += n;
    std::vector<Match> ms;
    ex.extract_from_chunk(data, n, ms);
    for (const Match& m : ms) {
        st.candidates++;
        st.by_type[m.type]++;
        QueryResult qr;
        // Database::lookup_extracted (database.rs:889-901)
        if (m.type == IT_IPV4 || m.type == IT_IPV6) {
            if (cache) {  // lookup_ip: key = addr.to_string() (database.rs:837-855)
                std::string key = m.type == IT_IPV4 ? fmt_ipv4(m.ip) : fmt_ipv6(m.ip);
                if (const QueryResult* c = cache->get(key)) qr = *c;
                else { qr = db.lookup_ip(m.type == IT_IPV6, m.ip); if (qr.kind != QueryResult::NONE) cache->put(key, qr); }
            } else qr = db.lookup_ip(m.type == IT_IPV6, m.ip);
        } else {
            if (cache) {  // lookup: key = query string (database.rs:725-804)
                std::string key((const char*)data + m.start, m.end - m.start);
                if (const QueryResult* c = cache->get(key)) qr = *c;
                else { qr = db.lookup_string(data + m.start, m.end - m.start); if (qr.kind != QueryResult::NONE) cache->put(key, qr); }
            } else qr = db.lookup_string(data + m.start, m.end - m.start);
        }
        if (qr.kind == QueryResult::NONE || qr.kind == QueryResult::NOT_FOUND) continue;
        st.matches++;
        Hit h;
        h.abs_start = base + m.start; h.abs_end = base + m.end; h.type = m.type;
        memcpy(h.ip, m.ip, 16);
        h.qr = std::move(qr);
        out.push_back(std::move(h));
    }
}

// format_cidr_into (bin/cli_utils.rs:107-141) given the already-parsed address
std::string format_cidr(uint8_t type, const uint8_t ip[16], uint8_t prefix) {
    uint8_t net[16];
    int nbytes = type == IT_IPV4 ? 4 : 16;
    for (int i = 0; i < nbytes; ++i) {
        int bits_left = (int)prefix - i * 8;
        uint8_t mask = bits_left >= 8 ? 0xFF : bits_left <= 0 ? 0 : (uint8_t)(0xFF << (8 - bits_left));
        net[i] = ip[i] & mask;
    }
    return (type == IT_IPV4 ? fmt_ipv4(net) : fmt_ipv6(net)) + "/" + std::to_string((unsigned)prefix);
}

// library_match_to_cli_match + output_cli_match (match_processor/parallel.rs:297-369); keys sorted (serde_json BTreeMap)
std::string hit_to_json(const Database& db, const Hit& h, const uint8_t* text, const std::string& source) {
    std::string o = "{";
    std::string matched((const char*)text, h.abs_end - h.abs_start);
    if (h.qr.kind == QueryResult::IP) {
        o += "\"cidr\":"; json_escape(format_cidr(h.type, h.ip, h.qr.prefix_len), o);
        o += ",\"data\":";
        if (!db.data_json(h.qr.ip_data_offset, o)) o += "null";
        o += ",\"match_type\":\"ip\",\"matched_text\":"; json_escape(matched, o);
        o += ",\"prefix_len\":" + std::to_string((unsigned)h.qr.prefix_len);
    } else {
        std::string arr;
        bool any = false;
        for (int64_t off : h.qr.data_offsets) {
            if (off < 0) continue;
            std::string one;
            if (!db.data_json((uint32_t)off, one)) one = "null";
            if (any) arr.push_back(',');
            arr += one;
            any = true;
        }
        if (any) o += "\"data\":[" + arr + "],";
        o += "\"match_type\":\"pattern\",\"matched_text\":"; json_escape(matched, o);
        o += ",\"pattern_count\":" + std::to_string(h.qr.pattern_ids.size());
    }
    o += ",\"source\":"; json_escape(source, o);
    o += ",\"timestamp\":\"0.000\"}";
    return o;
}

struct ScanResult {
    std::vector<Hit> hits;
    Stats stats;
    std::string ndjson;  // canonical order: (abs_start, chunk-path type rank)
};

int type_rank(uint8_t t) {  // chunk-path extractor order (lib.rs:449-485)
    switch (t) {
        case IT_IPV6: return 0; case IT_IPV4: return 1; case IT_EMAIL: return 2; case IT_DOMAIN: return 3;
        case IT_MD5: case IT_SHA1: case IT_SHA256: case IT_SHA384: case IT_SHA512: return 4;
        case IT_BITCOIN: return 5; case IT_ETHEREUM: return 6; case IT_MONERO: return 7;
    }
    return 8;
}

}  // namespace

extern "C" {

// ---- init: load the PSL container once. Returns 0 on success.
int orc_init(const char* psl_path) {
    if (g.psl_ok) return 0;
    g.psl_ok = g.psl.load(psl_path);
    // the Unicode lower-case table lies beside the suffix list (case-insensitive databases only)
    std::string lp = psl_path;
    const size_t k = lp.rfind('/');
    lp = (k == std::string::npos ? std::string() : lp.substr(0, k + 1)) + "lowercase.bin";
    try { Lowercase::table().load(lp.c_str()); } catch (const std::exception&) {}
    return g.psl_ok ? 0 : -1;
}
// Rust str::to_lowercase of valid UTF-8; returns the length of the result (call again with a larger buffer if > cap)
size_t orc_to_lowercase(const uint8_t* s, size_t n, uint8_t* out, size_t cap) {
    const std::string r = Lowercase::table().to_lowercase(std::string((const char*)s, n));
    if (r.size() <= cap) memcpy(out, r.data(), r.size());
    return r.size();
}
size_t orc_psl_count() { return g.psl.set.size(); }
int orc_psl_contains(const char* s, size_t n) { return g.psl.contains((const uint8_t*)s, n) ? 1 : 0; }

// ---- primitives exposed for known-answer tests
uint64_t orc_xxh64(const uint8_t* p, size_t n, uint64_t seed) { return xxh64(p, n, seed); }
void orc_sha256(const uint8_t* p, size_t n, uint8_t out[32]) { sha256(p, n, out); }
void orc_keccak256(const uint8_t* p, size_t n, uint8_t out[32]) { keccak256(p, n, out); }

// ---- extractor. out arrays sized cap; returns number of matches (may exceed cap → call again with larger cap)
struct orc_match_t { uint8_t type; uint8_t ip[16]; uint64_t start, end; };
size_t orc_extract_chunk(uint32_t flags, uint32_t min_labels, const uint8_t* data, size_t n, orc_match_t* out, size_t cap) {
    Extractor ex;
    ex.psl = &g.psl; ex.flags = flags; ex.min_domain_labels = min_labels ? min_labels : 2;
    std::vector<Match> ms;
    ex.extract_from_chunk(data, n, ms);
    for (size_t i = 0; i < ms.size() && i < cap; ++i) {
        out[i].type = ms[i].type; memcpy(out[i].ip, ms[i].ip, 16); out[i].start = ms[i].start; out[i].end = ms[i].end;
    }
    return ms.size();
}
const char* orc_type_name(uint8_t t) { return type_name(t); }
// canonical as_value() strings for IPs (lib.rs:300-311)
size_t orc_format_ip(int v6, const uint8_t* ip, char* out, size_t cap) {
    std::string s = v6 ? fmt_ipv6(ip) : fmt_ipv4(ip);
    if (s.size() + 1 <= cap) memcpy(out, s.c_str(), s.size() + 1);
    return s.size();
}
int orc_parse_ipv6(const uint8_t* s, size_t n, uint8_t out[16]) {
    uint16_t seg[8];
    if (!parse_ipv6_rust(s, n, seg)) return 0;
    for (int i = 0; i < 8; ++i) { out[2 * i] = (uint8_t)(seg[i] >> 8); out[2 * i + 1] = (uint8_t)seg[i]; }
    return 1;
}

// ---- database
void* orc_db_open(const uint8_t* bytes, size_t n, char* err, size_t errcap) {
    auto* db = new Database();
    if (!db->open(bytes, n)) {
        if (err && errcap) snprintf(err, errcap, "%s", db->error.c_str());
        delete db;
        return nullptr;
    }
    return db;
}
void orc_db_close(void* h) { delete (Database*)h; }
int orc_db_info(void* h, uint32_t* node_count, int* record_size, int* ip_version, int* has_literal, int* has_glob) {
    auto* db = (Database*)h;
    *node_count = db->node_count; *record_size = db->record_size; *ip_version = db->ip_version;
    *has_literal = db->has_literal; *has_glob = db->has_glob;
    return 0;
}
// Database::lookup (database.rs:725-804) for one query string: parse as IpAddr first, else string lookup.
// Writes a JSON description {"kind":"ip"|"pattern"|"notfound"|"none", ...} — test helper.
size_t orc_db_lookup_json(void* h, const char* q, size_t qn, char* out, size_t cap) {
    auto* db = (Database*)h;
    QueryResult qr;
    uint8_t ip[16];
    uint8_t type = 0xFF;
    {
        Extractor ex; ex.psl = &g.psl; ex.require_word_boundaries = false;
        uint8_t oct[4]; size_t end;
        uint16_t seg[8];
        // Rust Ipv4Addr::from_str: 4 decimal octets, no leading zeros, whole string
        if (ex.try_parse_ipv4((const uint8_t*)q, qn, 0, oct, end) && end == qn) { memcpy(ip, oct, 4); type = IT_IPV4; }
        else if (memchr(q, '.', qn) == nullptr && parse_ipv6_rust((const uint8_t*)q, qn, seg)) {
            for (int i = 0; i < 8; ++i) { ip[2 * i] = (uint8_t)(seg[i] >> 8); ip[2 * i + 1] = (uint8_t)seg[i]; }
            type = IT_IPV6;
        }
    }
    if (type == IT_IPV4 || type == IT_IPV6) qr = db->lookup_ip(type == IT_IPV6, ip);
    else qr = db->lookup_string((const uint8_t*)q, qn);
    std::string o;
    if (qr.kind == QueryResult::IP) {
        o = "{\"kind\":\"ip\",\"prefix_len\":" + std::to_string((unsigned)qr.prefix_len) + ",\"data\":";
        if (!db->data_json(qr.ip_data_offset, o)) o += "null";
        o += "}";
    } else if (qr.kind == QueryResult::PATTERN) {
        o = "{\"kind\":\"pattern\",\"pattern_ids\":[";
        for (size_t i = 0; i < qr.pattern_ids.size(); ++i) { if (i) o += ","; o += std::to_string(qr.pattern_ids[i]); }
        o += "],\"data\":[";
        for (size_t i = 0; i < qr.data_offsets.size(); ++i) {
            if (i) o += ",";
            if (qr.data_offsets[i] < 0 || !db->data_json((uint32_t)qr.data_offsets[i], o)) o += "null";
        }
        o += "]}";
    } else if (qr.kind == QueryResult::NOT_FOUND) o = "{\"kind\":\"notfound\"}";
    else o = "{\"kind\":\"none\"}";
    if (o.size() + 1 <= cap) memcpy(out, o.c_str(), o.size() + 1);
    return o.size();
}
size_t orc_db_metadata_json(void* h, char* out, size_t cap) {
    auto* db = (Database*)h;
    std::string o;
    value_to_json(db->metadata, o);
    if (o.size() + 1 <= cap) memcpy(out, o.c_str(), o.size() + 1);
    return o.size();
}

// ---- scan: the reference's parallel/chunk path over one in-memory input.
//   chunk_bytes == 0 → chunk_size_for(len) (parallel.rs:107-123); threads >= 1; cache_cap 0 disables the LRU.
//   extract_flags == 0 → derive from DB capabilities like match_cmd.rs:276-303.
struct orc_scan_stats_t { uint64_t lines, candidates, matches, bytes, by_type[12]; double seconds; uint64_t chunks; };
void* orc_scan(void* h, const uint8_t* data, size_t n, uint32_t extract_flags, size_t chunk_bytes, int threads, size_t cache_cap,
               const char* source, int want_json, orc_scan_stats_t* stats_out) {
    auto* db = (Database*)h;
    auto* res = new ScanResult();
    if (extract_flags == 0) {
        if (db->has_ip) extract_flags |= EX_IPV4 | EX_IPV6;
        if (db->has_literal || db->has_glob) extract_flags |= EX_DOMAINS | EX_EMAILS | EX_HASHES | EX_BITCOIN | EX_ETHEREUM | EX_MONERO;
    }
    if (chunk_bytes == 0) chunk_bytes = n < (1ull << 30) ? 256 * 1024 : n < (10ull << 30) ? 1024 * 1024 : 4 * 1024 * 1024;
    // FileReader::next_batch (mod.rs:206-251): read chunk_bytes, cut at last '\n', carry the leftover.
    std::vector<std::pair<size_t, size_t>> chunks;  // [begin,end)
    {
        size_t begin = 0, read_pos = 0;
        while (read_pos < n) {
            size_t rd = std::min(chunk_bytes, n - read_pos);
            read_pos += rd;
            const uint8_t* base = data + begin;
            size_t avail = read_pos - begin;
            const void* nl = memrchr(base, '\n', avail);
            if (!nl) continue;  // no newline yet: keep accumulating
            size_t cut = (const uint8_t*)nl - base + 1;
            chunks.emplace_back(begin, begin + cut);
            begin += cut;
        }
        if (begin < n) chunks.emplace_back(begin, n);  // EOF leftover
    }
    if (threads < 1) threads = 1;
    std::vector<std::vector<Hit>> per_thread(threads);
    std::vector<Stats> per_stats(threads);
    std::atomic<size_t> next{0};
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        Extractor ex; ex.psl = &g.psl; ex.flags = extract_flags;
        std::unique_ptr<Lru> cache;
        if (cache_cap) cache.reset(new Lru(cache_cap));
        for (;;) {
            size_t ci = next.fetch_add(1);
            if (ci >= chunks.size()) break;
            process_bytes(*db, ex, data + chunks[ci].first, chunks[ci].second - chunks[ci].first, chunks[ci].first, cache.get(),
                          per_thread[tid], per_stats[tid]);
        }
    };
    if (threads == 1) worker(0);
    else {
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; ++t) ts.emplace_back(worker, t);
        for (auto& t : ts) t.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; ++t) {
        for (auto& hh : per_thread[t]) res->hits.push_back(std::move(hh));
        res->stats.lines += per_stats[t].lines; res->stats.candidates += per_stats[t].candidates;
        res->stats.matches += per_stats[t].matches; res->stats.bytes += per_stats[t].bytes;
        for (int k = 0; k < 12; ++k) res->stats.by_type[k] += per_stats[t].by_type[k];
    }
    std::stable_sort(res->hits.begin(), res->hits.end(), [](const Hit& a, const Hit& b) {
        if (a.abs_start != b.abs_start) return a.abs_start < b.abs_start;
        int ra = type_rank(a.type), rb = type_rank(b.type);
        if (ra != rb) return ra < rb;
        return a.abs_end < b.abs_end;
    });
    if (want_json) {
        std::string src = source ? source : "-";
        for (const Hit& hh : res->hits) { res->ndjson += hit_to_json(*db, hh, data + hh.abs_start, src); res->ndjson.push_back('\n'); }
    }
    if (stats_out) {
        stats_out->lines = res->stats.lines; stats_out->candidates = res->stats.candidates; stats_out->matches = res->stats.matches;
        stats_out->bytes = res->stats.bytes;
        for (int k = 0; k < 12; ++k) stats_out->by_type[k] = res->stats.by_type[k];
        stats_out->seconds = std::chrono::duration<double>(t1 - t0).count();
        stats_out->chunks = chunks.size();
    }
    return res;
}
size_t orc_scan_hit_count(void* r) { return ((ScanResult*)r)->hits.size(); }
// flat hit record for set comparison with the device path
struct orc_hit_t { uint64_t start, end; uint8_t type, kind, prefix_len, pad; uint32_t ip_data_offset; uint32_t n_ids; };
void orc_scan_hit(void* r, size_t i, orc_hit_t* out, uint32_t* ids, int64_t* offs, size_t cap) {
    const Hit& h = ((ScanResult*)r)->hits[i];
    out->start = h.abs_start; out->end = h.abs_end; out->type = h.type; out->kind = (uint8_t)h.qr.kind; out->prefix_len = h.qr.prefix_len;
    out->pad = 0; out->ip_data_offset = h.qr.ip_data_offset; out->n_ids = (uint32_t)h.qr.pattern_ids.size();
    for (size_t k = 0; k < h.qr.pattern_ids.size() && k < cap; ++k) { ids[k] = h.qr.pattern_ids[k]; offs[k] = h.qr.data_offsets[k]; }
}
const char* orc_scan_ndjson(void* r, size_t* len) { auto* s = (ScanResult*)r; *len = s->ndjson.size(); return s->ndjson.data(); }
void orc_scan_free(void* r) { delete (ScanResult*)r; }

}  // extern "C"
