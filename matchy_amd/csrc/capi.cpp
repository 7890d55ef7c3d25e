// extern "C" surface of libmatchy_amd.so — see include/matchy_amd.h for the per-function reference citations.
#include "../../include/matchy_amd.h"

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "db_builder.h"
#include "db_image.h"
#include "engine.h"
#include "host_lookup.h"
#include "host_topology.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <thread>
#include "netaddr.h"

#include <list>
#include <unordered_map>
#include <unordered_set>

using namespace mxy;

namespace {

thread_local std::string g_last_error;
void set_error(const std::string& e) { g_last_error = e; }

struct Builder {
    DatabaseBuilder b;
};

struct Db {
    std::shared_ptr<DbImage> img;
    std::mutex mu;                                   // guards dev + query scanner (single queries are serialised)
    std::vector<std::shared_ptr<DeviceDb>> dev;      // per device ordinal, uploaded on first use
    std::unique_ptr<Scanner> query_scanner;        // single queries through the lookup kernels (MATCHY_AMD_QUERY_ON_GPU=1: tests)
    std::unique_ptr<HostTables> host_tables;       // single queries on the host (host_lookup.h): built by the first query
    DevBuf<uint8_t> qbuf;
    DevBuf<Candidate> qcand;
    std::string format;
    int default_device = 0;
    // DatabaseStats (database.rs:728-795)
    mutable std::atomic<uint64_t> st_total{0}, st_match{0}, st_nomatch{0}, st_ip{0}, st_str{0}, st_chit{0}, st_cmiss{0};
    // Query cache of Database::lookup (database.rs:32-40, 384-407, 725-804): LRU keyed by the query string, holds "not found" too.
    // ONE PER THREAD AND HANDLE, like the reference's thread-local caches (round 5; until then one per handle behind `mu`, which made
    // eight querying threads slower than one): a query takes no lock. 0 entries = disabled (matchy_open_options_t.cache_capacity).
    struct Cached { int kind; uint8_t prefix_len; bool has_data; DataValue data; };   // kind: 0 not found, 2 IP, 3 pattern
    struct Lru {
        std::list<std::pair<std::string, Cached>> lru;
        std::unordered_map<std::string, std::list<std::pair<std::string, Cached>>::iterator> index;
        const Cached* get(const std::string& q) {
            auto it = index.find(q);
            if (it == index.end()) return nullptr;
            lru.splice(lru.begin(), lru, it->second);
            return &it->second->second;
        }
        void put(const std::string& q, Cached&& c, size_t cap) {
            if (!cap) return;
            auto it = index.find(q);
            if (it != index.end()) { it->second->second = std::move(c); lru.splice(lru.begin(), lru, it->second); return; }
            lru.emplace_front(q, std::move(c));
            index[q] = lru.begin();
            if (lru.size() > cap) { index.erase(lru.back().first); lru.pop_back(); }
        }
        void clear() { lru.clear(); index.clear(); }
    };
    size_t cache_cap = 10000;
    const uint64_t uid = next_uid();   // names this handle in the threads' cache maps (a pointer could be reused by a later handle)
    static uint64_t next_uid() { static std::atomic<uint64_t> n{1}; return n.fetch_add(1); }
    static std::mutex& live_mu() { static std::mutex m; return m; }
    static std::unordered_set<uint64_t>& live() { static std::unordered_set<uint64_t> s; return s; }
    Db() { std::lock_guard<std::mutex> lk(live_mu()); live().insert(uid); }
    ~Db() { std::lock_guard<std::mutex> lk(live_mu()); live().erase(uid); }
    // the calling thread's cache of this handle; caches of closed handles are dropped when a thread meets a new handle
    Lru& cache() {
        thread_local std::unordered_map<uint64_t, Lru> mine;
        auto it = mine.find(uid);
        if (it != mine.end()) return it->second;
        if (mine.size() >= 4) {
            std::lock_guard<std::mutex> lk(live_mu());
            for (auto k = mine.begin(); k != mine.end();) k = live().count(k->first) ? std::next(k) : mine.erase(k);
        }
        return mine[uid];
    }
    std::once_flag host_tables_once;

    std::shared_ptr<DeviceDb> device_db(int device) {
        if ((int)dev.size() <= device) dev.resize(device + 1);
        if (!dev[device]) {
            auto d = std::make_shared<DeviceDb>();
            d->upload(*img, device);
            dev[device] = d;
        }
        return dev[device];
    }
};

struct ExtractorH {
    std::shared_ptr<DbImage> img;  // empty image: only the PSL tables are needed
    std::shared_ptr<DeviceDb> ddb;
    std::unique_ptr<Scanner> scanner;
    std::mutex mu;
    uint32_t flags;
};

struct MatchesInternal {
    std::vector<matchy_match_t> items;
    std::vector<std::string> values;
};

struct ScannerH {
    const Db* db;
    std::unique_ptr<Scanner> sc;
    // matchy_scan_result_to_ndjson: the JSON text of an entry's data by its data-section offset (indicator lists carry a few hundred
    // distinct payloads for millions of hits: decoding the MMDB value and serialising it per hit was most of the rendering time)
    std::unordered_map<uint32_t, std::string> json_of_data;
    std::string json_source, json_source_of;   // the last source name, JSON-escaped
    // matchy_scanner_submit_device -> matchy_scanner_wait
    bool pending = false;
    size_t pending_len = 0;
    uint32_t pending_mode = 0;
    void* pending_stream = nullptr;
};

struct ScanResultInternal {
    std::vector<matchy_scan_hit_t> hits;
    std::vector<uint32_t> ids;
    std::vector<int64_t> offs;
    bool on_device = false;   // MATCHY_SCAN_FETCH_DEVICE: the result's arrays are device pointers
};

int type_rank(uint32_t t) {  // chunk-path extractor order (matchy-extractor/src/lib.rs:449-485)
    switch (t) {
        case IT_IPV6: return 0; case IT_IPV4: return 1; case IT_EMAIL: return 2; case IT_DOMAIN: return 3;
        case IT_MD5: case IT_SHA1: case IT_SHA256: case IT_SHA384: case IT_SHA512: return 4;
        case IT_BITCOIN: return 5; case IT_ETHEREUM: return 6; case IT_MONERO: return 7;
    }
    return 8;
}

bool valid_utf8_host(const uint8_t* s, size_t n) {
    size_t i = 0;
    while (i < n) {
        uint8_t c = s[i];
        if (c < 0x80) { ++i; continue; }
        size_t l = (c >= 0xC2 && c <= 0xDF) ? 2 : (c >= 0xE0 && c <= 0xEF) ? 3 : (c >= 0xF0 && c <= 0xF4) ? 4 : 0;
        if (l == 0 || i + l > n) return false;
        uint8_t lo = 0x80, hi = 0xBF;
        if (c == 0xE0) lo = 0xA0;
        if (c == 0xED) hi = 0x9F;
        if (c == 0xF0) lo = 0x90;
        if (c == 0xF4) hi = 0x8F;
        if (s[i + 1] < lo || s[i + 1] > hi) return false;
        for (size_t k = 2; k < l; ++k) if ((s[i + k] & 0xC0) != 0x80) return false;
        i += l;
    }
    return true;
}

const char* item_type_name(uint8_t t) {
    static const char* N[] = {"Domain", "Email", "IPv4", "IPv6", "MD5", "SHA1", "SHA256", "SHA384", "SHA512", "Bitcoin", "Ethereum", "Monero"};
    return t < 12 ? N[t] : "Unknown";
}

bool read_file(const char* path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    uint8_t tmp[1 << 16];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) out.insert(out.end(), tmp, tmp + n);
    fclose(f);
    return true;
}

matchy_t* open_bytes(std::vector<uint8_t>&& bytes) {
    try {
        const bool trace = getenv("MATCHY_AMD_TRACE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        auto db = std::make_unique<Db>();
        db->img = std::make_shared<DbImage>();
        std::string err;
        if (!db->img->open(std::move(bytes), err)) { set_error(err); return nullptr; }
        db->format = db->img->format_name();
        if (trace) fprintf(stderr, "[matchy_amd] open: parsed and checked after %.1f ms\n", ms());
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
            set_error("matchy_amd: no HIP device available (scans and extraction run on the GPU only; a handle is not opened without one)");
            return nullptr;
        }
        // "uploaded once to device memory" at open. One process per GPU selects its device with MATCHY_AMD_DEVICE.
        int dev0 = 0;
        if (const char* e = getenv("MATCHY_AMD_DEVICE")) dev0 = atoi(e);
        if (dev0 < 0 || dev0 >= ndev) { set_error("MATCHY_AMD_DEVICE out of range"); return nullptr; }
        db->default_device = dev0;
        if (trace) { (void)hipSetDevice(dev0); (void)hipFree(nullptr); fprintf(stderr, "[matchy_amd] open: HIP runtime up after %.1f ms\n", ms()); }
        db->device_db(dev0);
        if (trace) fprintf(stderr, "[matchy_amd] open: uploaded after %.1f ms\n", ms());
        return reinterpret_cast<matchy_t*>(db.release());
    } catch (const HipError& e) { set_error(e.what); return nullptr; }
    catch (const std::exception& e) { set_error(e.what()); return nullptr; }
}

static_assert(sizeof(matchy_scan_ip4_hit_t) == sizeof(uint2) && MATCHY_SCAN_IP4_DATA_BITS == C4_DATA_BITS && MATCHY_ITEM_TYPE_IPV4 == IT_IPV4, "compact record layout");
static_assert(sizeof(FinalHit) == sizeof(matchy_scan_hit_t) && sizeof(FinalHit) == 16 && offsetof(FinalHit, value) == offsetof(matchy_scan_hit_t, value) &&
                  offsetof(FinalHit, n_ids) == offsetof(matchy_scan_hit_t, n_ids) && offsetof(FinalHit, kind) == offsetof(matchy_scan_hit_t, kind),
              "pack_record writes matchy_scan_hit_t records directly");

// Hand the dense records to the caller. borrowed=true: pointers go straight to the scanner's pinned buffers (no per-hit
// host work at all); otherwise the arrays are copied into the result and optionally put into canonical order.
void fill_result(const FinalHit* fin, size_t n_fin, const uint32_t* ids, const long long* offs, size_t n_ids, uint64_t lines, uint64_t cands,
                 uint64_t bytes, bool borrowed, bool sorted, matchy_scan_result_t* out) {
    memset(out, 0, sizeof(*out));
    out->lines = lines; out->candidates = cands; out->bytes = bytes;
    out->n_hits = n_fin; out->n_ids = n_ids;
    if (borrowed) {
        out->hits = reinterpret_cast<const matchy_scan_hit_t*>(fin);
        out->pattern_ids = ids;
        out->data_offsets = reinterpret_cast<const int64_t*>(offs);
        return;
    }
    auto* in = new ScanResultInternal();
    in->hits.resize(n_fin);
    if (n_fin) memcpy(in->hits.data(), fin, n_fin * sizeof(FinalHit));
    in->ids.assign(ids, ids + n_ids);
    in->offs.assign(offs, offs + n_ids);
    (void)sorted;  // canonical order is produced on the GPU (sort_hits.hip); the copy keeps it
    out->hits = in->hits.data();
    out->pattern_ids = in->ids.data(); out->data_offsets = in->offs.data();
    out->_internal = in;
}

// MATCHY_SCAN_FETCH_COMPACT only means something for unsorted host-resident records
bool compact_mode(uint32_t fetch_mode) { return (fetch_mode & MATCHY_SCAN_FETCH_COMPACT) && (fetch_mode & 7u) == MATCHY_SCAN_FETCH_HITS; }

// fetch_mode MATCHY_SCAN_FETCH_DEVICE: the records stay where the lookup kernel wrote them (device memory); only the counters
// come back.
void fill_device_result(Scanner& sc, const ScanOutput& so, uint64_t bytes, matchy_scan_result_t* out) {
    memset(out, 0, sizeof(*out));
    out->lines = so.lines; out->candidates = so.n_cand; out->bytes = bytes;
    out->n_hits = so.n_hits; out->n_ids = sc.device_final_id_count();
    out->hits = reinterpret_cast<const matchy_scan_hit_t*>(sc.device_final());
    out->pattern_ids = sc.device_final_ids();
    out->data_offsets = reinterpret_cast<const int64_t*>(sc.device_final_offs());
    auto* in = new ScanResultInternal();   // marks the residency: host-side readers of the result refuse device pointers
    in->on_device = true;
    out->_internal = in;
}

}  // namespace

extern "C" {

const char* matchy_amd_last_error(void) { return g_last_error.c_str(); }
int32_t matchy_amd_ac_dfa_states(const matchy_t* db_) {
    if (!db_) return -1;
    Db* db = const_cast<Db*>(reinterpret_cast<const Db*>(db_));
    try {
        std::lock_guard<std::mutex> lk(db->mu);
        return (int32_t)std::min<uint32_t>(db->device_db(db->default_device)->view.dfa ? db->device_db(db->default_device)->view.dfa_states : 0u, 0x7FFFFFFFu);
    } catch (...) { return -1; }
}
int32_t matchy_amd_suffix_filter(const matchy_t* db_) {
    if (!db_) return -1;
    Db* db = const_cast<Db*>(reinterpret_cast<const Db*>(db_));
    try {
        std::lock_guard<std::mutex> lk(db->mu);
        return db->device_db(db->default_device)->view.sfx_bm ? 1 : 0;
    } catch (...) { return -1; }
}
void* matchy_amd_pinned_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); set_error("matchy_amd_pinned_alloc: hipHostMalloc failed"); return nullptr; }
    return p;
}
void matchy_amd_pinned_free(void* p) { if (p) (void)hipHostFree(p); }
int32_t matchy_amd_host_register(const void* ptr, size_t bytes) {
    if (!ptr || !bytes) return MATCHY_ERROR_INVALID_PARAM;
    return mxy::pins::add_caller(ptr, bytes) == 0 ? MATCHY_SUCCESS : MATCHY_ERROR_IO;
}
void matchy_amd_host_unregister(const void* ptr) { if (ptr) mxy::pins::remove_caller(ptr); }
int32_t matchy_amd_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
const char* matchy_version(void) { return "matchy-amd 0.1.0 (reference matchy 1.2.2 surface)"; }

// ------------------------------------------------------------------------------------------------ builder
matchy_builder_t* matchy_builder_new(void) { return reinterpret_cast<matchy_builder_t*>(new Builder()); }
void matchy_builder_free(matchy_builder_t* b) { delete reinterpret_cast<Builder*>(b); }
int32_t matchy_builder_set_case_insensitive(matchy_builder_t* b, bool ci) {
    if (!b) return MATCHY_ERROR_INVALID_PARAM;
    reinterpret_cast<Builder*>(b)->b.set_case_insensitive(ci);
    return MATCHY_SUCCESS;
}
int32_t matchy_builder_set_build_epoch(matchy_builder_t* b, uint64_t epoch) {
    if (!b) return MATCHY_ERROR_INVALID_PARAM;
    reinterpret_cast<Builder*>(b)->b.set_build_epoch(epoch);
    return MATCHY_SUCCESS;
}
int32_t matchy_builder_add(matchy_builder_t* b, const char* key, const char* json_data) {
    if (!b || !key || !json_data) return MATCHY_ERROR_INVALID_PARAM;
    DataValue v;
    std::string err;
    if (!parse_json(json_data, strlen(json_data), NumberTyping::SERDE, v, err)) { set_error(err); return MATCHY_ERROR_INVALID_FORMAT; }
    if (v.type != DataValue::MAP) {  // single value: wrapped as {"value": v} (c_api/matchy.rs:427-435)
        DataValue m = DataValue::Map();
        m.map["value"] = std::move(v);
        v = std::move(m);
    }
    Builder* bb = reinterpret_cast<Builder*>(b);
    if (!bb->b.add_entry(key, v)) { set_error(bb->b.error()); return MATCHY_ERROR_INVALID_FORMAT; }
    return MATCHY_SUCCESS;
}
int32_t matchy_builder_set_description(matchy_builder_t* b, const char* description) {
    if (!b || !description) return MATCHY_ERROR_INVALID_PARAM;
    reinterpret_cast<Builder*>(b)->b.set_description("en", description);
    return MATCHY_SUCCESS;
}
int32_t matchy_builder_build(matchy_builder_t* b, uint8_t** buffer, uintptr_t* size) {
    if (!b || !buffer || !size) return MATCHY_ERROR_INVALID_PARAM;
    Builder* bb = reinterpret_cast<Builder*>(b);
    std::vector<uint8_t> out;
    if (!bb->b.build(out)) { set_error(bb->b.error()); return MATCHY_ERROR_INVALID_FORMAT; }
    uint8_t* p = (uint8_t*)malloc(out.size() ? out.size() : 1);
    if (!p) return MATCHY_ERROR_OUT_OF_MEMORY;
    memcpy(p, out.data(), out.size());
    *buffer = p;
    *size = out.size();
    return MATCHY_SUCCESS;
}
int32_t matchy_builder_save(matchy_builder_t* b, const char* filename) {
    if (!b || !filename) return MATCHY_ERROR_INVALID_PARAM;
    Builder* bb = reinterpret_cast<Builder*>(b);
    std::vector<uint8_t> out;
    if (!bb->b.build(out)) { set_error(bb->b.error()); return MATCHY_ERROR_INVALID_FORMAT; }
    FILE* f = fopen(filename, "wb");
    if (!f) return MATCHY_ERROR_IO;
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    ok = fclose(f) == 0 && ok;
    return ok ? MATCHY_SUCCESS : MATCHY_ERROR_IO;
}

// ------------------------------------------------------------------------------------------------ open / close
void matchy_init_open_options(matchy_open_options_t* o) {
    if (!o) return;
    o->cache_capacity = 10000; o->auto_reload = false; o->reload_callback = nullptr; o->reload_callback_user_data = nullptr;
}
matchy_t* matchy_open(const char* filename) {
    if (!filename) return nullptr;
    std::vector<uint8_t> bytes;
    if (!read_file(filename, bytes)) { set_error(std::string("Failed to open ") + filename); return nullptr; }
    return open_bytes(std::move(bytes));
}
matchy_t* matchy_open_with_options(const char* filename, const matchy_open_options_t* options) {
    if (!filename || !options) return nullptr;
    matchy_t* db = matchy_open(filename);
    if (db) reinterpret_cast<Db*>(db)->cache_cap = options->cache_capacity;   // 0 disables the query cache (c_api/matchy.rs:805-808)
    return db;
}
matchy_t* matchy_open_buffer(const uint8_t* buffer, uintptr_t size) {
    if (!buffer || size == 0) return nullptr;
    return open_bytes(std::vector<uint8_t>(buffer, buffer + size));
}
void matchy_close(matchy_t* db) { delete reinterpret_cast<Db*>(db); }

const char* matchy_format(const matchy_t* db) { return db ? reinterpret_cast<const Db*>(db)->format.c_str() : nullptr; }
bool matchy_has_ip_data(const matchy_t* db) { return db && reinterpret_cast<const Db*>(db)->img->has_ip; }
bool matchy_has_literal_data(const matchy_t* db) { return db && reinterpret_cast<const Db*>(db)->img->has_literal; }
bool matchy_has_glob_data(const matchy_t* db) { return db && reinterpret_cast<const Db*>(db)->img->has_glob; }
bool matchy_has_string_data(const matchy_t* db) { return matchy_has_literal_data(db) || matchy_has_glob_data(db); }
bool matchy_has_pattern_data(const matchy_t* db) { return matchy_has_string_data(db); }

void matchy_get_stats(const matchy_t* dbc, matchy_stats_t* st) {
    if (!dbc || !st) return;
    const Db* db = reinterpret_cast<const Db*>(dbc);
    *st = matchy_stats_t{db->st_total.load(), db->st_match.load(), db->st_nomatch.load(), db->st_chit.load(), db->st_cmiss.load(), db->st_ip.load(),
                         db->st_str.load()};
}
void matchy_clear_cache(const matchy_t* dbc) {
    if (!dbc) return;
    Db* db = const_cast<Db*>(reinterpret_cast<const Db*>(dbc));
    db->cache().clear();   // the calling thread's cache, like Database::clear_cache (thread-local caches)
}
uintptr_t matchy_pattern_count(const matchy_t* db) { return db ? reinterpret_cast<const Db*>(db)->img->pattern_count : 0; }
char* matchy_metadata(const matchy_t* db) {
    if (!db) return nullptr;
    std::string s;
    to_json(reinterpret_cast<const Db*>(db)->img->metadata, s);
    return strdup(s.c_str());
}
char* matchy_get_pattern_string(const matchy_t* db, uint32_t id) {
    if (!db) return nullptr;
    const DbImage& img = *reinterpret_cast<const Db*>(db)->img;
    if (!img.has_glob || id >= img.pattern_count) return nullptr;
    return strdup(img.pattern_string(id).c_str());
}
void matchy_free_string(char* s) { free(s); }

// ------------------------------------------------------------------------------------------------ single query
// Database::lookup (database.rs:725-804): a query that parses as an IP address takes the trie, anything else the string
// path. Fills the one-candidate "scan" both single-query entries run through the lookup kernels. False: query too long.
static bool query_candidate(const char* query, size_t qn, IpAddr& ip, bool& is_ip, std::string& text, Candidate& c) {
    text.assign(query, qn);
    c = Candidate{0, 0, 0, 0};
    is_ip = parse_ip(query, qn, ip);
    if (is_ip) {
        if (!ip.v6) {
            c.v4 = ((uint32_t)ip.b[0] << 24) | ((uint32_t)ip.b[1] << 16) | ((uint32_t)ip.b[2] << 8) | ip.b[3];
            c.len_type = (uint32_t)qn | ((uint32_t)IT_IPV4 << 24);
        } else {
            char buf[64];
            snprintf(buf, sizeof(buf), "%x:%x:%x:%x:%x:%x:%x:%x", (ip.b[0] << 8) | ip.b[1], (ip.b[2] << 8) | ip.b[3], (ip.b[4] << 8) | ip.b[5],
                     (ip.b[6] << 8) | ip.b[7], (ip.b[8] << 8) | ip.b[9], (ip.b[10] << 8) | ip.b[11], (ip.b[12] << 8) | ip.b[13], (ip.b[14] << 8) | ip.b[15]);
            text = buf;
            c.len_type = (uint32_t)text.size() | ((uint32_t)IT_IPV6 << 24);
        }
        return true;
    }
    if (qn >= (1u << 24)) return false;
    c.len_type = (uint32_t)qn | ((uint32_t)IT_DOMAIN << 24);
    return true;
}

// One uncached query. SURVEY §8b: single queries stay on the CPU (latency) — host_lookup.cpp walks the same sections the kernels get
// uploaded; MATCHY_AMD_QUERY_ON_GPU=1 sends the query through the lookup kernels instead (the GPU tests hold the two against each other
// and against the oracle). Fills `so` like Scanner::lookup_one: no hit, or hits[0] (+ its glob ids in so.ids).
// Takes no lock on the host path (the sections are read-only: queries of several threads proceed side by side, like the reference's
// per-thread lookups); the kernel path serialises on Db::mu.
static void single_lookup(Db* db, const std::string& text, const Candidate& c, const IpAddr& ip, bool is_ip, ScanOutput& so) {
    static const bool on_gpu = [] { const char* e = getenv("MATCHY_AMD_QUERY_ON_GPU"); return e && atoi(e) != 0; }();
    if (on_gpu) {
        std::lock_guard<std::mutex> lk(db->mu);
        if (!db->query_scanner) db->query_scanner = std::make_unique<Scanner>(db->img, db->device_db(db->default_device), EX_ALL, 2);
        db->query_scanner->lookup_one(text, c, so);
        return;
    }
    std::call_once(db->host_tables_once, [db] { db->host_tables = std::make_unique<HostTables>(*db->img); });
    HostHit hh;
    host_lookup(*db->img, *db->host_tables, text, is_ip ? &ip : nullptr, hh);
    so.hits.clear(); so.ids.clear();
    if (!hh.kind) return;
    if (hh.globs.size() > 0xFFFFu) throw std::runtime_error("query matches more than 65535 glob patterns");   // the limit of the scan records (engine.cpp)
    Hit h{};
    h.kind = hh.kind; h.prefix_len = hh.prefix_len; h.a = hh.a; h.ids_off = 0; h.n_globs = (uint16_t)hh.globs.size();
    h.start = 0; h.len_type = c.len_type;
    so.hits.push_back(h);
    so.ids = std::move(hh.globs);
}

void matchy_query_into(const matchy_t* dbc, const char* query, matchy_result_t* result) {
    if (!result) return;
    *result = matchy_result_t{false, 0, nullptr, nullptr};
    if (!dbc || !query) return;
    Db* db = const_cast<Db*>(reinterpret_cast<const Db*>(dbc));
    size_t qn = strlen(query);
    if (!valid_utf8_host((const uint8_t*)query, qn)) return;  // CStr::to_str failure -> found=false
    try {
        const std::string key(query, qn);
        Db::Lru& cache = db->cache();
        IpAddr ip;
        bool is_ip;
        std::string text;
        Candidate c;
        if (const Db::Cached* hit = db->cache_cap ? cache.get(key) : nullptr) {
            // cache hit (database.rs:727-757): same accounting as a lookup, except that a cached miss is typed by parsing the query
            db->st_total++; db->st_chit++;
            if (hit->kind == 0) { if (parse_ip(query, qn, ip)) db->st_ip++; else db->st_str++; db->st_nomatch++; return; }
            if (hit->kind == 2) db->st_ip++; else db->st_str++;
            db->st_match++;
            if (!hit->has_data) return;
            result->found = true; result->prefix_len = hit->prefix_len;
            result->_data_cache = new DataValue(hit->data);
            result->_db_ref = dbc;
            return;
        }
        if (!query_candidate(query, qn, ip, is_ip, text, c)) return;
        ScanOutput so;
        single_lookup(db, text, c, ip, is_ip, so);
        db->st_total++;
        if (db->cache_cap) db->st_cmiss++;
        if (so.hits.empty()) {   // a miss counts as a string query (database.rs:786-790)
            db->st_str++; db->st_nomatch++;
            cache.put(key, Db::Cached{0, 0, false, DataValue()}, db->cache_cap);
            return;
        }
        const Hit& h = so.hits[0];
        if (h.kind == 2) db->st_ip++; else db->st_str++;
        db->st_match++;
        DataValue* dv = new DataValue();
        bool ok = false;
        if (h.kind == 2) { ok = db->img->decode_data(h.a, *dv); result->prefix_len = h.prefix_len; }
        else {
            // first pattern's data only (c_api/matchy.rs:1143-1154)
            uint32_t off;
            bool have = false;
            if (h.a != 0xFFFFFFFFu && db->img->lit_data_offset(h.a, off)) have = true;
            else if ((h.a == 0xFFFFFFFFu || !db->img->lit_data_offset(h.a, off)) && h.n_globs > 0) have = db->img->glob_data_offset(so.ids[h.ids_off], off);
            ok = have && db->img->decode_data(off, *dv);
        }
        if (!ok) { delete dv; result->prefix_len = 0; cache.put(key, Db::Cached{h.kind, 0, false, DataValue()}, db->cache_cap); return; }
        result->found = true;
        result->_data_cache = dv;
        result->_db_ref = dbc;
        cache.put(key, Db::Cached{h.kind, result->prefix_len, true, *dv}, db->cache_cap);
    } catch (const HipError& e) { set_error(e.what); }
    catch (const std::exception& e) { set_error(e.what()); }
}
// The answer `matchy query DB QUERY` prints (bin/commands/query_cmd.rs:8-69), as compact JSON: an array with one object per
// pattern that carries data (literal first, then globs by id), or one object for an IP hit (its data plus "cidr" and
// "prefix_len"), or [] when nothing matches. *found follows the command's exit status rule (:19-21).
char* matchy_amd_query_json(const matchy_t* dbc, const char* query, int32_t* found) {
    if (found) *found = 0;
    if (!dbc || !query) return nullptr;
    Db* db = const_cast<Db*>(reinterpret_cast<const Db*>(dbc));
    const size_t qn = strlen(query);
    if (!valid_utf8_host((const uint8_t*)query, qn)) return strdup("[]");
    try {
        IpAddr ip;
        bool is_ip;
        std::string text;
        Candidate c;
        if (!query_candidate(query, qn, ip, is_ip, text, c)) return strdup("[]");
        ScanOutput so;
        single_lookup(db, text, c, ip, is_ip, so);
        if (so.hits.empty()) return strdup("[]");
        const Hit& h = so.hits[0];
        if (found) *found = 1;
        std::string o = "[";
        if (h.kind == 2) {
            DataValue dv;
            if (!db->img->decode_data(h.a, dv)) { set_error("query: cannot decode the data record"); return nullptr; }
            if (dv.type == DataValue::MAP) {
                dv.map["cidr"] = DataValue::String(format_cidr(ip, h.prefix_len));
                dv.map["prefix_len"] = DataValue::Uint16(h.prefix_len);
            }
            to_json(dv, o);
        } else {
            bool any = false;
            auto push = [&](uint32_t off) {
                DataValue dv;
                if (!db->img->decode_data(off, dv)) return;
                if (any) o.push_back(',');
                to_json(dv, o);
                any = true;
            };
            uint32_t off;
            if (h.a != 0xFFFFFFFFu && db->img->lit_data_offset(h.a, off)) push(off);
            for (uint32_t k = 0; k < h.n_globs; ++k) if (db->img->glob_data_offset(so.ids[h.ids_off + k], off)) push(off);
        }
        o.push_back(']');
        return strdup(o.c_str());
    } catch (const HipError& e) { set_error(e.what); }
    catch (const std::exception& e) { set_error(e.what()); }
    return nullptr;
}
matchy_result_t matchy_query(const matchy_t* db, const char* query) {
    matchy_result_t r;
    matchy_query_into(db, query, &r);
    return r;
}
void matchy_free_result(matchy_result_t* r) {
    if (!r) return;
    delete reinterpret_cast<DataValue*>(r->_data_cache);
    r->_data_cache = nullptr;
}
char* matchy_result_to_json(const matchy_result_t* r) {
    if (!r || !r->found || !r->_data_cache) return nullptr;
    std::string s;
    to_json(*reinterpret_cast<const DataValue*>(r->_data_cache), s);
    return strdup(s.c_str());
}

// ------------------------------------------------------------------------------------------------ structured data
namespace {
matchy_entry_data_t entry_empty() {
    matchy_entry_data_t e;
    memset(&e, 0, sizeof(e));
    return e;
}
// matchy_entry_data_t::from_data_value (c_api/matchy.rs:1599-1690)
matchy_entry_data_t entry_from(const DataValue& v) {
    matchy_entry_data_t e = entry_empty();
    e.has_data = true;
    e.type_ = (uint32_t)v.type;
    switch (v.type) {
        case DataValue::POINTER: e.value.pointer = (uint32_t)v.u; break;
        case DataValue::STRING: e.value.utf8_string = v.str.c_str(); e.data_size = (uint32_t)v.str.size(); break;
        case DataValue::DOUBLE: e.value.double_value = v.f64; e.data_size = 8; break;
        case DataValue::BYTES: e.value.bytes = (const uint8_t*)v.str.data(); e.data_size = (uint32_t)v.str.size(); break;
        case DataValue::UINT16: e.value.uint16 = (uint16_t)v.u; e.data_size = 2; break;
        case DataValue::UINT32: e.value.uint32 = (uint32_t)v.u; e.data_size = 4; break;
        case DataValue::MAP: e.data_size = (uint32_t)v.map.size(); break;
        case DataValue::INT32: e.value.int32 = v.i32; e.data_size = 4; break;
        case DataValue::UINT64: e.value.uint64 = v.u; e.data_size = 8; break;
        case DataValue::UINT128:
            for (int i = 0; i < 8; ++i) { e.value.uint128[i] = (uint8_t)(v.uhi >> (56 - 8 * i)); e.value.uint128[8 + i] = (uint8_t)(v.u >> (56 - 8 * i)); }
            e.data_size = 16;
            break;
        case DataValue::ARRAY: e.data_size = (uint32_t)v.arr.size(); break;
        case DataValue::BOOL: e.value.boolean = v.u != 0; e.data_size = 1; break;
        case DataValue::FLOAT: e.value.float_value = v.f32; e.data_size = 4; break;
    }
    return e;
}
const DataValue* entry_root(const matchy_entry_s* entry) {
    if (!entry || !entry->data_ptr) return nullptr;
    const matchy_result_t* r = reinterpret_cast<const matchy_result_t*>(entry->data_ptr);
    return reinterpret_cast<const DataValue*>(r->_data_cache);
}
void flatten(const DataValue& v, matchy_entry_data_list_t**& tail) {
    auto* node = new matchy_entry_data_list_t{entry_from(v), nullptr};
    *tail = node;
    tail = &node->next;
    if (v.type == DataValue::MAP) for (const auto& kv : v.map) flatten(kv.second, tail);
    else if (v.type == DataValue::ARRAY) for (const DataValue& c : v.arr) flatten(c, tail);
}
}  // namespace

int32_t matchy_result_get_entry(const matchy_result_t* result, matchy_entry_s* entry) {
    if (!result || !entry) return MATCHY_ERROR_INVALID_PARAM;
    if (!result->found) return MATCHY_ERROR_NO_DATA;
    entry->db = result->_db_ref;
    entry->data_ptr = result;
    return MATCHY_SUCCESS;
}
int32_t matchy_aget_value(const matchy_entry_s* entry, matchy_entry_data_t* out, const char* const* path) {
    if (!entry || !out || !path) return MATCHY_ERROR_INVALID_PARAM;
    for (size_t i = 0; path[i]; ++i) if (!valid_utf8_host((const uint8_t*)path[i], strlen(path[i]))) return MATCHY_ERROR_INVALID_PARAM;
    *out = entry_empty();
    const DataValue* v = entry_root(entry);
    if (!v) return MATCHY_ERROR_NO_DATA;
    for (size_t i = 0; path[i]; ++i) {   // navigate_path (c_api/matchy.rs:1692-1710)
        if (v->type == DataValue::MAP) {
            auto it = v->map.find(path[i]);
            if (it == v->map.end()) return MATCHY_ERROR_LOOKUP_PATH_INVALID;
            v = &it->second;
        } else if (v->type == DataValue::ARRAY) {
            const char* s = path[i];
            if (*s == '+') ++s;   // usize::from_str accepts a leading '+'
            if (!*s) return MATCHY_ERROR_LOOKUP_PATH_INVALID;
            size_t idx = 0;
            for (; *s; ++s) {
                if (*s < '0' || *s > '9' || idx > ((size_t)1 << 56)) return MATCHY_ERROR_LOOKUP_PATH_INVALID;
                idx = idx * 10 + (size_t)(*s - '0');
            }
            if (idx >= v->arr.size()) return MATCHY_ERROR_LOOKUP_PATH_INVALID;
            v = &v->arr[idx];
        } else {
            return MATCHY_ERROR_LOOKUP_PATH_INVALID;
        }
    }
    *out = entry_from(*v);
    return MATCHY_SUCCESS;
}
int32_t matchy_get_entry_data_list(const matchy_entry_s* entry, matchy_entry_data_list_t** list) {
    if (!entry || !list) return MATCHY_ERROR_INVALID_PARAM;
    const DataValue* v = entry_root(entry);
    if (!v) return MATCHY_ERROR_NO_DATA;
    *list = nullptr;
    matchy_entry_data_list_t** tail = list;
    flatten(*v, tail);
    return MATCHY_SUCCESS;
}
void matchy_free_entry_data_list(matchy_entry_data_list_t* list) {
    while (list) { matchy_entry_data_list_t* n = list->next; delete list; list = n; }
}

int32_t matchy_validate(const char* filename, int32_t level, char** error_message) {
    if (error_message) *error_message = nullptr;
    if (!filename || !valid_utf8_host((const uint8_t*)filename, strlen(filename))) return MATCHY_ERROR_INVALID_PARAM;
    if (level != MATCHY_VALIDATION_STANDARD && level != MATCHY_VALIDATION_STRICT) return MATCHY_ERROR_INVALID_PARAM;
    std::vector<uint8_t> bytes;
    if (!read_file(filename, bytes)) {
        if (error_message) *error_message = strdup("Failed to validate database");
        return MATCHY_ERROR_IO;
    }
    DbImage img;
    std::string err;
    if (!img.open(std::move(bytes), err)) {
        if (error_message) *error_message = strdup(err.empty() ? "Validation failed (no error details)" : err.c_str());
        return MATCHY_ERROR_CORRUPT_DATA;
    }
    return MATCHY_SUCCESS;
}
int32_t matchy_builder_set_schema(matchy_builder_t* b, const char* name) {
    if (!b || !name) return MATCHY_ERROR_INVALID_PARAM;
    return MATCHY_ERROR_UNKNOWN_SCHEMA;
}

// ------------------------------------------------------------------------------------------------ extractor
matchy_extractor_t* matchy_extractor_create(uint32_t flags) { return matchy_amd_extractor_create(flags, 2); }
// ExtractorBuilder::min_domain_labels (matchy-extractor/src/lib.rs:101-104; `matchy extract --min-labels`)
matchy_extractor_t* matchy_amd_extractor_create(uint32_t flags, uint32_t min_domain_labels) {
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("matchy_amd: no HIP device available"); return nullptr; }
        auto e = std::make_unique<ExtractorH>();
        e->flags = flags;
        e->img = std::make_shared<DbImage>();  // no sections: has_ip/has_literal/has_glob all false
        e->img->node_count = 0;
        e->ddb = std::make_shared<DeviceDb>();
        int dev0 = 0;   // like matchy_open: one process per GPU selects its device with MATCHY_AMD_DEVICE
        if (const char* ev = getenv("MATCHY_AMD_DEVICE")) dev0 = atoi(ev);
        if (dev0 < 0 || dev0 >= ndev) { set_error("MATCHY_AMD_DEVICE out of range"); return nullptr; }
        e->ddb->upload(*e->img, dev0);
        e->scanner = std::make_unique<Scanner>(e->img, e->ddb, flags, min_domain_labels ? min_domain_labels : 2);
        return reinterpret_cast<matchy_extractor_t*>(e.release());
    } catch (const HipError& e) { set_error(e.what); return nullptr; }
    catch (const std::exception& e) { set_error(e.what()); return nullptr; }
}
void matchy_extractor_free(matchy_extractor_t* e) { delete reinterpret_cast<ExtractorH*>(e); }
const char* matchy_item_type_name(uint8_t t) { return item_type_name(t); }

int32_t matchy_extractor_extract_chunk(const matchy_extractor_t* ec, const uint8_t* data, uintptr_t len, matchy_matches_t* out) {
    if (!ec || !out || (!data && len)) return MATCHY_ERROR_INVALID_PARAM;
    ExtractorH* e = const_cast<ExtractorH*>(reinterpret_cast<const ExtractorH*>(ec));
    try {
        std::lock_guard<std::mutex> lk(e->mu);
        ScanOutput so;
        std::vector<uint64_t> bases;
        e->scanner->scan_host(data, len, false, true, so, &bases, nullptr, nullptr, nullptr);
        size_t nc = so.cands.size();
        std::vector<uint32_t> order(nc);
        for (uint32_t i = 0; i < nc; ++i) order[i] = i;
        auto abs_start = [&](uint32_t i) { return (uint64_t)so.cands[i].start + bases[i]; };
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            int ra = type_rank(so.cands[a].len_type >> 24), rb = type_rank(so.cands[b].len_type >> 24);
            if (ra != rb) return ra < rb;
            uint64_t sa = abs_start(a), sb = abs_start(b);
            if (sa != sb) return sa < sb;
            return (so.cands[a].len_type >> 24) < (so.cands[b].len_type >> 24);
        });
        auto* in = new MatchesInternal();
        in->values.reserve(nc);
        in->items.reserve(nc);
        for (uint32_t oi : order) {
            const Candidate& c = so.cands[oi];
            uint64_t s = abs_start(oi), n = c.len_type & 0xFFFFFF;
            uint8_t ty = (uint8_t)(c.len_type >> 24);
            std::string val;
            if (ty == IT_IPV4) {  // as_value(): canonical Display (matchy-extractor/src/lib.rs:300-311)
                uint8_t b[4] = {(uint8_t)(c.v4 >> 24), (uint8_t)(c.v4 >> 16), (uint8_t)(c.v4 >> 8), (uint8_t)c.v4};
                val = format_ipv4(b);
            } else if (ty == IT_IPV6) {
                uint8_t b[16];
                if (parse_ipv6((const char*)data + s, n, b)) val = format_ipv6(b);
            } else val.assign((const char*)data + s, n);
            in->values.push_back(std::move(val));
            in->items.push_back(matchy_match_t{ty, nullptr, (uintptr_t)s, (uintptr_t)(s + n)});
        }
        for (size_t i = 0; i < in->items.size(); ++i) in->items[i].value = in->values[i].c_str();
        out->items = in->items.data(); out->count = in->items.size(); out->_internal = in;
        return MATCHY_SUCCESS;
    } catch (const HipError& ex) { set_error(ex.what); return MATCHY_ERROR_IO; }
    catch (const std::exception& ex) { set_error(ex.what()); return MATCHY_ERROR_IO; }
}
void matchy_matches_free(matchy_matches_t* m) {
    if (!m) return;
    delete reinterpret_cast<MatchesInternal*>(m->_internal);
    m->items = nullptr; m->count = 0; m->_internal = nullptr;
}

// ------------------------------------------------------------------------------------------------ bulk scan
matchy_scanner_t* matchy_scanner_create(const matchy_t* dbc, uint32_t extract_flags, int32_t device) {
    if (!dbc || device < 0) return nullptr;
    Db* db = const_cast<Db*>(reinterpret_cast<const Db*>(dbc));
    try {
        if (extract_flags == 0) {  // match_cmd.rs:276-303
            if (db->img->has_ip) extract_flags |= EX_IPV4 | EX_IPV6;
            if (db->img->has_literal || db->img->has_glob) extract_flags |= EX_DOMAINS | EX_EMAILS | EX_HASHES | EX_BITCOIN | EX_ETHEREUM | EX_MONERO;
        }
        std::shared_ptr<DeviceDb> ddb;
        { std::lock_guard<std::mutex> lk(db->mu); ddb = db->device_db(device); }
        auto s = std::make_unique<ScannerH>();
        s->db = db;
        s->sc = std::make_unique<Scanner>(db->img, ddb, extract_flags, 2);
        return reinterpret_cast<matchy_scanner_t*>(s.release());
    } catch (const HipError& e) { set_error(e.what); return nullptr; }
    catch (const std::exception& e) { set_error(e.what()); return nullptr; }
}
void matchy_scanner_free(matchy_scanner_t* s) { delete reinterpret_cast<ScannerH*>(s); }
void matchy_scanner_set_profile(matchy_scanner_t* s, bool on) { if (s) reinterpret_cast<ScannerH*>(s)->sc->set_profile(on); }
void matchy_scanner_set_slices(matchy_scanner_t* s, int32_t n) { if (s) reinterpret_cast<ScannerH*>(s)->sc->set_slices(n); }
int32_t matchy_scanner_last_slices(const matchy_scanner_t* s) { return s ? reinterpret_cast<const ScannerH*>(s)->sc->last_slice_count() : 0; }
void matchy_scanner_get_timing(const matchy_scanner_t* s, float out[5]) {
    if (!s || !out) return;
    const ScanTiming& t = reinterpret_cast<const ScannerH*>(s)->sc->timing();
    out[0] = t.anchor_ms; out[1] = t.validate_ms; out[2] = t.rare_ms; out[3] = t.lookup_ms; out[4] = t.total_ms;
}

int32_t matchy_scanner_scan(matchy_scanner_t* s, const uint8_t* data, size_t len, matchy_scan_result_t* out) {
    if (!s || !out || (!data && len) || len > 0xFFFFFFFFull) return MATCHY_ERROR_INVALID_PARAM;
    ScannerH* h = reinterpret_cast<ScannerH*>(s);
    try {
        ScanOutput so;
        std::vector<FinalHit> fin;
        std::vector<uint32_t> fids;
        std::vector<long long> foffs;
        h->sc->scan_host(data, len, true, false, so, nullptr, &fin, &fids, &foffs);
        fill_result(fin.data(), fin.size(), fids.data(), foffs.data(), fids.size(), so.lines, so.n_cand, len, false, true, out);
        return MATCHY_SUCCESS;
    } catch (const HipError& e) { set_error(e.what); return MATCHY_ERROR_IO; }
    catch (const std::exception& e) { set_error(e.what()); return MATCHY_ERROR_IO; }
}

int32_t matchy_scanner_scan_device(matchy_scanner_t* s, const void* dptr, size_t len, void* stream, uint32_t fetch_mode, matchy_scan_result_t* out) {
    if (!s || !out || !dptr) return MATCHY_ERROR_INVALID_PARAM;
    if (len >= 0x7FFF0000ull) return MATCHY_ERROR_INVALID_PARAM;
    ScannerH* h = reinterpret_cast<ScannerH*>(s);
    try {
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        const bool sorted = (fetch_mode & 2) != 0;
        const bool compact = compact_mode(fetch_mode);
        h->sc->scan_device(reinterpret_cast<const uint8_t*>(dptr), (uint32_t)len, true, st, (fetch_mode & 1) && !sorted, /*fork=*/true, h->sc->slices(), compact);
        ScanOutput so;
        h->sc->fetch(so, false, st, (fetch_mode & 1) ? HITS_FINAL : HITS_NONE, sorted);
        if ((fetch_mode & 7u) == MATCHY_SCAN_FETCH_DEVICE) { fill_device_result(*h->sc, so, len, out); return MATCHY_SUCCESS; }
        fill_result(so.fin, so.n_fin, so.fin_ids, so.fin_offs, so.n_fin_ids, so.lines, so.n_cand, len, !sorted, sorted, out);
        out->ip4_hits = reinterpret_cast<const matchy_scan_ip4_hit_t*>(so.c4); out->n_ip4_hits = so.n_c4;
        if (!(fetch_mode & 1)) out->n_hits = so.n_hits;  // count only; `hits` stays NULL
        return MATCHY_SUCCESS;
    } catch (const HipError& e) { set_error(e.what); return MATCHY_ERROR_IO; }
    catch (const std::exception& e) { set_error(e.what()); return MATCHY_ERROR_IO; }
}

// The two halves of matchy_scanner_scan_device: submit launches the kernels of one batch on `stream` and returns; wait
// blocks until that batch is done and hands out its result. A host that keeps two scanners busy (each on its own stream)
// overlaps one batch's result transfer and latency-bound kernels with the next batch's streaming kernel.
int32_t matchy_scanner_submit_device(matchy_scanner_t* s, const void* dptr, size_t len, void* stream, uint32_t fetch_mode) {
    if (!s || !dptr) return MATCHY_ERROR_INVALID_PARAM;
    if (len >= 0x7FFF0000ull) return MATCHY_ERROR_INVALID_PARAM;
    ScannerH* h = reinterpret_cast<ScannerH*>(s);
    try {
        const bool sorted = (fetch_mode & 2) != 0;
        h->sc->scan_device(reinterpret_cast<const uint8_t*>(dptr), (uint32_t)len, true, reinterpret_cast<hipStream_t>(stream), (fetch_mode & 1) && !sorted, /*fork=*/false, 0,
                           compact_mode(fetch_mode));
        h->pending = true; h->pending_len = len; h->pending_mode = fetch_mode; h->pending_stream = stream;
        return MATCHY_SUCCESS;
    } catch (const HipError& e) { set_error(e.what); return MATCHY_ERROR_IO; }
    catch (const std::exception& e) { set_error(e.what()); return MATCHY_ERROR_IO; }
}
int32_t matchy_scanner_wait(matchy_scanner_t* s, matchy_scan_result_t* out) {
    if (!s || !out) return MATCHY_ERROR_INVALID_PARAM;
    ScannerH* h = reinterpret_cast<ScannerH*>(s);
    if (!h->pending) { set_error("matchy_scanner_wait: nothing was submitted"); return MATCHY_ERROR_INVALID_PARAM; }
    h->pending = false;
    try {
        const uint32_t fetch_mode = h->pending_mode;
        const bool sorted = (fetch_mode & 2) != 0;
        hipStream_t st = reinterpret_cast<hipStream_t>(h->pending_stream);
        ScanOutput so;
        h->sc->fetch(so, false, st, (fetch_mode & 1) ? HITS_FINAL : HITS_NONE, sorted);
        if ((fetch_mode & 7u) == MATCHY_SCAN_FETCH_DEVICE) { fill_device_result(*h->sc, so, h->pending_len, out); return MATCHY_SUCCESS; }
        fill_result(so.fin, so.n_fin, so.fin_ids, so.fin_offs, so.n_fin_ids, so.lines, so.n_cand, h->pending_len, !sorted, sorted, out);
        out->ip4_hits = reinterpret_cast<const matchy_scan_ip4_hit_t*>(so.c4); out->n_ip4_hits = so.n_c4;
        if (!(fetch_mode & 1)) out->n_hits = so.n_hits;
        return MATCHY_SUCCESS;
    } catch (const HipError& e) { set_error(e.what); return MATCHY_ERROR_IO; }
    catch (const std::exception& e) { set_error(e.what()); return MATCHY_ERROR_IO; }
}

void matchy_scan_result_free(matchy_scan_result_t* r) {
    if (!r) return;
    delete reinterpret_cast<ScanResultInternal*>(r->_internal);
    memset(r, 0, sizeof(*r));
}

bool matchy_scan_result_on_device(const matchy_scan_result_t* r) {
    return r && r->_internal && reinterpret_cast<const ScanResultInternal*>(r->_internal)->on_device;
}

char* matchy_scan_hit_to_json(const matchy_scanner_t* s, const matchy_scan_result_t* r, size_t i, const uint8_t* text, const char* source) {
    // i counts through hits, then through the compact IPv4 records of a MATCHY_SCAN_FETCH_COMPACT result
    if (!s || !r || !text || i >= r->n_hits + r->n_ip4_hits) return nullptr;
    if (i < r->n_hits ? !r->hits : !r->ip4_hits) return nullptr;
    if (matchy_scan_result_on_device(r)) { set_error("matchy_scan_hit_to_json: the records of this result are in device memory (MATCHY_SCAN_FETCH_DEVICE)"); return nullptr; }
    const DbImage& img = reinterpret_cast<const ScannerH*>(s)->sc->image();
    const matchy_scan_hit_t h = i < r->n_hits ? r->hits[i] : matchy_scan_ip4_hit_expand(r->ip4_hits[i - r->n_hits]);
    const uint32_t hlen = MATCHY_SCAN_HIT_LEN(h);
    std::string matched((const char*)text + h.start, hlen), o = "{";
    if (h.kind == 2) {
        IpAddr ip;
        std::string cidr;
        // format_cidr_into parses matched_text again (cli_utils.rs:113); it always parses for extracted IPs
        if (parse_ip(matched.data(), matched.size(), ip)) cidr = format_cidr(ip, h.prefix_len);
        else cidr = matched + "/" + std::to_string((unsigned)h.prefix_len);
        o += "\"cidr\":"; json_escape(cidr, o);
        o += ",\"data\":";
        DataValue dv;
        if (img.decode_data(h.value, dv)) to_json(dv, o); else o += "null";
        o += ",\"match_type\":\"ip\",\"matched_text\":"; json_escape(matched, o);
        o += ",\"prefix_len\":" + std::to_string((unsigned)h.prefix_len);
    } else {
        std::string arr;
        bool any = false;
        for (uint32_t k = 0; k < h.n_ids; ++k) {
            int64_t off = r->data_offsets[h.value + k];
            if (off < 0) continue;
            if (any) arr.push_back(',');
            DataValue dv;
            if (img.decode_data((uint32_t)off, dv)) to_json(dv, arr); else arr += "null";
            any = true;
        }
        if (any) o += "\"data\":[" + arr + "],";
        o += "\"match_type\":\"pattern\",\"matched_text\":"; json_escape(matched, o);
        o += ",\"pattern_count\":" + std::to_string(h.n_ids);
    }
    o += ",\"source\":"; json_escape(source ? source : "-", o);
    o += ",\"timestamp\":\"0.000\"}";
    return strdup(o.c_str());
}


// Every match of a result as NDJSON, one line per hit in the order of the arrays (hits, then compact IPv4 records) — the text
// matchy_scan_hit_to_json returns for each, concatenated with '\n' behind every line — in ONE call and one allocation: the default output
// of `matchy match` (match_processor/parallel.rs:297-369). Per scanner the rendered data payloads are cached by data offset and the
// fixed parts of a line are appended as literals, so a line costs a few memcpy instead of a decode of the MMDB value, a dozen
// std::string temporaries and a strdup: the renderer was what bounded the command line with --format json (~6 GB/s of log against
// 40+ with --format summary).
int32_t matchy_scan_result_to_ndjson(matchy_scanner_t* s, const matchy_scan_result_t* r, const uint8_t* text, const char* source, char** out, size_t* out_len) {
    if (!s || !r || !out || !out_len || (!text && (r->n_hits || r->n_ip4_hits))) return MATCHY_ERROR_INVALID_PARAM;
    if (matchy_scan_result_on_device(r)) { set_error("matchy_scan_result_to_ndjson: the records of this result are in device memory (MATCHY_SCAN_FETCH_DEVICE)"); return MATCHY_ERROR_INVALID_PARAM; }
    ScannerH* sh = reinterpret_cast<ScannerH*>(s);
    const DbImage& img = sh->sc->image();
    const size_t n = (r->hits ? r->n_hits : 0) + (r->ip4_hits ? r->n_ip4_hits : 0);
    if (sh->json_source_of != (source ? source : "-") || sh->json_source.empty()) {
        sh->json_source_of = source ? source : "-";
        sh->json_source.clear();
        json_escape(sh->json_source_of, sh->json_source);
    }
    // One piece of the result per thread (large results only: a 256 MiB batch of a web-server log carries ~70 K matches, ~15 MB of text; one thread
    // renders ~9 M lines a second, which is less than two scanners on one GPU deliver). While pieces are rendered side by side the cache of
    // payload texts is only read: a payload that is not in it yet is decoded into the piece's own list and joins the cache behind the threads.
    struct Piece {
        std::string o;
        std::deque<std::pair<uint32_t, std::string>> fresh;          // payloads decoded by this piece (stable addresses)
        std::unordered_map<uint32_t, const std::string*> fresh_at;
    };
    auto render = [&](size_t i0, size_t i1, Piece& pc, bool alone) {
        std::string& o = pc.o;
        o.reserve((i1 - i0) * 192 + 64);
        auto data_json = [&](uint32_t off) -> const std::string& {
            auto it = sh->json_of_data.find(off);
            if (it != sh->json_of_data.end()) return it->second;
            if (alone) {
                if (sh->json_of_data.size() > (1u << 20)) sh->json_of_data.clear();   // a database with millions of distinct payloads: bounded
                std::string js;
                DataValue dv;
                if (img.decode_data(off, dv)) to_json(dv, js); else js = "null";
                return sh->json_of_data.emplace(off, std::move(js)).first->second;
            }
            auto f = pc.fresh_at.find(off);
            if (f != pc.fresh_at.end()) return *f->second;
            std::string js;
            DataValue dv;
            if (img.decode_data(off, dv)) to_json(dv, js); else js = "null";
            pc.fresh.emplace_back(off, std::move(js));
            pc.fresh_at.emplace(off, &pc.fresh.back().second);
            return pc.fresh.back().second;
        };
        char num[16];
        auto append_uint = [&](unsigned v) { const int k = snprintf(num, sizeof(num), "%u", v); o.append(num, (size_t)k); };
        // does the text need more than quotes around it? (extracted items are almost always plain ASCII without '"' or a backslash)
        auto append_quoted = [&](const char* p, size_t len) {
            bool plain = true;
            for (size_t k = 0; k < len; ++k) { const unsigned char c = (unsigned char)p[k]; if (c < 0x20 || c == '"' || c == 0x5C || c >= 0x7F) { plain = false; break; } }
            if (plain) { o.push_back('"'); o.append(p, len); o.push_back('"'); }
            else json_escape(std::string(p, len), o);
        };
        for (size_t i = i0; i < i1; ++i) {
            const bool in_hits = r->hits && i < r->n_hits;
            const matchy_scan_hit_t h = in_hits ? r->hits[i] : matchy_scan_ip4_hit_expand(r->ip4_hits[i - (r->hits ? r->n_hits : 0)]);
            const uint32_t hlen = MATCHY_SCAN_HIT_LEN(h);
            const char* mt = (const char*)text + h.start;
            if (h.kind == 2) {
                IpAddr ip;
                o += "{\"cidr\":";
                // format_cidr_into parses matched_text again (cli_utils.rs:113); it always parses for extracted IPs
                if (parse_ip(mt, hlen, ip)) { const std::string c = format_cidr(ip, h.prefix_len); o.push_back('"'); o += c; o.push_back('"'); }
                else { std::string c(mt, hlen); c += "/"; c += std::to_string((unsigned)h.prefix_len); json_escape(c, o); }
                o += ",\"data\":";
                o += data_json(h.value);
                o += ",\"match_type\":\"ip\",\"matched_text\":";
                append_quoted(mt, hlen);
                o += ",\"prefix_len\":";
                append_uint((unsigned)h.prefix_len);
            } else {
                o.push_back('{');
                bool any = false;
                for (uint32_t k = 0; k < h.n_ids; ++k) {
                    const int64_t off = r->data_offsets[h.value + k];
                    if (off < 0) continue;
                    o += any ? "," : "\"data\":[";
                    o += data_json((uint32_t)off);
                    any = true;
                }
                if (any) o += "],";
                o += "\"match_type\":\"pattern\",\"matched_text\":";
                append_quoted(mt, hlen);
                o += ",\"pattern_count\":";
                append_uint((unsigned)h.n_ids);
            }
            o += ",\"source\":";
            o += sh->json_source;
            o += ",\"timestamp\":\"0.000\"}\n";
        }
    };
    static const unsigned max_threads = [] {
        if (const char* e = getenv("MATCHY_AMD_JSON_THREADS")) return (unsigned)std::max(1, atoi(e));
        const unsigned hw = std::thread::hardware_concurrency();
        return hw >= 8 ? 4u : hw >= 4 ? 2u : 1u;
    }();
    const unsigned nt = (unsigned)std::min<size_t>(max_threads, std::max<size_t>(1, n / 16384));
    std::vector<Piece> pieces(nt);
    try {
        if (nt == 1) render(0, n, pieces[0], true);
        else {
            // An exception must not leave a render thread (std::terminate), and none may leave this scope while a thread is still
            // joinable: every piece catches its own, the joiner runs on every way out, the first failure is rethrown afterwards.
            std::atomic<bool> failed{false};
            std::vector<std::thread> th;
            struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } joiner{th};
            auto guarded = [&](size_t i0, size_t i1, Piece& pc) {
                try { render(i0, i1, pc, false); } catch (...) { failed.store(true); }
            };
            th.reserve(nt);
            for (unsigned t = 1; t < nt; ++t) th.emplace_back([&, t] { guarded(n * t / nt, n * (t + 1) / nt, pieces[t]); });
            guarded(0, n / nt, pieces[0]);
            for (auto& x : th) x.join();
            if (failed.load()) throw std::bad_alloc();
            for (Piece& pc : pieces)
                for (auto& e : pc.fresh) {
                    if (sh->json_of_data.size() > (1u << 20)) sh->json_of_data.clear();
                    sh->json_of_data.emplace(e.first, std::move(e.second));
                }
        }
    } catch (const std::exception& e) { set_error(e.what()); return MATCHY_ERROR_OUT_OF_MEMORY; }
    size_t total = 0;
    for (const Piece& pc : pieces) total += pc.o.size();
    char* buf = (char*)malloc(total + 1);
    if (!buf) { set_error("matchy_scan_result_to_ndjson: out of memory"); return MATCHY_ERROR_OUT_OF_MEMORY; }
    {
        std::vector<std::thread> th;
        struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } joiner{th};
        size_t at = 0;
        for (unsigned t = 0; t < nt; ++t) {
            char* dst = buf + at;
            at += pieces[t].o.size();
            bool spawned = false;
            if (t + 1 < nt) {
                try { th.emplace_back([dst, &pieces, t] { memcpy(dst, pieces[t].o.data(), pieces[t].o.size()); }); spawned = true; }
                catch (...) {}   // no thread to be had: copy here
            }
            if (!spawned) memcpy(dst, pieces[t].o.data(), pieces[t].o.size());
        }
    }
    buf[total] = 0;
    *out = buf; *out_len = total;
    return MATCHY_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ multi-device scanner
// The reader -> per-device workers -> ordered gather of `matchy match` (reference: process_files_parallel,
// crates/matchy/src/processing/parallel.rs:494-505 and its workers :594-704) behind the C ABI: the host submits newline-aligned
// batches, one worker thread per device entry scans them with a scanner of its own (host-buffer entry: the batch is pinned for its copy,
// results come back in canonical order), and the host takes the results back IN SUBMISSION ORDER. No data-path collective: line blocks
// are independent (N4), the database is replicated per device, the counters are summed by the caller.
}  // extern "C"

namespace {
struct MultiJob { size_t seq; const uint8_t* data; size_t len; void* tag; const void* pinned; int32_t node; };   // node: NUMA node the bytes live on, -1 = anywhere
struct MultiDone { int32_t status = MATCHY_SUCCESS; matchy_scan_result_t res{}; const uint8_t* data = nullptr; size_t len = 0; void* tag = nullptr; void* payload = nullptr; size_t worker = 0; };
struct MultiScanner {
    const matchy_t* db = nullptr;
    uint32_t flags = 0;
    std::vector<int> devices;
    std::vector<matchy_scanner_t*> scanners;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done, cv_space;
    std::deque<MultiJob> q;
    std::map<size_t, MultiDone> done;
    size_t submitted = 0, taken = 0, max_q = 2;
    // END-TO-END back-pressure (the reference bounds its channels for the same reason, processing/parallel.rs:563-577 "prevent memory
    // explosion"): at most max_inflight batches exist between submit() and next() — queued, being scanned, or finished and not yet
    // taken. Without it a slow consumer of next() (a blocked stdout) lets the reader buffer the whole input and every result.
    size_t max_inflight = 4;
    bool closing = false;
    matchy_multi_batch_fn hook = nullptr;
    void* hook_user = nullptr;
    std::string first_error;   // of a worker (scanner creation, scan): reported through matchy_amd_last_error by next()

    std::vector<int32_t> worker_node, worker_cpus;   // NUMA node of each worker's GPU (-1 unknown), CPUs its thread was bound to (0 = unbound)

    void worker(size_t w) {
        // this thread faults its batches' pages in, pins them and queues their copies: on the NUMA node of its GPU (the binding is
        // taken against the process's affinity at load time, host_topology.cpp: whoever created this thread may have bound itself)
        static const bool no_bind = getenv("MATCHY_AMD_NO_NUMA_BIND") != nullptr;
        {
            const int32_t node = matchy_amd_device_numa_node(devices[w]);
            const int32_t cpus = no_bind ? 0 : matchy_amd_bind_thread_to_device(devices[w]);
            std::lock_guard<std::mutex> lk(mu);
            worker_node[w] = node; worker_cpus[w] = cpus;
        }
        for (;;) {
            MultiJob j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return closing || !q.empty(); });
                if (q.empty()) return;
                // a batch whose bytes live on this worker's node (or anywhere) first — the copy then stays off the socket link; a worker
                // with nothing of its own takes the oldest batch of another node rather than idling
                auto it = q.begin();
                for (auto k = q.begin(); k != q.end(); ++k) if (k->node < 0 || k->node == worker_node[w]) { it = k; break; }
                j = *it;
                q.erase(it);
                cv_space.notify_one();
            }
            MultiDone d;
            d.data = j.data; d.len = j.len; d.tag = j.tag; d.worker = w;
            if (!scanners[w]) scanners[w] = matchy_scanner_create(db, flags, devices[w]);
            std::string err;
            if (!scanners[w]) { d.status = MATCHY_ERROR_IO; err = std::string("multi scanner: no scanner on device ") + std::to_string(devices[w]) + ": " + matchy_amd_last_error(); }
            else if (j.len) {
                d.status = matchy_scanner_scan(scanners[w], j.data, j.len, &d.res);
                if (d.status != MATCHY_SUCCESS) err = matchy_amd_last_error();
            }
            if (d.status == MATCHY_SUCCESS && hook) d.payload = hook(hook_user, w, scanners[w], &d.res, j.data, j.len, j.tag);
            if (j.pinned) matchy_amd_host_unregister(j.pinned);   // behind the hook: the unpin of this batch then runs beside the next worker's copy, not in front of this one's per-hit work
            std::lock_guard<std::mutex> lk(mu);
            if (!err.empty() && first_error.empty()) first_error = err;
            done.emplace(j.seq, d);
            cv_done.notify_all();
        }
    }
};
}  // namespace

extern "C" {

int32_t matchy_amd_device_numa_node(int32_t device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return mxy::numa_node_of_pci("/sys", bus);
}
int32_t matchy_amd_bind_thread_to_device(int32_t device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return mxy::bind_calling_thread(mxy::cpus_near_pci("/sys", bus));
}
int32_t matchy_amd_unbind_thread(void) { return mxy::unbind_calling_thread(); }
int32_t matchy_amd_numa_cpus(const char* sysfs_root, const char* pci_bus_id, int32_t* out, size_t cap) {
    if (!sysfs_root || !pci_bus_id) return -1;
    const std::vector<int> cpus = mxy::cpus_near_pci(sysfs_root, pci_bus_id);
    if (out) for (size_t i = 0; i < cpus.size() && i < cap; ++i) out[i] = cpus[i];
    return (int32_t)cpus.size();
}

matchy_multi_scanner_t* matchy_multi_scanner_create(const matchy_t* db, uint32_t extract_flags, const int32_t* devices, size_t n_devices) {
    if (!db) return nullptr;
    auto ms = std::make_unique<MultiScanner>();
    ms->db = db; ms->flags = extract_flags;
    if (!devices || !n_devices) ms->devices.push_back(reinterpret_cast<const Db*>(db)->default_device);
    else for (size_t i = 0; i < n_devices; ++i) { if (devices[i] < 0) { set_error("matchy_multi_scanner_create: negative device"); return nullptr; } ms->devices.push_back(devices[i]); }
    ms->scanners.assign(ms->devices.size(), nullptr);
    ms->worker_node.assign(ms->devices.size(), -1);
    ms->worker_cpus.assign(ms->devices.size(), 0);
    ms->max_q = ms->devices.size() + 1;
    ms->max_inflight = 2 * ms->devices.size() + 2;
    // the first scanner now, so that a database or device that cannot be used fails here; the others are created by their workers
    // when the first batch reaches them (a small input never pays for scanners it does not use)
    ms->scanners[0] = matchy_scanner_create(db, extract_flags, ms->devices[0]);
    if (!ms->scanners[0]) return nullptr;
    MultiScanner* raw = ms.get();
    for (size_t w = 0; w < ms->devices.size(); ++w) ms->workers.emplace_back([raw, w] { raw->worker(w); });
    return reinterpret_cast<matchy_multi_scanner_t*>(ms.release());
}
void matchy_multi_scanner_free(matchy_multi_scanner_t* h) {
    if (!h) return;
    MultiScanner* ms = reinterpret_cast<MultiScanner*>(h);
    { std::lock_guard<std::mutex> lk(ms->mu); ms->closing = true; ms->cv_work.notify_all(); }
    for (auto& t : ms->workers) t.join();
    for (auto& kv : ms->done) matchy_scan_result_free(&kv.second.res);   // results nobody took
    for (auto* sc : ms->scanners) if (sc) matchy_scanner_free(sc);
    delete ms;
}
size_t matchy_multi_scanner_workers(const matchy_multi_scanner_t* h) { return h ? reinterpret_cast<const MultiScanner*>(h)->devices.size() : 0; }
matchy_scanner_t* matchy_multi_scanner_worker_scanner(const matchy_multi_scanner_t* h, size_t worker) {
    const MultiScanner* ms = reinterpret_cast<const MultiScanner*>(h);
    return ms && worker < ms->scanners.size() ? ms->scanners[worker] : nullptr;
}
void matchy_multi_scanner_set_batch_hook(matchy_multi_scanner_t* h, matchy_multi_batch_fn fn, void* user) {
    if (!h) return;
    MultiScanner* ms = reinterpret_cast<MultiScanner*>(h);
    std::lock_guard<std::mutex> lk(ms->mu);
    ms->hook = fn; ms->hook_user = user;
}
int32_t matchy_multi_scanner_submit(matchy_multi_scanner_t* h, const uint8_t* data, size_t len, void* tag, const void* pinned_range) {
    return matchy_multi_scanner_submit_near(h, data, len, tag, pinned_range, -1);
}
int32_t matchy_multi_scanner_worker_numa(const matchy_multi_scanner_t* h, size_t worker, int32_t* node, int32_t* cpus_bound) {
    MultiScanner* ms = const_cast<MultiScanner*>(reinterpret_cast<const MultiScanner*>(h));
    if (!ms || worker >= ms->devices.size()) return MATCHY_ERROR_INVALID_PARAM;
    std::lock_guard<std::mutex> lk(ms->mu);
    if (node) *node = ms->worker_node[worker];
    if (cpus_bound) *cpus_bound = ms->worker_cpus[worker];
    return MATCHY_SUCCESS;
}
int32_t matchy_multi_scanner_submit_near(matchy_multi_scanner_t* h, const uint8_t* data, size_t len, void* tag, const void* pinned_range, int32_t numa_node) {
    if (!h || (!data && len) || len > 0xFFFFFFFFull) return MATCHY_ERROR_INVALID_PARAM;
    MultiScanner* ms = reinterpret_cast<MultiScanner*>(h);
    std::unique_lock<std::mutex> lk(ms->mu);
    // blocks while the job queue is full OR max_inflight batches are out (someone has to call matchy_multi_scanner_next: a caller that
    // submits and gathers on ONE thread interleaves the two — matchy_multi_scanner_pending() tells it when a next() is due)
    ms->cv_space.wait(lk, [&] { return ms->q.size() < ms->max_q && ms->submitted - ms->taken < ms->max_inflight; });
    ms->q.push_back(MultiJob{ms->submitted++, data, len, tag, pinned_range, numa_node});
    ms->cv_work.notify_one();
    return MATCHY_SUCCESS;
}
size_t matchy_multi_scanner_pending(const matchy_multi_scanner_t* h) {
    if (!h) return 0;
    MultiScanner* ms = const_cast<MultiScanner*>(reinterpret_cast<const MultiScanner*>(h));
    std::lock_guard<std::mutex> lk(ms->mu);
    return ms->submitted - ms->taken;
}
size_t matchy_multi_scanner_max_pending(const matchy_multi_scanner_t* h) { return h ? reinterpret_cast<const MultiScanner*>(h)->max_inflight : 0; }
int32_t matchy_multi_scanner_next(matchy_multi_scanner_t* h, matchy_multi_batch_t* out) {
    if (!h || !out) return MATCHY_ERROR_INVALID_PARAM;
    MultiScanner* ms = reinterpret_cast<MultiScanner*>(h);
    std::unique_lock<std::mutex> lk(ms->mu);
    if (ms->taken == ms->submitted) return 0;   // nothing pending
    ms->cv_done.wait(lk, [&] { return ms->done.count(ms->taken) != 0; });
    const MultiDone d = ms->done[ms->taken];
    ms->done.erase(ms->taken);
    out->seq = ms->taken++;
    ms->cv_space.notify_all();
    out->status = d.status; out->result = d.res; out->data = d.data; out->len = d.len; out->tag = d.tag; out->payload = d.payload; out->worker = d.worker;
    if (d.status != MATCHY_SUCCESS) set_error(ms->first_error.empty() ? "multi scanner: a batch failed" : ms->first_error);
    return 1;
}

// One buffer through all workers: cut at newlines into pieces (batch_bytes each; 0 = the buffer spread twice over the workers, at
// least 4 MiB and at most 256 MiB a piece), results merged into ONE result with offsets into `data` — what matchy_scanner_scan
// returns for the same bytes, whatever the device list.
int32_t matchy_multi_scanner_scan(matchy_multi_scanner_t* h, const uint8_t* data, size_t len, size_t batch_bytes, matchy_scan_result_t* out) {
    if (!h || !out || (!data && len) || len > 0xFFFFFFFFull) return MATCHY_ERROR_INVALID_PARAM;
    MultiScanner* ms = reinterpret_cast<MultiScanner*>(h);
    { std::lock_guard<std::mutex> lk(ms->mu); if (ms->taken != ms->submitted) { set_error("matchy_multi_scanner_scan: batches of an earlier submit are still pending"); return MATCHY_ERROR_INVALID_PARAM; } }
    if (!batch_bytes) {
        batch_bytes = (len + 2 * ms->devices.size() - 1) / (2 * ms->devices.size());
        batch_bytes = std::min<size_t>(std::max<size_t>(batch_bytes, (size_t)4 << 20), (size_t)256 << 20);
    }
    auto in = std::make_unique<ScanResultInternal>();
    uint64_t lines = 0, cands = 0;
    int32_t status = MATCHY_SUCCESS;
    std::string err;
    enum { READY = 0, ALL = 1, ONE = 2 };   // the finished ones / everything pending / exactly the next one (blocking)
    auto take_mode = [&](int mode) {
        for (;;) {
            { std::lock_guard<std::mutex> lk(ms->mu); if (ms->taken == ms->submitted) return; if (mode == READY && !ms->done.count(ms->taken)) return; }
            matchy_multi_batch_t b;
            if (matchy_multi_scanner_next(h, &b) != 1) return;
            if (b.status != MATCHY_SUCCESS) { if (status == MATCHY_SUCCESS) { status = b.status; err = matchy_amd_last_error(); } }
            else {
                const uint32_t base = (uint32_t)(b.data - data), id_shift = (uint32_t)in->ids.size();
                for (size_t i = 0; i < b.result.n_hits; ++i) {
                    matchy_scan_hit_t x = b.result.hits[i];
                    x.start += base;
                    if (x.kind == 3) x.value += id_shift;
                    in->hits.push_back(x);
                }
                in->ids.insert(in->ids.end(), b.result.pattern_ids, b.result.pattern_ids + b.result.n_ids);
                in->offs.insert(in->offs.end(), b.result.data_offsets, b.result.data_offsets + b.result.n_ids);
                lines += b.result.lines; cands += b.result.candidates;
            }
            matchy_scan_result_free(&b.result);
            if (mode == ONE) return;
        }
    };
    auto take = [&](bool all) { take_mode(all ? ALL : READY); };
    auto take_one = [&] { take_mode(ONE); };
    for (size_t pos = 0; pos < len;) {
        size_t end = std::min(len, pos + batch_bytes);
        if (end < len) {   // newline-aligned cut; a line longer than the piece extends it to that line's end
            const void* nl = memrchr(data + pos, '\n', end - pos);
            if (nl) end = (size_t)((const uint8_t*)nl - data) + 1;
            else { const void* fw = memchr(data + end, '\n', len - end); end = fw ? (size_t)((const uint8_t*)fw - data) + 1 : len; }
        }
        // this thread submits AND gathers: make room before a submit that would block on the in-flight bound
        while (matchy_multi_scanner_pending(h) >= ms->max_inflight) take_one();
        const int32_t src = matchy_multi_scanner_submit(h, data + pos, end - pos, nullptr, nullptr);
        if (src != MATCHY_SUCCESS) { if (status == MATCHY_SUCCESS) { status = src; err = "matchy_multi_scanner_scan: a line of 4 GiB or more cannot be submitted"; } break; }
        pos = end;
        take(false);
    }
    take(true);
    if (status != MATCHY_SUCCESS) { set_error(err); return status; }
    memset(out, 0, sizeof(*out));
    out->lines = lines; out->candidates = cands; out->bytes = len;
    out->n_hits = in->hits.size(); out->n_ids = in->ids.size();
    out->hits = in->hits.data(); out->pattern_ids = in->ids.data(); out->data_offsets = in->offs.data();
    out->_internal = in.release();
    return MATCHY_SUCCESS;
}

// A regular file (mapped; a reader thread cuts it into newline-aligned batches, faults their pages in, pins them and submits them) or a
// stream ("-" = stdin, a pipe: read into buffers of batch_bytes). `fn` is called on the calling thread for every batch IN FILE ORDER
// with the batch's result and its offset in the file (batch->tag); the result is released when fn returns. Compressed inputs are the
// caller's business (decompress and matchy_multi_scanner_submit: `matchy match` does that for .gz). Returns 0, or the first error.
int32_t matchy_multi_scanner_scan_file(matchy_multi_scanner_t* h, const char* path, size_t batch_bytes, matchy_multi_ordered_fn fn, void* user, matchy_multi_totals_t* totals) {
    if (!h || !path) return MATCHY_ERROR_INVALID_PARAM;
    MultiScanner* ms = reinterpret_cast<MultiScanner*>(h);
    { std::lock_guard<std::mutex> lk(ms->mu); if (ms->taken != ms->submitted) { set_error("matchy_multi_scanner_scan_file: batches of an earlier submit are still pending"); return MATCHY_ERROR_INVALID_PARAM; } }
    if (!batch_bytes) batch_bytes = (size_t)256 << 20;
    if (batch_bytes > 0xF0000000ull) batch_bytes = 0xF0000000ull;
    const bool is_stdin = strcmp(path, "-") == 0;
    const int fd = is_stdin ? 0 : open(path, O_RDONLY);
    if (fd < 0) { set_error(std::string("matchy_multi_scanner_scan_file: cannot open ") + path + ": " + strerror(errno)); return MATCHY_ERROR_FILE_NOT_FOUND; }
    struct stat sb;
    const bool regular = !is_stdin && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0;
    void* map = MAP_FAILED;
    size_t map_len = 0;
    if (regular) { map_len = (size_t)sb.st_size; map = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, 0); }
    std::atomic<bool> reader_ok{true}, stop{false}, reader_finished{false};
    const bool mapped = map != MAP_FAILED;
    std::thread reader([&] {
        struct Finished { std::atomic<bool>& f; ~Finished() { f = true; } } fin{reader_finished};
        if (mapped) {
            (void)madvise(map, map_len, MADV_SEQUENTIAL);
            const uint8_t* base = (const uint8_t*)map;
            for (size_t pos = 0; pos < map_len && !stop;) {
                size_t end = std::min(map_len, pos + batch_bytes);
                if (end < map_len) {
                    const void* nl = memrchr(base + pos, '\n', end - pos);
                    if (nl) end = (size_t)((const uint8_t*)nl - base) + 1;
                    else { const void* fw = memchr(base + end, '\n', map_len - end); end = fw ? (size_t)((const uint8_t*)fw - base) + 1 : map_len; }
                }
                const uint8_t* p = base + pos;
                const size_t n = end - pos;
                const void* pinned = nullptr;
                if (n >= ((size_t)4 << 20)) {
                    const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, z = ((uintptr_t)p + n) & ~(uintptr_t)4095;
#ifdef MADV_POPULATE_READ
                    (void)madvise((void*)((uintptr_t)p & ~(uintptr_t)4095), (uintptr_t)p + n - ((uintptr_t)p & ~(uintptr_t)4095), MADV_POPULATE_READ);
#endif
                    if (z > a && matchy_amd_host_register((const void*)a, z - a) == MATCHY_SUCCESS) pinned = (const void*)a;
                }
                if (matchy_multi_scanner_submit(h, p, n, (void*)(uintptr_t)pos, pinned) != MATCHY_SUCCESS) {   // a line of 4 GiB or more
                    if (pinned) matchy_amd_host_unregister(pinned);
                    reader_ok = false;
                    return;
                }
                pos = end;
            }
            return;
        }
        // stream: read batch_bytes, cut at the last newline, carry the rest into the next buffer; a submitted buffer belongs to the
        // gathering thread, which frees it when its batch has been handed to the callback
        size_t cap = batch_bytes + 16, have = 0;
        uint64_t off = 0;
        uint8_t* buf = (uint8_t*)malloc(cap);
        if (!buf) { reader_ok = false; return; }
        for (;;) {
            if (have == cap - 16) { cap *= 2; uint8_t* nb = (uint8_t*)realloc(buf, cap); if (!nb) { reader_ok = false; break; } buf = nb; }
            const ssize_t r = read(fd, buf + have, std::min(cap - 16 - have, (size_t)1 << 30));
            if (r < 0) { if (errno == EINTR) continue; reader_ok = false; break; }
            have += (size_t)r;
            const bool eof = r == 0;
            if (!eof && have < batch_bytes) continue;
            size_t cut = have;
            if (!eof) { const void* nl = memrchr(buf, '\n', have); if (!nl) continue; cut = (size_t)((const uint8_t*)nl - buf) + 1; }
            uint8_t* nxt = nullptr;
            const size_t rest = have - cut;
            if (!eof) { nxt = (uint8_t*)malloc(std::max(batch_bytes, rest) + 16); if (!nxt) { reader_ok = false; break; } memcpy(nxt, buf + cut, rest); }
            if (cut) {
                if (matchy_multi_scanner_submit(h, buf, cut, (void*)(uintptr_t)off, nullptr) != MATCHY_SUCCESS) { free(buf); free(nxt); buf = nullptr; reader_ok = false; break; }
            } else free(buf);
            off += cut;
            buf = nxt; cap = std::max(batch_bytes, rest) + 16; have = rest;
            if (eof || stop) break;
        }
        if (buf) free(buf);
    });
    // the calling thread gathers in order while the reader is still submitting
    int32_t status = MATCHY_SUCCESS;
    std::string err;
    matchy_multi_totals_t t{};
    for (;;) {
        matchy_multi_batch_t b;
        int32_t r = matchy_multi_scanner_next(h, &b);
        if (r == 0) {
            if (!reader_finished) { std::this_thread::sleep_for(std::chrono::microseconds(200)); continue; }   // between two submits
            r = matchy_multi_scanner_next(h, &b);   // the reader has submitted its last batch: anything still pending?
            if (r == 0) break;
        }
        if (r != 1) { if (status == MATCHY_SUCCESS) { status = r; err = "matchy_multi_scanner_scan_file: gather failed"; } break; }
        if (b.status != MATCHY_SUCCESS) { if (status == MATCHY_SUCCESS) { status = b.status; err = matchy_amd_last_error(); stop = true; } }
        else {
            t.batches += 1; t.bytes += b.len; t.lines += b.result.lines; t.candidates += b.result.candidates; t.matches += b.result.n_hits + b.result.n_ip4_hits;
            if (fn && status == MATCHY_SUCCESS) { const int32_t fr = fn(user, &b); if (fr != 0) { status = fr; err = "matchy_multi_scanner_scan_file: the batch callback asked to stop"; stop = true; } }
        }
        matchy_scan_result_free(&b.result);
        if (!mapped) free(const_cast<uint8_t*>(b.data));
    }
    reader.join();
    if (map != MAP_FAILED) munmap(map, map_len);
    if (!is_stdin) close(fd);
    if (totals) *totals = t;
    if (!reader_ok && status == MATCHY_SUCCESS) { status = MATCHY_ERROR_IO; err = std::string("matchy_multi_scanner_scan_file: reading ") + path + " failed"; }
    if (status != MATCHY_SUCCESS) set_error(err);
    return status;
}

}  // extern "C"
