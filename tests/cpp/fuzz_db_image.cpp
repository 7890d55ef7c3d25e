// Mutation fuzzer for the host side of the .mxy reader (DbImage::open / check_structure and everything the device upload
// derives from an accepted image), meant to be built with -fsanitize=address,undefined (tests/test_host_units.py).
// A mutated file must either be rejected with a message or be read without any out-of-bounds access: the device kernels
// dereference the uploaded tables without per-access checks, so what open() accepts must be structurally sound.
// Usage: fuzz_db_image <iterations> <seed> [file.mxy ...]   (with no files: only the databases built in-process)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "db_builder.h"
#include "db_image.h"
#include "host_lookup.h"
#include "netaddr.h"

using namespace mxy;

static uint64_t rng_state;
static uint64_t rnd() {   // splitmix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static std::vector<uint8_t> built_db(bool ci, bool v6, int n) {
    DatabaseBuilder b(ci);
    DataValue d = DataValue::Map();
    d.map["threat"] = DataValue::String("x");
    d.map["n"] = DataValue::Uint32(7);
    for (int i = 0; i < n; ++i) {
        char buf[96];
        snprintf(buf, sizeof buf, "10.%d.%d.%d/%d", i % 250, (i * 7) % 250, (i * 13) % 250, 16 + i % 17);
        b.add_entry(buf, d);
        if (v6) { snprintf(buf, sizeof buf, "2001:db8:%x::%x/%d", i, i * 3, 48 + i % 80); b.add_entry(buf, d); }
        snprintf(buf, sizeof buf, "host%d.example%d.com", i, i % 7);
        b.add_entry(buf, d);
        snprintf(buf, sizeof buf, "*.bad%d.example.org", i);
        b.add_entry(buf, d);
        snprintf(buf, sizeof buf, "evil%d-*-[a-c]?.net", i % 11);
        b.add_entry(buf, d);
    }
    b.set_build_epoch(1);
    std::vector<uint8_t> out;
    if (!b.build(out)) { fprintf(stderr, "seed database did not build: %s\n", b.error().c_str()); exit(2); }
    return out;
}

static size_t walk_value(const DataValue& v, int depth) {   // touch everything a decoded value holds
    size_t n = v.str.size() + (size_t)v.u + (size_t)v.i32;
    if (depth > 64) return n;
    for (const auto& kv : v.map) n += kv.first.size() + walk_value(kv.second, depth + 1);
    for (const auto& e : v.arr) n += walk_value(e, depth + 1);
    return n;
}

// Everything the upload and the result decoding do with an accepted image (engine.cpp DeviceDb, capi.cpp)
static size_t exercise(const DbImage& img) {
    size_t sink = 0;
    std::vector<uint2> nodes;
    uint32_t v4_start = 0;
    img.build_ip_nodes(nodes, v4_start);
    sink += nodes.size() + v4_start;
    // data pointers of the tree, as the lookup results are decoded (record value > node_count + 16 -> data offset)
    size_t decoded = 0;
    for (const uint2& nd : nodes) {
        for (uint32_t rec : {nd.x, nd.y}) {
            if (rec > img.node_count && rec - img.node_count >= 16 && decoded < 4096) {
                DataValue v;
                if (img.decode_data(rec - img.node_count - 16, v)) sink += walk_value(v, 0);
                ++decoded;
            }
        }
    }
    if (img.has_literal) {
        std::vector<LitSlot> slots;
        uint32_t mask = 0;
        img.build_lit_table(slots, mask);
        sink += slots.size() + mask;
        for (uint32_t pid = 0; pid < 64; ++pid) {
            uint32_t off;
            if (img.lit_data_offset(pid, off)) { DataValue v; if (img.decode_data(off, v)) sink += walk_value(v, 0); }
        }
    }
    if (img.has_glob) {
        std::vector<uint32_t> off, ids;
        img.build_lit2pat(off, ids);
        sink += off.size() + ids.size();
        std::vector<uint32_t> nx, noff;
        std::vector<uint8_t> cls;
        uint32_t k = 0;
        if (img.build_ac_dfa(nx, cls, k, noff, (size_t)2 << 20)) sink += nx.size() + k;
        const uint32_t np = img.pattern_count < 256 ? img.pattern_count : 256;
        for (uint32_t pid = 0; pid < np; ++pid) {
            sink += img.pattern_string(pid).size();
            uint32_t o;
            if (img.glob_data_offset(pid, o)) { DataValue v; if (img.decode_data(o, v)) sink += walk_value(v, 0); }
        }
    }
    sink += walk_value(img.metadata, 0) + img.format_name().size();
    // ... and the single-query path of matchy_query (csrc/host_lookup.cpp) walks the same tables on the host: addresses, keys the seeds
    // hold, names their globs match, long and non-ASCII texts. Whatever it answers on a mutated image, it answers within the image.
    {
        static const char* const Q[] = {
            "10.0.0.0", "10.7.49.91", "10.249.243.237", "255.255.255.255", "0.0.0.0", "2001:db8:3::9", "2001:db8:18::48", "::1", "::ffff:10.1.2.3",
            "host0.example0.com", "host24.example3.com", "HOST3.EXAMPLE3.COM", "www.bad7.example.org", "a.b.c.bad39.example.org", "evil3-x-a1.net",
            "evil10-some-thing-cz.net", "evil.example.com", "x.evil.example.com", "", "a", "caf\xc3\xa9.fr", "\xc4\xb0stanbul.example.org",
            "aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa.bad1.example.org",
        };
        const HostTables tables(img);
        sink += tables.v4_start_node + tables.lit2pat.size();
        for (const char* q : Q) {
            const std::string query(q);
            IpAddr ip;
            const bool is_ip = parse_ip(query.data(), query.size(), ip);
            HostHit h;
            host_lookup(img, tables, query, is_ip ? &ip : nullptr, h);
            sink += h.kind + h.prefix_len + h.globs.size();
            uint32_t off = 0;
            if (h.kind == 2) { DataValue v; if (img.decode_data(h.a, v)) sink += walk_value(v, 0); }
            if (h.kind == 3 && h.a != 0xFFFFFFFFu && img.lit_data_offset(h.a, off)) { DataValue v; if (img.decode_data(off, v)) sink += walk_value(v, 0); }
            for (uint32_t g : h.globs) if (img.glob_data_offset(g, off)) { DataValue v; if (img.decode_data(off, v)) sink += walk_value(v, 0); }
        }
    }
    return sink;
}

// JSON entry data (matchy_builder_add, `matchy build -f json`): a mutated text either fails to parse with a message or yields a
// value that survives data-section encoding and decoding unchanged (compared through the canonical JSON rendering).
static long fuzz_json(long iters) {
    static const char* SEEDS[] = {
        R"({"threat_level":"high","score":97,"tags":["c2","botnet"],"first_seen":1700000000,"ratio":0.25,"active":true,"parent":null})",
        R"({"a":{"b":{"c":[1,2,3,{"d":"\u00e9\n\t\"q\""}]}},"neg":-5,"big":18446744073709551615,"exp":1e-7,"e":[]})",
        R"([{"entry":"1.2.3.4","data":{"k":"v"}},{"entry":"*.evil.com","data":{"n":65536,"m":-2147483649}}])",
    };
    long parsed = 0;
    for (long it = 0; it < iters; ++it) {
        std::string t = SEEDS[rnd() % 3];
        const int nmut = (int)(rnd() % 4);
        for (int m = 0; m < nmut && !t.empty(); ++m) {
            const size_t pos = rnd() % t.size();
            switch (rnd() % 5) {
                case 0: t[pos] = (char)rnd(); break;
                case 1: t.erase(pos, 1 + rnd() % 4); break;
                case 2: { static const char CH[] = "{}[]\",:0-9eE.tfn\\u"; t.insert(pos, 1, CH[rnd() % (sizeof CH - 1)]); break; }
                case 3: t.insert(pos, t.substr(rnd() % t.size(), rnd() % 24)); break;
                default: t.resize(pos);
            }
        }
        for (NumberTyping ty : {NumberTyping::SERDE, NumberTyping::CLI}) {
            DataValue v; std::string err;
            if (!parse_json(t.data(), t.size(), ty, v, err)) { if (err.empty()) { fprintf(stderr, "JSON rejected without a message: %s\n", t.c_str()); exit(1); } continue; }
            ++parsed;
            DataEncoder enc;
            const uint32_t off = enc.encode(v);
            DataValue back;
            std::string j1, j2;
            to_json(v, j1);
            if (!decode_value(enc.bytes().data(), enc.bytes().size(), off, back)) { fprintf(stderr, "encoded value does not decode: %s\n", j1.c_str()); exit(1); }
            to_json(back, j2);
            if (j1 != j2) { fprintf(stderr, "round trip changed the value:\n %s\n %s\n", j1.c_str(), j2.c_str()); exit(1); }
        }
    }
    return parsed;
}

int main(int argc, char** argv) {
    const long iters = argc > 1 ? atol(argv[1]) : 2000;
    rng_state = argc > 2 ? strtoull(argv[2], nullptr, 0) : 1;
    std::vector<std::vector<uint8_t>> seeds;
    seeds.push_back(built_db(false, false, 40));
    seeds.push_back(built_db(true, true, 25));
    for (int a = 3; a < argc; ++a) {
        std::ifstream f(argv[a], std::ios::binary);
        std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (d.empty()) { fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
        seeds.push_back(std::move(d));
    }
    // the unmodified seeds must open
    for (const auto& s : seeds) {
        DbImage img; std::string err;
        std::vector<uint8_t> copy = s;
        if (!img.open(std::move(copy), err)) { fprintf(stderr, "seed rejected: %s\n", err.c_str()); return 1; }
        (void)exercise(img);
    }
    long accepted = 0, rejected = 0;
    size_t sink = 0;
    for (long it = 0; it < iters; ++it) {
        std::vector<uint8_t> d = seeds[rnd() % seeds.size()];
        const int nmut = 1 + (int)(rnd() % 4);
        for (int m = 0; m < nmut && !d.empty(); ++m) {
            const size_t pos = rnd() % d.size();
            switch (rnd() % 8) {
                case 0: d[pos] ^= (uint8_t)(1u << (rnd() % 8)); break;                          // bit flip
                case 1: d[pos] = (uint8_t)rnd(); break;                                          // random byte
                case 2: {                                                                        // extreme 32-bit value
                    static const uint32_t X[] = {0u, 1u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFFu, 0xFFFFFFF0u, 0x00FFFFFFu, 0x01000000u};
                    const uint32_t v = X[rnd() % 8];
                    const size_t p4 = pos & ~(size_t)3;
                    if (p4 + 4 <= d.size()) memcpy(&d[p4], &v, 4);
                    break;
                }
                case 3: d.resize(pos); break;                                                    // truncate
                case 4: {                                                                        // copy a block over another
                    const size_t src = rnd() % d.size(), n = rnd() % 64;
                    for (size_t k = 0; k < n && pos + k < d.size() && src + k < d.size(); ++k) d[pos + k] = d[src + k];
                    break;
                }
                case 5: {                                                                        // small delta on a 32-bit field
                    const size_t p4 = pos & ~(size_t)3;
                    if (p4 + 4 <= d.size()) { uint32_t v; memcpy(&v, &d[p4], 4); v += (uint32_t)(rnd() % 65) - 32u; memcpy(&d[p4], &v, 4); }
                    break;
                }
                case 6: d.insert(d.begin() + (long)pos, (size_t)(rnd() % 16), (uint8_t)rnd()); break;   // insert
                default: {                                                                       // zero a block
                    const size_t n = rnd() % 128;
                    for (size_t k = 0; k < n && pos + k < d.size(); ++k) d[pos + k] = 0;
                }
            }
        }
        DbImage img; std::string err;
        if (getenv("FUZZ_DUMP")) {   // FUZZ_DUMP=<iteration>: write that iteration's image to fuzz_case.mxy (to reproduce a finding)
            if (it == atol(getenv("FUZZ_DUMP"))) { std::ofstream o("fuzz_case.mxy", std::ios::binary); o.write((const char*)d.data(), (long)d.size()); }
        }
        if (getenv("FUZZ_TRACE")) fprintf(stderr, "it %ld size %zu\n", it, d.size());
        if (img.open(std::move(d), err)) { ++accepted; sink += exercise(img); }
        else { ++rejected; if (err.empty()) { fprintf(stderr, "rejected without a message\n"); return 1; } }
    }
    const long jp = fuzz_json(iters);
    printf("OK: %ld mutated images (%ld accepted and exercised, %ld rejected) sink=%zu; %ld mutated JSON texts (%ld parses round-tripped)\n",
           iters, accepted, rejected, sink % 997, iters, jp);
    return 0;
}
