// CPU test driver for csrc/host_lookup.cpp (the single-query path of matchy_query): opens a .mxy with DbImage — no GPU, no library —
// and answers the queries of a file (one per line; empty lines are empty queries) as one line each:
//   {"kind":"notfound"} | {"kind":"ip","prefix_len":N,"data":...} | {"kind":"pattern","pattern_ids":[...],"data":[...]}
// tests/test_host_units.py compares them with the oracle's Database::lookup.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "data_codec.h"
#include "db_image.h"
#include "host_lookup.h"
#include "netaddr.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    mxy::DbImage img;
    std::string err;
    if (!img.open(std::move(bytes), err)) { fprintf(stderr, "open failed: %s\n", err.c_str()); return 1; }
    const mxy::HostTables tables(img);
    std::ifstream q(argv[2], std::ios::binary);
    std::string line;
    while (std::getline(q, line)) {
        mxy::IpAddr ip;
        const bool is_ip = mxy::parse_ip(line.data(), line.size(), ip);
        mxy::HostHit h;
        mxy::host_lookup(img, tables, line, is_ip ? &ip : nullptr, h);
        // the shape of the oracle's lookup JSON: literal id first (when the table maps it to data: database.rs:916-930), then the glob ids
        if (h.kind == 0) { puts("{\"kind\":\"notfound\"}"); continue; }
        std::string o;
        auto data = [&](uint32_t off) { mxy::DataValue dv; if (img.decode_data(off, dv)) mxy::to_json(dv, o); else o += "null"; };
        if (h.kind == 2) {
            o = "{\"kind\":\"ip\",\"prefix_len\":" + std::to_string((unsigned)h.prefix_len) + ",\"data\":";
            data(h.a);
            o += "}";
        } else {
            std::vector<uint32_t> ids, offs;
            std::vector<bool> has;
            uint32_t off;
            if (h.a != 0xFFFFFFFFu && img.lit_data_offset(h.a, off)) { ids.push_back(h.a); offs.push_back(off); has.push_back(true); }
            for (uint32_t g : h.globs) { ids.push_back(g); const bool ok = img.glob_data_offset(g, off); offs.push_back(ok ? off : 0); has.push_back(ok); }
            if (ids.empty()) { puts("{\"kind\":\"notfound\"}"); continue; }
            o = "{\"kind\":\"pattern\",\"pattern_ids\":[";
            for (size_t i = 0; i < ids.size(); ++i) { if (i) o += ","; o += std::to_string(ids[i]); }
            o += "],\"data\":[";
            for (size_t i = 0; i < ids.size(); ++i) { if (i) o += ","; if (has[i]) data(offs[i]); else o += "null"; }
            o += "]}";
        }
        puts(o.c_str());
    }
    return 0;
}
