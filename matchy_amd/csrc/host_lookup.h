// Single-query lookup on the host (round 5). SURVEY §8b: "single queries stay on the CPU path (latency); the GPU is for bulk scans" — a
// `matchy_query` that launches kernels pays a launch + a synchronisation (tens of microseconds) where the reference answers in 0.2 us
// (book/src/architecture/performance-results.md:29-51). This is that CPU path, written against the on-disk format the uploader reads
// (DbImage): MMDB trie walk (crates/matchy-format/src/mmdb/tree.rs:46-277), literal hash probe
// (crates/matchy-literal-hash/src/lib.rs:467-543) and Paraglob::find_all — Aho-Corasick walk, literal -> pattern map, glob verifier with
// its 100 000-step budget (crates/matchy-paraglob/src/paraglob_offset.rs:1028-1639). It serves `matchy_query*` / `matchy_amd_query_json`
// ONLY; every scan and every extraction runs the HIP kernels (engine.cpp), and the GPU tests hold the two paths against each other.
// Not the oracle: nothing under oracle/ is included, linked or called from here.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "db_image.h"
#include "netaddr.h"

namespace mxy {

struct HostHit {
    uint8_t kind = 0;          // 0 = nothing found, 2 = IP, 3 = pattern (the discriminants of Hit::kind)
    uint8_t prefix_len = 0;
    uint32_t a = 0xFFFFFFFFu;  // IP: data-section offset; pattern: literal pattern id or 0xFFFFFFFF
    std::vector<uint32_t> globs;   // glob pattern ids, ascending, no duplicates
};

// Tables derived once per opened database (first query): the start node of IPv4 addresses in an IPv6 tree and the
// AC literal -> pattern ids map (enumerated from the ACLH slots: no dependence on its slot hash).
struct HostTables {
    uint32_t v4_start_node = 0;
    std::vector<uint32_t> lit2pat_off, lit2pat;
    explicit HostTables(const DbImage& img);
};

// Database::lookup (database.rs:725-804) without its cache: `ip` non-null = the query parsed as an address.
void host_lookup(const DbImage& img, const HostTables& t, const std::string& query, const IpAddr* ip, HostHit& out);

}  // namespace mxy
