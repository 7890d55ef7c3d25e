// Unicode default lower-casing as Rust's `str::to_lowercase` does it (alloc/src/str.rs: per-character `to_lower` mapping,
// U+03A3 by the Final_Sigma rule) — what the reference applies to literal keys, AC literals and literal queries of
// case-insensitive databases (matchy-literal-hash/src/lib.rs:162-165, 469-472; matchy-ac/src/lib.rs:209).
// The mapping itself is data: matchy_amd/data/lowercase.bin (tools/gen_lowercase.py).
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace mxy {

struct LowerMapEntry { uint32_t cp; uint8_t len; uint8_t utf8[7]; };   // 12 bytes, as stored in the file
struct CpRange { uint32_t first, last; };

struct LowerTable {
    uint32_t unicode_version = 0;
    std::vector<LowerMapEntry> map;          // sorted by cp
    std::vector<CpRange> ignorable, cased;   // sorted, disjoint
    static const LowerTable& get();          // throws std::runtime_error if lowercase.bin cannot be found
    std::string to_lowercase(const std::string& s) const;   // s: valid UTF-8
};

}  // namespace mxy
