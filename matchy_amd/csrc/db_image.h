// Host view of an opened .mxy image: section discovery with bounds validation at open (instead of the
// reference's per-access checks), plus the derived tables the device upload needs.
// Follows Database::from_storage (crates/matchy/src/database.rs:649-713, 1023-1069, 1218-1415).
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "data_codec.h"
#include "scan_types.h"

namespace mxy {

struct DbImage {
    std::vector<uint8_t> bytes;

    uint32_t node_count = 0;
    int record_size = 24;
    int ip_version = 4;
    size_t tree_size = 0;
    bool has_ip = false;
    int match_mode = 0;
    DataValue metadata;

    // literal hash section (relative to lh)
    bool has_literal = false;
    size_t lh_off = 0, lh_len = 0;
    uint32_t lh_table_size = 0, lh_strings_offset = 0, lh_strings_size = 0, lh_num_shards = 0;
    size_t lh_table_start = 0;
    std::vector<uint32_t> lit_data_offsets;               // dense by pattern id when ids are 0..n-1
    std::unordered_map<uint32_t, uint32_t> lit_data_map;  // general fallback (first mapping wins, lh:560-572)

    // paraglob section
    bool has_glob = false;
    size_t pg_off = 0, pg_len = 0;
    size_t pdm_off = 0, pdm_count = 0;  // PatternDataMappings (database.rs:203-229)
    uint32_t pattern_count = 0;

    bool open(std::vector<uint8_t>&& data, std::string& err);
    // Structural validation of everything the device kernels and the host decoders dereference without a per-access check
    // of their own (called by open(): a file that fails is never uploaded; matchy_validate reports the message): data
    // pointers of the IP tree, the string of every literal-hash slot, the wildcard / pattern / glob-segment arrays and the
    // reachable part of the Aho-Corasick automaton (edges, dense tables, failure links, output lists). All arithmetic in
    // 64 bits.
    bool check_structure(std::string& err) const;
    // largest record of the IP tree (check_structure): above node_count + 16 it is the largest data-section offset an IP result can carry
    mutable uint32_t max_ip_record = 0;
    uint32_t max_ip_data_offset() const { return max_ip_record > node_count + 16 ? max_ip_record - node_count - 16 : 0u; }

    const uint8_t* data_section() const { return bytes.data() + tree_size + 16; }
    size_t data_section_len() const { return bytes.size() - (tree_size + 16); }
    bool decode_data(uint32_t offset, DataValue& out) const { return decode_value(data_section(), data_section_len(), offset, out); }
    bool lit_data_offset(uint32_t pid, uint32_t& off) const;
    bool glob_data_offset(uint32_t pid, uint32_t& off) const;
    std::string format_name() const;
    std::string pattern_string(uint32_t pid) const;

    // derived host-side tables for the device image
    void build_ip_nodes(std::vector<uint2>& out, uint32_t& v4_start) const;
    void build_lit_table(std::vector<LitSlot>& slots, uint32_t& mask) const;
    void build_lit2pat(std::vector<uint32_t>& off, std::vector<uint32_t>& ids) const;
    // Aho-Corasick automaton of the paraglob section flattened into a dense DFA (goto and failure links resolved):
    // next[state * k + cls[byte]] = next state | 0x80000000 when that state has output literals; node_off[state] is the
    // node's offset in the AC section. Returns false (nothing built) when the section is empty, malformed in a way the
    // flattening does not handle, or the table would exceed `max_bytes`; the kernels then walk the stored nodes.
    // *alnum_literal (optional): some literal consists of ASCII letters and digits only (a node with output whose path from the root does).
    bool build_ac_dfa(std::vector<uint32_t>& next, std::vector<uint8_t>& cls, uint32_t& k, std::vector<uint32_t>& node_off,
                      size_t max_bytes, bool* alnum_literal = nullptr) const;
};

}  // namespace mxy
