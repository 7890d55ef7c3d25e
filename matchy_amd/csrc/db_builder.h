// .mxy database builder: IP search tree + MMDB data section + PARAGLOB (AC + ACLH + glob segments)
// + LHSH literal table + metadata, laid out as the reference writes them so that either side can read
// the other's files.
//
// Interface mirrors the reference's `DatabaseBuilder` (crates/matchy-format/src/mmdb_builder.rs:43-760):
// add_entry (auto-detect, with literal:/glob:/ip: prefixes), add_ip, add_literal, add_glob, build.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "data_codec.h"
#include "netaddr.h"

namespace mxy {

enum class EntryKind { IP, LITERAL, GLOB };

struct BuildStats {
    size_t ip_entries = 0, literal_entries = 0, glob_entries = 0;
    uint32_t node_count = 0;
    int record_size = 24;
    int ip_version = 4;
    size_t data_section_bytes = 0;
    size_t ac_nodes = 0;
    bool record_size_bumped = false;  // deviation from the reference: see DatabaseBuilder::build
};

class DatabaseBuilder {
public:
    explicit DatabaseBuilder(bool case_insensitive = false) : case_insensitive_(case_insensitive) {}

    // All return false and set error() on invalid input (reference: FormatError::InvalidPattern).
    bool add_entry(const std::string& key, const DataValue& data_map);
    bool add_ip(const std::string& ip_or_cidr, const DataValue& data_map);
    bool add_literal(const std::string& pattern, const DataValue& data_map);
    bool add_glob(const std::string& pattern, const DataValue& data_map);

    void set_database_type(const std::string& t) { database_type_ = t; }
    void set_description(const std::string& lang, const std::string& text) { description_[lang] = text; }
    void set_case_insensitive(bool ci) { case_insensitive_ = ci; }
    void set_build_epoch(uint64_t e) { build_epoch_ = e; has_epoch_ = true; }

    bool build(std::vector<uint8_t>& out);
    const std::string& error() const { return error_; }
    const BuildStats& stats() const { return stats_; }

    // detect_entry_type (mmdb_builder.rs:392-429). On success fills kind and the (prefix-stripped) key / address.
    static bool detect_entry_type(const std::string& key, EntryKind& kind, std::string& stripped, IpAddr& addr,
                                  uint8_t& prefix_len, std::string& err);
    static bool parse_ip_entry(const std::string& key, IpAddr& addr, uint8_t& prefix_len);

private:
    struct Entry {
        EntryKind kind;
        IpAddr addr;
        uint8_t prefix_len = 0;
        std::string text;
        uint32_t data_offset = 0;
    };
    uint32_t encode_data(const DataValue& data_map) { return encoder_.encode(data_map); }

    std::vector<Entry> entries_;
    DataEncoder encoder_;
    bool case_insensitive_;
    std::string database_type_;
    std::map<std::string, std::string> description_;
    uint64_t build_epoch_ = 0;
    bool has_epoch_ = false;
    std::string error_;
    BuildStats stats_;
};

// Glob syntax check: GlobPattern::new (crates/matchy-paraglob/src/glob.rs:307-451).
bool validate_glob_pattern(const std::string& pattern, std::string& err);

}  // namespace mxy
