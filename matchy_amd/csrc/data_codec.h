// MMDB data-section codec used by the database builder (write) and by hit decoding (read).
// Format follows the reference's data crate (crates/matchy-data-format/src/lib.rs:294-623 encoder,
// :635-1048 decoder): control byte type<<5|size, size extensions 29/30/31, extended types via a second
// byte, pointers of 11/19/27/32 bits relative to the data-section start.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace mxy {

struct DataValue {
    enum Type : uint8_t {
        POINTER = 1, STRING = 2, DOUBLE = 3, BYTES = 4, UINT16 = 5, UINT32 = 6, MAP = 7, INT32 = 8, UINT64 = 9,
        UINT128 = 10, ARRAY = 11, BOOL = 14, FLOAT = 15
    };
    Type type = STRING;
    std::string str;            // STRING, BYTES
    uint64_t u = 0, uhi = 0;    // UINT16/32/64/128 (uhi = high half), BOOL, POINTER
    int32_t i32 = 0;
    double f64 = 0;
    float f32 = 0;
    std::map<std::string, DataValue> map;
    std::vector<DataValue> arr;

    static DataValue String(std::string s) { DataValue v; v.type = STRING; v.str = std::move(s); return v; }
    static DataValue Uint16(uint16_t x) { DataValue v; v.type = UINT16; v.u = x; return v; }
    static DataValue Uint32(uint32_t x) { DataValue v; v.type = UINT32; v.u = x; return v; }
    static DataValue Uint64(uint64_t x) { DataValue v; v.type = UINT64; v.u = x; return v; }
    static DataValue Int32(int32_t x) { DataValue v; v.type = INT32; v.i32 = x; return v; }
    static DataValue Double(double x) { DataValue v; v.type = DOUBLE; v.f64 = x; return v; }
    static DataValue Bool(bool x) { DataValue v; v.type = BOOL; v.u = x; return v; }
    static DataValue Map() { DataValue v; v.type = MAP; return v; }
    static DataValue Array() { DataValue v; v.type = ARRAY; return v; }
};

// How JSON / CSV scalars become typed values.
enum class NumberTyping {
    // serde `Deserialize for DataValue` (data-format/lib.rs:106-206), used by matchy_builder_add:
    // non-negative integers → Uint16/Uint32/Uint64 by magnitude, negative → Int32 (or Double below i32::MIN), floats → Double
    SERDE,
    // CLI build path (bin/cli_utils.rs:203-236, build_cmd.rs:226-236): any i64 → Int32 (truncating), else u64 → Uint64, else f64 → Double
    CLI,
};

class DataEncoder {
public:
    // Encode `v`, returning its offset; identical values return the first offset (whole-value dedup on the
    // non-interned serialisation), strings and map keys seen before are written as pointers (interning).
    uint32_t encode(const DataValue& v);
    const std::vector<uint8_t>& bytes() const { return buf_; }
    std::vector<uint8_t> take() { return std::move(buf_); }

    static void encode_plain(const DataValue& v, std::vector<uint8_t>& out);

private:
    void encode_interned(const DataValue& v);
    std::vector<uint8_t> buf_;
    std::unordered_map<std::string, uint32_t> dedup_;
    std::unordered_map<std::string, uint32_t> strings_;
};

// Decode the value at `offset` of a data section, resolving pointers (decode + resolve_pointers).
bool decode_value(const uint8_t* section, size_t len, uint32_t offset, DataValue& out);

// serde_json-compatible compact JSON with sorted object keys (bin/cli_utils.rs:177-201 value mapping).
void to_json(const DataValue& v, std::string& out);
void json_escape(const std::string& s, std::string& out);

// Minimal JSON parser (objects, arrays, strings with escapes, numbers, true/false/null).
bool parse_json(const char* text, size_t len, NumberTyping typing, DataValue& out, std::string& err);

}  // namespace mxy
