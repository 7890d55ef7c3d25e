// Device helpers shared by the validation and the lookup kernels: the IPv6 text parser and the flattened Aho-Corasick
// automaton (DbImage::build_ac_dfa) with its LDS-resident shallow part.
#pragma once
#include "device_common.h"

namespace mxy {

// SearchTree::lookup_v4 (matchy-format/src/mmdb/tree.rs:46-90). true + data offset + prefix length, or false for "not found".
// The first levels come from a table built at upload (DeviceDb::upload): the outcome of the first 24 levels for every /24 prefix
// (DevDb::ip_l24, 128 MB, when the tree is big enough to pay for it) and, for a /24 with entries below it, a leaf table with the
// outcome for each of its 256 addresses (ip_leaf) — two dependent loads per lookup. Small trees: the first 16 levels (ip_l1) and
// the reference's walk, one dependent 8-byte load per level.
struct IpTables { const uint2* l24; const uint2* l1; const uint2* leaf; const uint2* nodes; uint32_t node_count; };
__device__ __forceinline__ bool trie_v4_tables(const IpTables& t, uint32_t addr, uint32_t& data_off, uint32_t& prefix) {
    uint2 e = t.l24 ? t.l24[addr >> 8] : t.l1[addr >> 16];
    uint32_t kind = e.x & 0xFF;
    if (kind == 3) {   // leaf table of this /24: the outcome for each of its 256 addresses
        e = t.leaf[((size_t)e.y << 8) | (addr & 0xFFu)];
        kind = e.x & 0xFF;
    }
    if (kind == 1) return false;
    if (kind == 2) { data_off = e.y; prefix = e.x >> 8; return true; }
    uint32_t node = e.y;
    for (int bi = t.l24 ? 24 : 16; bi < 32; ++bi) {
        const uint2 nd = t.nodes[node];
        const uint32_t rec = ((addr >> (31 - bi)) & 1) ? nd.y : nd.x;
        if (rec == t.node_count) return false;
        if (rec < t.node_count) node = rec;
        else {
            const uint32_t off = rec - t.node_count;
            if (off < 16) return false;  // reference: MmdbError -> lookup error; treated as not found (never produced by builders)
            data_off = off - 16;
            prefix = (uint32_t)bi + 1;   // tree:76-80: depth counts from 96 in v6 trees and 96 is subtracted again
            return true;
        }
    }
    return false;
}
__device__ __forceinline__ bool trie_v4(const DevDb& db, uint32_t addr, uint32_t& data_off, uint32_t& prefix) {
    return trie_v4_tables(IpTables{db.ip_l24, db.ip_l1, db.ip_leaf, db.ip_nodes, db.node_count}, addr, data_off, prefix);
}

// Rust `<Ipv6Addr as FromStr>` restricted to [0-9A-Fa-f:] input (no embedded IPv4 possible): read_ipv6_addr.
__device__ bool d_parse_ipv6(const uint8_t* s, uint32_t n, uint16_t seg[8]) {
    uint32_t pos = 0;
    uint16_t head[8], tail[7];
    for (int i = 0; i < 8; ++i) head[i] = 0;
    for (int i = 0; i < 7; ++i) tail[i] = 0;
    auto read_groups = [&](uint16_t* g, uint32_t limit) -> uint32_t {
        for (uint32_t i = 0; i < limit; ++i) {
            uint32_t save = pos;
            if (i > 0) {
                if (pos < n && s[pos] == ':') ++pos;
                else { pos = save; return i; }
            }
            uint32_t v = 0, digits = 0, q = pos;
            bool ok = true;
            while (q < n && d_is_hex(s[q])) {
                uint32_t ch = s[q];
                v = v * 16 + (d_is_digit(ch) ? ch - '0' : (ch | 0x20) - 'a' + 10);
                ++digits;
                ++q;
                if (digits > 4) { ok = false; break; }
            }
            if (!ok || digits == 0) { pos = save; return i; }
            g[i] = (uint16_t)v;
            pos = q;
        }
        return limit;
    };
    uint32_t hs = read_groups(head, 8);
    if (hs == 8) {
        if (pos != n) return false;
        for (int i = 0; i < 8; ++i) seg[i] = head[i];
        return true;
    }
    if (!(pos < n && s[pos] == ':')) return false;
    ++pos;
    if (!(pos < n && s[pos] == ':')) return false;
    ++pos;
    uint32_t limit = 8 - (hs + 1);
    uint32_t ts = read_groups(tail, limit);
    if (pos != n) return false;
    for (uint32_t i = 0; i < ts; ++i) head[8 - ts + i] = tail[i];
    for (int i = 0; i < 8; ++i) seg[i] = head[i];
    return true;
}

// One transition of the flattened automaton. States are numbered breadth-first, so the rows of the shallowest states —
// where almost every step of a non-matching text happens — are the first ones; the kernels keep them in LDS.
struct DfaView { const uint8_t* cls; const uint32_t* rows; uint32_t lds_states; };   // cls, rows: LDS
__device__ __forceinline__ uint32_t dfa_step(const DevDb& db, const DfaView& dv, uint32_t st, uint32_t byte) {
    const uint32_t c = dv.cls[byte];
    return st < dv.lds_states ? dv.rows[st * db.dfa_k + c] : db.dfa[(size_t)st * db.dfa_k + c];
}
template <uint32_t ENTRIES>
__device__ __forceinline__ DfaView dfa_stage(const DevDb& db, uint8_t* cls, uint32_t* rows) {   // call before a __syncthreads()
    DfaView dv{cls, rows, 0};
    if (db.dfa) {
        dv.lds_states = min(db.dfa_states, ENTRIES / db.dfa_k);
        for (uint32_t k = threadIdx.x; k < 256; k += blockDim.x) cls[k] = db.dfa_cls[k];
        for (uint32_t k = threadIdx.x, nk = dv.lds_states * db.dfa_k; k < nk; k += blockDim.x) rows[k] = db.dfa[k];
    } else {
        for (uint32_t k = threadIdx.x; k < 256; k += blockDim.x) cls[k] = 0;
    }
    return dv;
}

}  // namespace mxy
