// Hash functions shared by host code and HIP kernels.
//   xxh64        XXH64 (seed 0 in all call sites) — literal-hash keys (crates/matchy-literal-hash/src/lib.rs:666-671)
//   fx_u32       rustc-hash 2.x FxHasher over one u32 — ACLH slot placement (matchy-paraglob/src/literal_hash.rs:95-99)
//   psl_hash     our own 64-bit hash for the device PSL table (any good hash works: membership is verified bytewise)
#pragma once
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MXY_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define MXY_HD inline
#endif

namespace mxy {

MXY_HD uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

// Byte-fetch functor based XXH64 so the same code runs over host pointers and over device "log + offset" views.
template <class Fetch>
MXY_HD uint64_t xxh64_fetch(Fetch f, size_t len, uint64_t seed) {
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL,
                   P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
    auto rd64 = [&](size_t o) {
        uint64_t v = 0;
        for (int k = 7; k >= 0; --k) v = (v << 8) | (uint64_t)f(o + k);
        return v;
    };
    auto rd32 = [&](size_t o) {
        uint64_t v = 0;
        for (int k = 3; k >= 0; --k) v = (v << 8) | (uint64_t)f(o + k);
        return v;
    };
    auto round = [&](uint64_t acc, uint64_t in) { acc += in * P2; acc = rotl64(acc, 31); return acc * P1; };
    auto merge = [&](uint64_t acc, uint64_t val) { val = round(0, val); acc ^= val; return acc * P1 + P4; };
    size_t p = 0;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        size_t limit = len - 32;
        do {
            v1 = round(v1, rd64(p)); v2 = round(v2, rd64(p + 8)); v3 = round(v3, rd64(p + 16)); v4 = round(v4, rd64(p + 24));
            p += 32;
        } while (p <= limit);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = merge(h, v1); h = merge(h, v2); h = merge(h, v3); h = merge(h, v4);
    } else {
        h = seed + P5;
    }
    h += (uint64_t)len;
    while (p + 8 <= len) { h ^= round(0, rd64(p)); h = rotl64(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= len) { h ^= rd32(p) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
    while (p < len) { h ^= (uint64_t)f(p) * P5; h = rotl64(h, 11) * P1; ++p; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// ASCII lower-casing of 8 packed bytes ('A'..'Z' get bit 5; bytes >= 0x80 and everything else stay)
MXY_HD uint64_t ascii_lower8(uint64_t v) {
    const uint64_t t = v & 0x7F7F7F7F7F7F7F7Full;
    const uint64_t up = (t + 0x3F3F3F3F3F3F3F3Full) & ~(t + 0x2525252525252525ull) & ~v & 0x8080808080808080ull;   // 0x41 <= byte <= 0x5A
    return v | (up >> 2);
}
MXY_HD uint32_t ascii_lower1(uint32_t c) { return (c - 'A' < 26u) ? c + 32 : c; }

// XXH64 over contiguous memory: 8- and 4-byte lanes are read with one (unaligned) load each. Host and gfx950 are
// little-endian, which is the byte order XXH64 specifies for its lanes. FOLD hashes the ASCII-lower-cased bytes instead
// (literal queries of case-insensitive databases whose text is pure ASCII).
template <bool FOLD = false>
MXY_HD uint64_t xxh64(const uint8_t* p, size_t len, uint64_t seed) {
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL,
                   P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
    auto rd64 = [&](size_t o) { uint64_t v; __builtin_memcpy(&v, p + o, 8); return FOLD ? ascii_lower8(v) : v; };
    auto rd32 = [&](size_t o) { uint32_t v; __builtin_memcpy(&v, p + o, 4); return FOLD ? (ascii_lower8((uint64_t)v) & 0xFFFFFFFFull) : (uint64_t)v; };
    auto round = [&](uint64_t acc, uint64_t in) { acc += in * P2; acc = rotl64(acc, 31); return acc * P1; };
    auto merge = [&](uint64_t acc, uint64_t val) { val = round(0, val); acc ^= val; return acc * P1 + P4; };
    size_t q = 0;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const size_t limit = len - 32;
        do {
            v1 = round(v1, rd64(q)); v2 = round(v2, rd64(q + 8)); v3 = round(v3, rd64(q + 16)); v4 = round(v4, rd64(q + 24));
            q += 32;
        } while (q <= limit);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = merge(h, v1); h = merge(h, v2); h = merge(h, v3); h = merge(h, v4);
    } else {
        h = seed + P5;
    }
    h += (uint64_t)len;
    if (len >= 8) {
        // The last up to 7 bytes come out of ONE 8-byte load that ends at the end of the input (it overlaps bytes already hashed,
        // which are shifted out), and for short inputs every load is issued before the chain of multiplies starts: on the GPU a
        // key is a handful of dependent memory round trips otherwise (one per 8 / 4 / 1-byte step of the tail).
        const uint64_t last = rd64(len - 8);
        if (len < 32) {
            const uint64_t w0 = rd64(0), w1 = rd64(len >= 16 ? 8 : len - 8), w2 = rd64(len >= 24 ? 16 : len - 8);
            h ^= round(0, w0); h = rotl64(h, 27) * P1 + P4;
            if (len >= 16) { h ^= round(0, w1); h = rotl64(h, 27) * P1 + P4; }
            if (len >= 24) { h ^= round(0, w2); h = rotl64(h, 27) * P1 + P4; }
            q = len & ~(size_t)7;
        } else {
            while (q + 8 <= len) { h ^= round(0, rd64(q)); h = rotl64(h, 27) * P1 + P4; q += 8; }
        }
        size_t r = len - q;                                   // 0..7 bytes left: the top r bytes of `last`
        uint64_t tail = r ? last >> (8 * (8 - r)) : 0;
        if (r >= 4) { h ^= (tail & 0xFFFFFFFFull) * P1; h = rotl64(h, 23) * P2 + P3; tail >>= 32; r -= 4; }
        for (; r; --r) { h ^= (tail & 0xFF) * P5; h = rotl64(h, 11) * P1; tail >>= 8; }
    } else {
        if (q + 4 <= len) { h ^= rd32(q) * P1; h = rotl64(h, 23) * P2 + P3; q += 4; }
        while (q < len) { h ^= (uint64_t)(FOLD ? ascii_lower1(p[q]) : (uint32_t)p[q]) * P5; h = rotl64(h, 11) * P1; ++q; }
    }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// XXH64 of a byte stream that is produced one byte at a time (the lower-cased form of a non-ASCII query of a
// case-insensitive database, which is never materialised): same digest as xxh64() of the concatenated bytes.
struct Xxh64Stream {
    static constexpr uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL,
                              P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
    uint64_t v1, v2, v3, v4, lane[4] = {0, 0, 0, 0}, cur = 0, total = 0;
    uint32_t nlane = 0, ncur = 0;   // complete 8-byte lanes of the current 32-byte stripe, bytes in `cur`
    MXY_HD explicit Xxh64Stream(uint64_t seed) : v1(seed + P1 + P2), v2(seed + P2), v3(seed), v4(seed - P1) {}
    static MXY_HD uint64_t round(uint64_t acc, uint64_t in) { acc += in * P2; acc = rotl64(acc, 31); return acc * P1; }
    static MXY_HD uint64_t merge(uint64_t acc, uint64_t val) { val = round(0, val); acc ^= val; return acc * P1 + P4; }
    MXY_HD void push(uint32_t byte) {
        cur |= (uint64_t)(byte & 0xFF) << (8 * ncur);
        ++total;
        if (++ncur == 8) {
            lane[nlane] = cur; cur = 0; ncur = 0;
            if (++nlane == 4) {
                v1 = round(v1, lane[0]); v2 = round(v2, lane[1]); v3 = round(v3, lane[2]); v4 = round(v4, lane[3]);
                nlane = 0;
            }
        }
    }
    MXY_HD uint64_t finish(uint64_t seed) const {
        uint64_t h;
        if (total >= 32) {
            h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
            h = merge(h, v1); h = merge(h, v2); h = merge(h, v3); h = merge(h, v4);
        } else {
            h = seed + P5;
        }
        h += total;
        for (uint32_t k = 0; k < nlane; ++k) { h ^= round(0, lane[k]); h = rotl64(h, 27) * P1 + P4; }
        uint64_t rest = cur;
        uint32_t n = ncur;
        if (n >= 4) { h ^= (rest & 0xFFFFFFFFull) * P1; h = rotl64(h, 23) * P2 + P3; rest >>= 32; n -= 4; }
        for (; n; --n) { h ^= (rest & 0xFF) * P5; h = rotl64(h, 11) * P1; rest >>= 8; }
        h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
        return h;
    }
};

// Hash of a name of n <= 31 bytes given as four little-endian 8-byte lanes (bytes at and past n are ignored) for the
// device-side "can any literal key be this name" bitmap (DevDb::lit_bm). Host (bitmap construction from the stored keys)
// and device (k_validate_dom) only have to agree with each other, so this is a cheap 32-bit multiply-xorshift mix instead
// of the XXH64 that the literal table itself is keyed by: 9 32-bit multiplies instead of ~16 64-bit ones per name.
MXY_HD uint32_t name_hash31(uint64_t l0, uint64_t l1, uint64_t l2, uint64_t l3, uint32_t n) {
    const uint64_t lane[4] = {l0, l1, l2, l3};
    uint32_t h = n * 0x9E3779B1u + 0x7F4A7C15u;
    for (int k = 0; k < 4; ++k) {
        const int rem = (int)n - 8 * k;   // bytes of this lane that belong to the name
        const uint64_t m = rem >= 8 ? ~0ull : rem <= 0 ? 0ull : ((1ull << (8 * rem)) - 1ull);
        const uint64_t v = lane[k] & m;
        h = (h ^ (uint32_t)v) * 0x85EBCA6Bu; h ^= h >> 15;
        h = (h ^ (uint32_t)(v >> 32)) * 0xC2B2AE35u; h ^= h >> 13;
    }
    h *= 0x27D4EB2Fu; h ^= h >> 16;
    return h;
}

// rustc-hash 2.x (64-bit): hash = (hash + x) * K, finish = rotl(hash, 26). See DESIGN.md (unverified vs crate source).
MXY_HD uint64_t fx_u32(uint32_t v) { return rotl64((uint64_t)v * 0xf1357aea2e62a9c5ULL, 26); }

// PSL hashing: suffixes are hashed from their LAST byte to their first so that a right-to-left walk over a
// domain can extend the hash one label at a time.
// slot of a last label of <= 7 bytes (packed little-endian in lo / the low 24 bits of hi) in the exact TLD table
constexpr uint32_t TLD_TAB_BITS = 11;
MXY_HD uint32_t tld_tab_slot(uint32_t lo, uint32_t hi24) { return ((lo ^ (hi24 * 0x9E3779B1u)) * 0x85EBCA6Bu) >> (32 - TLD_TAB_BITS); }

// Bloom-filter bit (before masking) of a last label of <= 8 bytes packed little-endian in lo / hi (zero padded); the
// streaming kernel's prefilter uses this two-multiply hash, longer labels use the byte-wise one (tld_hash_step).
MXY_HD uint32_t tld_hash8(uint32_t lo, uint32_t hi) {
    uint32_t h = (lo * 0x9E3779B1u) ^ (hi * 0x85EBCA6Bu);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15;   // finaliser: the low bits of a product only depend on low input bits
    return h;
}

MXY_HD uint64_t psl_hash_init() { return 0xcbf29ce484222325ULL; }
MXY_HD uint64_t psl_hash_step(uint64_t h, uint8_t b) { return (h ^ b) * 0x100000001b3ULL; }
MXY_HD uint64_t psl_hash_finish(uint64_t h) { h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ULL; h ^= h >> 32; return h; }

}  // namespace mxy
