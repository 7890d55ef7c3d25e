// oracle/crypto.h — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product library.
//
// Plain restatements of the third-party primitives the reference extractor calls
// (crates/matchy-extractor/src/lib.rs:1799-1920; versions pinned in /root/reference/Cargo.lock):
//   sha2 0.10.9      SHA-256                      (FIPS 180-4)
//   tiny-keccak 2.0.2 Keccak-256 (NOT SHA3-256: padding byte 0x01)
//   bs58 0.5.1       base58, Bitcoin alphabet, whole-string big-number decode
//   bech32 0.11.1    bech32::decode (accepts Bech32 OR Bech32m checksum)
//   xxhash-rust 0.8.15 XXH64 seed 0              (crates/matchy-literal-hash/src/lib.rs:666-671)
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace orc {

// ---------------------------------------------------------------- SHA-256
inline void sha256(const uint8_t* data, size_t len, uint8_t out[32]) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
        0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
        0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    std::vector<uint8_t> msg(data, data + len);
    msg.push_back(0x80);
    while (msg.size() % 64 != 56) msg.push_back(0);
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 7; i >= 0; --i) msg.push_back((uint8_t)(bits >> (8 * i)));
    auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
    for (size_t off = 0; off < msg.size(); off += 64) {
        uint32_t w[64];
        for (int i = 0; i < 16; ++i)
            w[i] = ((uint32_t)msg[off + 4 * i] << 24) | ((uint32_t)msg[off + 4 * i + 1] << 16) |
                   ((uint32_t)msg[off + 4 * i + 2] << 8) | msg[off + 4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
            uint32_t ch = (e & f) ^ (~e & g);
            uint32_t t1 = hh + S1 + ch + K[i] + w[i];
            uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
            uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
            uint32_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    for (int i = 0; i < 8; ++i) {
        out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i];
    }
}

// ---------------------------------------------------------------- Keccak-256 (original padding 0x01)
inline void keccak_f1600(uint64_t st[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    auto rotl = [](uint64_t x, int n) { return (x << n) | (x >> (64 - n)); };
    for (int round = 0; round < 24; ++round) {
        uint64_t bc[5];
        for (int i = 0; i < 5; ++i) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; ++i) {
            uint64_t t = bc[(i + 4) % 5] ^ rotl(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        uint64_t t = st[1];
        for (int i = 0; i < 24; ++i) {
            int j = PILN[i];
            uint64_t b = st[j];
            st[j] = rotl(t, ROTC[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; ++i) bc[i] = st[j + i];
            for (int i = 0; i < 5; ++i) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[round];
    }
}

inline void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
    const size_t rate = 136;
    uint64_t st[25];
    memset(st, 0, sizeof(st));
    std::vector<uint8_t> msg(data, data + len);
    size_t padded = (len / rate + 1) * rate;
    msg.resize(padded, 0);
    msg[len] ^= 0x01;
    msg[padded - 1] ^= 0x80;
    for (size_t off = 0; off < padded; off += rate) {
        for (size_t i = 0; i < rate / 8; ++i) {
            uint64_t v = 0;
            for (int b = 7; b >= 0; --b) v = (v << 8) | msg[off + 8 * i + b];
            st[i] ^= v;
        }
        keccak_f1600(st);
    }
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 8; ++b) out[8 * i + b] = (uint8_t)(st[i] >> (8 * b));
}

// ---------------------------------------------------------------- base58 (bs58::decode(..).into_vec())
// Whole-string big-number decode with the Bitcoin alphabet; each leading '1' yields one 0x00 byte.
inline bool base58_decode(const uint8_t* s, size_t n, std::vector<uint8_t>& out) {
    static const char* ALPHA = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz";
    int8_t map[256];
    memset(map, -1, sizeof(map));
    for (int i = 0; i < 58; ++i) map[(uint8_t)ALPHA[i]] = (int8_t)i;
    std::vector<uint8_t> num;  // little-endian base-256 digits
    size_t zeros = 0;
    bool leading = true;
    for (size_t i = 0; i < n; ++i) {
        int v = map[s[i]];
        if (v < 0) return false;
        if (leading && v == 0) { ++zeros; continue; }
        leading = false;
        uint32_t carry = (uint32_t)v;
        for (size_t j = 0; j < num.size(); ++j) {
            carry += (uint32_t)num[j] * 58u;
            num[j] = (uint8_t)carry;
            carry >>= 8;
        }
        while (carry) { num.push_back((uint8_t)carry); carry >>= 8; }
    }
    out.assign(zeros, 0);
    for (size_t i = num.size(); i-- > 0;) out.push_back(num[i]);
    return true;
}

// ---------------------------------------------------------------- bech32::decode → hrp == "bc"
// Restates bech32 0.11.1 `decode()` as used at matchy-extractor/src/lib.rs:1825-1835 for tokens that
// already start with "bc1": separator = LAST '1'; every char right of it must be a bech32 symbol;
// no mixed case across the whole string; data part >= 6 symbols; checksum residue 1 (Bech32) or
// 0x2bc830a3 (Bech32m). Returns true iff decode succeeds AND the hrp equals "bc".
inline bool bech32_decode_is_bc(const uint8_t* s, size_t n) {
    static const char* CHARSET = "qpzry9x8gf2tvdw0s3jn54khce6mua7l";
    int8_t map[128];
    memset(map, -1, sizeof(map));
    for (int i = 0; i < 32; ++i) {
        map[(uint8_t)CHARSET[i]] = (int8_t)i;
        char c = CHARSET[i];
        if (c >= 'a' && c <= 'z') map[(uint8_t)(c - 32)] = (int8_t)i;
    }
    bool has_upper = false, has_lower = false;
    size_t sep = (size_t)-1;
    for (size_t k = n; k-- > 0;) {
        uint8_t ch = s[k];
        if (ch == '1' && sep == (size_t)-1) {
            sep = k;
        } else if (sep == (size_t)-1) {
            if (ch >= 128 || map[ch] < 0) return false;  // InvalidChar
        }
        if (ch >= 'A' && ch <= 'Z') has_upper = true;
        else if (ch >= 'a' && ch <= 'z') has_lower = true;
    }
    if (has_upper && has_lower) return false;
    if (sep == (size_t)-1) return false;
    // Hrp::parse: 1..=83 chars, each 33..=126
    if (sep == 0 || sep > 83) return false;
    for (size_t k = 0; k < sep; ++k)
        if (s[k] < 33 || s[k] > 126) return false;
    size_t dlen = n - sep - 1;
    if (dlen < 6) return false;
    if (n > 1023) return false;
    auto polymod_step = [](uint32_t chk, uint32_t v) {
        static const uint32_t GEN[5] = {0x3b6a57b2, 0x26508e6d, 0x1ea119fa, 0x3d4233dd, 0x2a1462b3};
        uint32_t top = chk >> 25;
        chk = ((chk & 0x1ffffff) << 5) ^ v;
        for (int i = 0; i < 5; ++i)
            if ((top >> i) & 1) chk ^= GEN[i];
        return chk;
    };
    uint32_t chk = 1;
    auto lower = [](uint8_t c) { return (uint8_t)((c >= 'A' && c <= 'Z') ? c + 32 : c); };
    for (size_t k = 0; k < sep; ++k) chk = polymod_step(chk, lower(s[k]) >> 5);
    chk = polymod_step(chk, 0);
    for (size_t k = 0; k < sep; ++k) chk = polymod_step(chk, lower(s[k]) & 31);
    for (size_t k = sep + 1; k < n; ++k) chk = polymod_step(chk, (uint32_t)map[s[k]]);
    if (chk != 1 && chk != 0x2bc830a3u) return false;
    // hrp == Hrp::parse("bc") (case-insensitive compare)
    return sep == 2 && lower(s[0]) == 'b' && lower(s[1]) == 'c';
}

// ---------------------------------------------------------------- XXH64
inline uint64_t xxh64(const uint8_t* p, size_t len, uint64_t seed) {
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL,
                   P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto rd64 = [](const uint8_t* q) { uint64_t v; memcpy(&v, q, 8); return v; };
    auto rd32 = [](const uint8_t* q) { uint32_t v; memcpy(&v, q, 4); return v; };
    auto round = [&](uint64_t acc, uint64_t in) { acc += in * P2; acc = rotl(acc, 31); return acc * P1; };
    auto merge = [&](uint64_t acc, uint64_t val) { val = round(0, val); acc ^= val; return acc * P1 + P4; };
    const uint8_t* end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t* limit = end - 32;
        do {
            v1 = round(v1, rd64(p)); v2 = round(v2, rd64(p + 8));
            v3 = round(v3, rd64(p + 16)); v4 = round(v4, rd64(p + 24));
            p += 32;
        } while (p <= limit);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
        h = merge(h, v1); h = merge(h, v2); h = merge(h, v3); h = merge(h, v4);
    } else {
        h = seed + P5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= round(0, rd64(p)); h = rotl(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = rotl(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (*p) * P5; h = rotl(h, 11) * P1; ++p; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// rustc-hash 2.1.1 FxHasher over one u32 then finish() — used for ACLH slot placement
// (crates/matchy-paraglob/src/literal_hash.rs:95-99). 64-bit constants: K = 0xf1357aea2e62a9c5,
// finish = rotate_left(26).  [UNVERIFIED against the crate source, which is not vendored; see DESIGN.md]
inline uint64_t fxhash_u32(uint32_t v) {
    uint64_t h = (0 + (uint64_t)v) * 0xf1357aea2e62a9c5ULL;
    return (h << 26) | (h >> 38);
}

}  // namespace orc
