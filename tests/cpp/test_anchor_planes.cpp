// CPU check of the bit-sliced front end of k_anchor (matchy_amd/csrc/anchor_planes.h): the bit transpose and every class
// plane against the byte-wise class definitions, for all 256 byte values in every lane/bit slot.
// Build: g++ -O1 -std=c++17 -I matchy_amd/csrc tests/cpp/test_anchor_planes.cpp -o /tmp/test_anchor_planes
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "anchor_planes.h"

using namespace mxy;

// the byte classes as the extractor defines them (matchy-extractor/src/lib.rs:1568-1593 boundary set, :1597-1629 domain chars)
static bool is_boundary(unsigned b) {
    static const unsigned char cs[] = {0x09, 0x0a, 0x0d, 0x20, 0x22, 0x27, 0x28, 0x29, 0x2c, 0x2f, 0x3a, 0x3b, 0x3c, 0x3d, 0x3e, 0x40, 0x5b, 0x5d, 0x7b, 0x7d};
    for (unsigned char c : cs) if (c == b) return true;
    return false;
}
static bool is_digit(unsigned b) { return b >= '0' && b <= '9'; }
static bool is_alpha(unsigned b) { return (b >= 'a' && b <= 'z') || (b >= 'A' && b <= 'Z'); }

int main() {
    int bad = 0;
    std::mt19937 rng(12345);
    for (int round = 0; round < 4000; ++round) {
        uint8_t bytes[32];   // [row q][byte b]
        for (int i = 0; i < 32; ++i) bytes[i] = round < 256 ? (uint8_t)((round + i * 37) & 0xFF) : (uint8_t)(rng() & 0xFF);
        if (round >= 256 && round < 1200) for (int i = 0; i < 32; ++i) bytes[i] = (uint8_t)(0x20 + (rng() % 0x5F));   // printable ASCII
        uint32_t w[8];
        for (int q = 0; q < 8; ++q) memcpy(&w[q], bytes + 4 * q, 4);
        bit_transpose8(w);
        for (int c = 0; c < 8; ++c)
            for (int q = 0; q < 8; ++q)
                for (int b = 0; b < 4; ++b) {
                    const unsigned want = (bytes[4 * q + b] >> c) & 1, got = (w[c] >> (8 * b + q)) & 1;
                    if (want != got) { if (bad++ < 10) printf("transpose mismatch round %d c %d q %d b %d\n", round, c, q, b); }
                }
        for (int wide = 0; wide < 2; ++wide) {
            const ClassPlanes cp = classify_planes(w, wide != 0);
            for (int q = 0; q < 8; ++q)
                for (int b = 0; b < 4; ++b) {
                    const unsigned x = bytes[4 * q + b], t = 8 * b + q;
                    auto chk = [&](const char* name, uint32_t plane, bool want) {
                        if (((plane >> t) & 1) != (want ? 1u : 0u)) { if (bad++ < 20) printf("class %s mismatch for byte 0x%02x (wide %d)\n", name, x, wide); }
                    };
                    chk("B", cp.B, is_boundary(x));
                    chk("D", cp.D, is_digit(x));
                    chk("T", cp.T, x == '.');
                    chk("C", cp.C, x == ':');
                    chk("AT", cp.AT, x == '@');
                    chk("NL", cp.NL, x == '\n');
                    chk("LD", cp.LD, is_digit(x) || is_alpha(x) || x >= 0x80);
                    chk("TL", cp.TL, wide ? (is_digit(x) || is_alpha(x) || x >= 0x80 || x == '-') : ((x >= 'a' && x <= 'z') || x >= 0x80));
                }
        }
    }
    // geometry helper
    for (uint32_t lane = 0; lane < 64; ++lane)
        for (uint32_t t = 0; t < 32; ++t)
            if (plane_bit_offset(lane, t) != (t % 8) * 256 + lane * 4 + t / 8) ++bad;
    if (bad) { printf("FAILED: %d mismatches\n", bad); return 1; }
    printf("anchor_planes ok\n");
    return 0;
}
