// Bit-sliced front end of k_anchor (host + device, so that the CPU test suite can check it exhaustively).
//
// A wavefront stages 2 KiB of log per block as 8 rows of 256 bytes: lane L holds dword L of every row, w[q] = bytes
// [256 q + 4 L, 256 q + 4 L + 4) of the block. bit_transpose8() turns the lane's 8 dwords into the 8 bit planes of its 32
// bytes: afterwards bit (8 b + q) of w[c] is bit c of byte b of row q. classify_planes() evaluates the byte classes of
// the extractor (reference: BOUNDARY_LOOKUP matchy-extractor/src/lib.rs:1568-1593, DOMAIN_CHAR_LOOKUP :1597-1629) as
// boolean functions of those planes — one 32-bit operation handles 32 log positions per lane, where byte-lane SWAR on
// class bytes handled 4 — and the anchor patterns then need nothing but shifted copies of the class planes:
// "the byte k positions earlier" is v_alignbyte(P, P_of_previous_dword, 4 - k), because consecutive dwords of a row sit in
// consecutive lanes with the same bit layout.
#pragma once
#include <cstdint>

#include "hashes.h"   // MXY_HD

namespace mxy {

struct ClassPlanes {
    uint32_t B;    // boundary byte (20 byte values)
    uint32_t D;    // '0'..'9'
    uint32_t T;    // '.'
    uint32_t C;    // ':'
    uint32_t AT;   // '@'
    uint32_t NL;   // '\n'
    uint32_t LD;   // label byte: alphanumeric or >= 0x80
    uint32_t TL;   // may start the last label of a public suffix (superset: 'a'..'z' or >= 0x80; wide mode: any label byte or '-')
};

// (x & m) | (y & ~m): v_bfi_b32
MXY_HD uint32_t bsel(uint32_t m, uint32_t x, uint32_t y) { return (x & m) | (y & ~m); }
// y & ~m
MXY_HD uint32_t andn(uint32_t m, uint32_t y) { return y & ~m; }

// 8 x 8 bit-matrix transpose of the four byte lanes of w[0..7] at once (three butterfly stages, v_lshl/v_lshr + v_bfi). In two parts
// for k_anchor: the first stage reads the registers the block was loaded into and writes new ones, so the loads of the next block can be
// issued into the old ones right behind it — with the whole transpose in place the block first had to be copied out of their way
// (8 moves per block, and 4 wide ones back on the path without a next block).
#define MXY_SWAP(a, b, s, m) { const uint32_t ta = bsel(m, w[a], w[b] << s), tb = bsel(m, w[a] >> s, w[b]); w[a] = ta; w[b] = tb; }
MXY_HD void bit_transpose8_first(const uint32_t (&in)[8], uint32_t (&w)[8]) {
#pragma unroll
    for (int a = 0; a < 8; a += 2) {
        w[a] = bsel(0x55555555u, in[a], in[a + 1] << 1);
        w[a + 1] = bsel(0x55555555u, in[a] >> 1, in[a + 1]);
    }
}
MXY_HD void bit_transpose8_rest(uint32_t (&w)[8]) {
    MXY_SWAP(0, 2, 2, 0x33333333u) MXY_SWAP(1, 3, 2, 0x33333333u) MXY_SWAP(4, 6, 2, 0x33333333u) MXY_SWAP(5, 7, 2, 0x33333333u)
    MXY_SWAP(0, 4, 4, 0x0F0F0F0Fu) MXY_SWAP(1, 5, 4, 0x0F0F0F0Fu) MXY_SWAP(2, 6, 4, 0x0F0F0F0Fu) MXY_SWAP(3, 7, 4, 0x0F0F0F0Fu)
}
#undef MXY_SWAP
MXY_HD void bit_transpose8(uint32_t (&w)[8]) {
    uint32_t t[8];
    bit_transpose8_first(w, t);
    bit_transpose8_rest(t);
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = t[q];
}

// Byte classes from the bit planes p[0] (bit 0 of every byte) .. p[7]. `tl_wide`: the public-suffix list in use has a last
// label that starts with something other than 'a'..'z' / a byte >= 0x80 (never the case for the shipped list): then every
// byte that can be part of a label counts as a possible first byte.
MXY_HD ClassPlanes classify_planes(const uint32_t (&p)[8], bool tl_wide) {
    const uint32_t p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4], p5 = p[5], p6 = p[6], p7 = p[7];
    // high nibble
    const uint32_t o76 = p7 | p6;
    const uint32_t a5 = andn(o76, p5);          // 0x20..0x3F
    const uint32_t h3 = a5 & p4;                // 0x30..0x3F
    const uint32_t h2 = andn(p4, a5);           // 0x20..0x2F
    const uint32_t o54 = p5 | p4;
    const uint32_t o7654 = o76 | o54;           // clear: 0x00..0x0F
    const uint32_t b = andn(p7, p6);            // 0x40..0x7F
    const uint32_t h4 = andn(o54, b);           // 0x40..0x4F
    const uint32_t b4 = b & p4;                 // 0x50..0x5F, 0x70..0x7F
    // low nibble terms
    const uint32_t a10 = p1 & p0, o10 = p1 | p0, x10 = p1 ^ p0;
    const uint32_t a32 = p3 & p2, o32 = p3 | p2;
    const uint32_t o3210 = o32 | o10;           // clear: low nibble 0
    ClassPlanes c;
    c.T = h2 & andn(p0, a32 & p1);                          // 0x2E
    const uint32_t g9 = p3 & (p2 | p1);                     // low nibble > 9
    c.D = andn(g9, h3);                                     // 0x30..0x39
    const uint32_t h3g = h3 & g9;                           // 0x3A..0x3F
    const uint32_t loA = andn(p2 | p0, p3 & p1);            // low nibble == 0xA
    c.NL = andn(o7654, loA);                                // 0x0A
    c.C = h3 & loA;                                         // 0x3A
    c.AT = andn(o3210, h4);                                 // 0x40
    // letters: 0x41..0x4F / 0x61..0x6F (p4 clear, low nibble != 0), 0x50..0x5A / 0x70..0x7A (p4 set, low nibble <= 0xA)
    const uint32_t gA = p3 & (p2 | a10);                    // low nibble > 0xA
    const uint32_t not_letter = bsel(p4, gA, ~o3210);
    const uint32_t alpha = andn(not_letter, b);
    c.LD = alpha | c.D | p7;
    // boundary bytes: 09 0A 0D | 20 22 27 28 29 2C 2F | 3A 3B 3C 3D 3E | 40 | 5B 5D 7B 7D
    const uint32_t B0 = andn(o7654, p3 & andn(p2 & p1, x10));                  // low nibble 9, A, D
    const uint32_t f2 = bsel(p3, bsel(p2, ~x10, ~p1), bsel(p2, a10, ~p0));      // low nibble 0, 2, 7, 8, 9, C, F
    const uint32_t B2 = h2 & f2;
    const uint32_t B3 = andn(p2 & a10, h3g);                                    // 3A..3E
    const uint32_t B57 = b4 & (p3 & p0) & (p2 ^ p1);                            // low nibble B, D
    c.B = B0 | B2 | B3 | c.AT | B57;
    c.TL = tl_wide ? (c.LD | (h2 & a32 & andn(p1, p0))) : ((alpha & p5) | p7);  // wide: label byte or '-' (0x2D)
    return c;
}

// Row/lane geometry of a block
constexpr uint32_t AB_ROWS = 8, AB_ROW_BYTES = 256, AB_BLOCK = AB_ROWS * AB_ROW_BYTES;
// offset inside the block of the position that bit t of lane `lane` stands for
MXY_HD uint32_t plane_bit_offset(uint32_t lane, uint32_t t) { return ((t & 7u) << 8) + (lane << 2) + (t >> 3); }

}  // namespace mxy
