// Device-side helpers shared by the HIP kernels (byte classes, wave helpers, PSL probes).
#pragma once
#include <hip/hip_runtime.h>

#include "hashes.h"
#include "scan_types.h"

namespace mxy {

// ------------------------------------------------------------------------------------------------ byte classes
constexpr uint32_t C_B = 1, C_DIG = 2, C_DOT = 4, C_COLON = 8, C_AT = 16, C_LD = 32, C_NL = 64, C_TLD1 = 128;

// BOUNDARY_LOOKUP (ext:1568-1593) as a 128-bit bitmap
constexpr uint64_t bnd_lo() {
    uint64_t m = 0;
    const int cs[] = {0x09, 0x0a, 0x0d, 0x20, 0x22, 0x27, 0x28, 0x29, 0x2c, 0x2f, 0x3a, 0x3b, 0x3c, 0x3d, 0x3e};
    for (int c : cs) m |= 1ull << c;
    return m;
}
constexpr uint64_t bnd_hi() {
    uint64_t m = 0;
    const int cs[] = {0x40, 0x5b, 0x5d, 0x7b, 0x7d};
    for (int c : cs) m |= 1ull << (c - 64);
    return m;
}
__device__ __forceinline__ bool d_is_boundary(uint32_t b) {
    if (b >= 128) return false;
    uint64_t m = b < 64 ? bnd_lo() : bnd_hi();
    return (m >> (b & 63)) & 1;
}
__device__ __forceinline__ bool d_is_digit(uint32_t b) { return b - '0' < 10u; }
__device__ __forceinline__ bool d_is_alpha(uint32_t b) { return (b | 0x20) - 'a' < 26u; }
__device__ __forceinline__ bool d_is_alnum(uint32_t b) { return d_is_digit(b) || d_is_alpha(b); }
__device__ __forceinline__ bool d_is_hex(uint32_t b) { return d_is_digit(b) || ((b | 0x20) - 'a' < 6u); }
__device__ __forceinline__ bool d_is_domain_char_fast(uint32_t b) { return d_is_alnum(b) || b == '-' || b == '.' || b >= 0x80; }  // ext:1597-1629
__device__ __forceinline__ bool d_is_domain_char(uint32_t b) { return d_is_alnum(b) || b == '-' || b == '.'; }                      // ext:1639
__device__ __forceinline__ bool d_is_email_local(uint32_t b) { return d_is_alnum(b) || b == '.' || b == '-' || b == '_' || b == '+'; }  // ext:1644

__device__ __forceinline__ uint32_t class_of(uint32_t b) {
    uint32_t c = 0;
    if (d_is_boundary(b)) c |= C_B;
    if (d_is_digit(b)) c |= C_DIG;
    if (b == '.') c |= C_DOT;
    if (b == ':') c |= C_COLON;
    if (b == '@') c |= C_AT;
    if (d_is_alnum(b) || b >= 0x80) c |= C_LD;
    if (b == '\n') c |= C_NL;
    return c;
}

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }
// base + number of set bits of m below this lane
__device__ __forceinline__ uint32_t mbcnt64_add(uint64_t m, uint32_t base) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base));
}
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Wave-private chunked list writer. A wave reserves CHUNK slots of a global list with ONE atomic and fills them with
// ballot/mbcnt-compacted appends; slots it never fills are set to `sentinel` so readers can skip them. This keeps
// the number of atomics on the shared list counter ~CHUNK/64 times lower than one atomic per append.
template <class T, uint32_t CHUNK>
struct ChunkWriter {
    static_assert(CHUNK >= 64, "one append can carry 64 entries");
    uint32_t base = 0xFFFFFFFFu;  // wave-uniform; 0xFFFFFFFF = no chunk yet
    uint32_t used = 0;
    uint32_t total = 0;           // entries really appended by this wave

    __device__ __forceinline__ void pad_rest(T* out, uint32_t cap, const T& sentinel) {
        if (base == 0xFFFFFFFFu) return;
        for (uint32_t k = used + lane_id(); k < CHUNK; k += 64)
            if (base + k < cap) out[base + k] = sentinel;
        used = CHUNK;
    }
    // All lanes of the (converged) wave call this; lanes with emit=true store `v`.
    __device__ __forceinline__ void append(bool emit, const T& v, T* out, uint32_t cap, uint32_t* counter, const T& sentinel) {
        const uint64_t m = __ballot(emit);
        if (m == 0) return;
        const uint32_t n = (uint32_t)__popcll(m);
        if (base == 0xFFFFFFFFu || used + n > CHUNK) {
            pad_rest(out, cap, sentinel);
            uint32_t b = 0;
            if (lane_id() == 0) b = atomicAdd(counter, CHUNK);
            base = __builtin_amdgcn_readfirstlane(b);
            used = 0;
        }
        if (emit) {
            const uint32_t slot = base + used + (uint32_t)__popcll(m & lanemask_lt());
            if (slot < cap) out[slot] = v;
        }
        used += n;
        total += n;
    }
};

// What an unused slot of a reserved chunk holds: every consumer of these lists skips it (Candidate: len_type 0xFFFFFFFF; anchors:
// kind 0xFF; work-list indices: 0xFFFFFFFF).
template <class T> __device__ __forceinline__ T list_sentinel();
template <> __device__ __forceinline__ Candidate list_sentinel<Candidate>() { return Candidate{0u, 0xFFFFFFFFu, 0u, 0u}; }
template <> __device__ __forceinline__ RareAnchor list_sentinel<RareAnchor>() { return RareAnchor{0xFFFFFFFFu, 0xFFu}; }
template <> __device__ __forceinline__ uint32_t list_sentinel<uint32_t>() { return 0xFFFFFFFFu; }
template <> __device__ __forceinline__ uint2 list_sentinel<uint2>() { return make_uint2(0xFFFFFFFFu, 0xFFu); }

// Writer for the sparse lists (rare anchors, long tokens, heavy tokens, prefiltered candidates): entries collect in a 64-slot LDS buffer of the wave
// and leave together — one atomic per flush for exactly the entries there are, so these lists carry no per-wave padding
// (with ~10 000 producer waves a 64-slot chunk per wave was mostly padding, which the consumers dragged through their
// loops) and appends that carry one or two entries do not pay an atomic round trip each.
// ... WHILE THEY ARE SPARSE. On input of another shape the same list is dense (two file hashes per log line: 18 M tokens, every one a
// candidate), and one returning atomic per 64 entries from thousands of waves on ONE counter — atomics on one line are served one after
// the other, ~12 ns each — was the whole kernel (278 K atomics = 3.3 of k_validate's 3.6 ms, profiles/r04_log_shapes.txt). So the
// reservation grows with the wave's own history: its first four flushes reserve what they carry, the next twelve a chunk of 256 slots,
// later ones a chunk of 2048 that the following flushes fill without an atomic; what a wave leaves unused of its last chunk is set to
// list_sentinel<T>() by flush() — the call at the end of a wave's work.
template <class T, uint32_t CAP = 64>
struct BufferedWriter {
    T* buf;              // CAP entries of LDS owned by this wave
    uint32_t cnt = 0;    // wave-uniform
    uint32_t total = 0;  // entries appended by this wave
    uint32_t base = 0xFFFFFFFFu, used = 0, ccap = 0, flushes = 0;   // the wave's current chunk (wave-uniform)
    __device__ __forceinline__ explicit BufferedWriter(T* lds) : buf(lds) {}
    __device__ __forceinline__ void pad_rest(T* out, uint32_t cap) {
        if (base == 0xFFFFFFFFu) return;
        for (uint32_t k = used + lane_id(); k < ccap; k += 64) if (base + k < cap) out[base + k] = list_sentinel<T>();
        used = ccap;
    }
    // slots for n entries: from the current chunk, or from a new reservation; returns the first slot
    __device__ __forceinline__ uint32_t reserve(uint32_t n, T* out, uint32_t cap, uint32_t* counter) {
        if (base == 0xFFFFFFFFu || used + n > ccap) {
            pad_rest(out, cap);
            const uint32_t chunk = flushes < 4 ? n : (flushes < 16 ? 256u : 2048u);
            ccap = chunk < n ? n : chunk;
            uint32_t b = 0;
            if (lane_id() == 0) b = atomicAdd(counter, ccap);
            base = __builtin_amdgcn_readfirstlane(b);
            used = 0;
        }
        ++flushes;
        const uint32_t at = base + used;
        used += n;
        return at;
    }
    // the staged entries leave (called by append when the stage is full)
    template <class F>
    __device__ __forceinline__ void drain_then(T* out, uint32_t cap, uint32_t* counter, F&& after) {
        if (cnt == 0) return;
        const uint32_t b = reserve(cnt, out, cap, counter);
        __builtin_amdgcn_wave_barrier();
        if (lane_id() < cnt && b + lane_id() < cap) out[b + lane_id()] = buf[lane_id()];
        after(b, cnt);
        __builtin_amdgcn_wave_barrier();
        cnt = 0;
    }
    // END of the wave's work on this list: staged entries out, the unused rest of its last chunk marked
    __device__ __forceinline__ void flush(T* out, uint32_t cap, uint32_t* counter) {
        flush_then(out, cap, counter, [](uint32_t, uint32_t) {});
    }
    // `after(b, n)` is called by the whole wave once the n staged entries have been written to out[b, b + n) (they are still in buf)
    template <class F>
    __device__ __forceinline__ void flush_then(T* out, uint32_t cap, uint32_t* counter, F&& after) {
        drain_then(out, cap, counter, after);
        pad_rest(out, cap);
    }
    // all lanes of the (converged) wave call this
    __device__ __forceinline__ void append(bool emit, const T& v, T* out, uint32_t cap, uint32_t* counter) {
        const uint64_t m = __ballot(emit);
        if (m == 0) return;
        const uint32_t n = (uint32_t)__popcll(m);
        const uint32_t rank = (uint32_t)__popcll(m & lanemask_lt());
        if (CAP < 64 && n > CAP) {
            // more entries than the staging buffer holds (dense phases): they leave directly, through the same chunks
            const uint32_t b = reserve(n, out, cap, counter);
            if (emit && b + rank < cap) out[b + rank] = v;
            total += n;
            return;
        }
        if (cnt + n > CAP) drain_then(out, cap, counter, [](uint32_t, uint32_t) {});
        if (emit) buf[cnt + rank] = v;
        cnt += n;
        total += n;
    }
    // append for a full-width stage (CAP = 64) whose flushes call `after` (flush_then)
    template <class F>
    __device__ __forceinline__ void append_then(bool emit, const T& v, T* out, uint32_t cap, uint32_t* counter, F&& after) {
        static_assert(CAP >= 64, "append_then: the stage holds a whole wave's entries");
        const uint64_t m = __ballot(emit);
        if (m == 0) return;
        const uint32_t n = (uint32_t)__popcll(m);
        const uint32_t rank = (uint32_t)__popcll(m & lanemask_lt());
        if (cnt + n > CAP) drain_then(out, cap, counter, after);
        if (emit) buf[cnt + rank] = v;
        cnt += n;
        total += n;
    }
};

// Writer for a list that may be sparse or dense (k_anchor's IPv4 candidates: a few per block when the database's /24 bitmap
// filters, one per log line when it does not): the wave reserves chunks of `chunk` slots of the global list (one returning
// atomic per chunk — atomics on one counter are served one at a time, ~12 ns each, so 4096 waves cannot afford one per
// append), small appends collect in a CAP-entry LDS buffer and leave together as one coalesced store, large ones go
// straight into the chunk. Unused slots of a chunk hold `sentinel`.
template <class T, uint32_t CAP>
struct StagedChunkWriter {
    T* buf;                        // CAP entries of LDS owned by this wave
    uint32_t cnt = 0;              // staged entries (wave-uniform)
    uint32_t base = 0xFFFFFFFFu;   // current chunk (0xFFFFFFFF: none yet), wave-uniform
    uint32_t used = 0, chunk;
    __device__ __forceinline__ StagedChunkWriter(T* lds, uint32_t chunk_slots) : buf(lds), chunk(chunk_slots < 64 ? 64u : chunk_slots) {}
    __device__ __forceinline__ void pad_rest(T* out, uint32_t cap, const T& sentinel) {
        if (base == 0xFFFFFFFFu) return;
        for (uint32_t k = used + lane_id(); k < chunk; k += 64)
            if (base + k < cap) out[base + k] = sentinel;
        used = chunk;
    }
    // room for n more entries in the current chunk, else a new chunk
    __device__ __forceinline__ void need(uint32_t n, T* out, uint32_t cap, uint32_t* counter, const T& sentinel) {
        if (base != 0xFFFFFFFFu && used + n <= chunk) return;
        pad_rest(out, cap, sentinel);
        uint32_t b = 0;
        if (lane_id() == 0) b = atomicAdd(counter, chunk);
        base = __builtin_amdgcn_readfirstlane(b);
        used = 0;
    }
    __device__ __forceinline__ void flush(T* out, uint32_t cap, uint32_t* counter, const T& sentinel) {
        if (cnt == 0) return;
        need(cnt, out, cap, counter, sentinel);
        __builtin_amdgcn_wave_barrier();
        if (lane_id() < cnt && base + used + lane_id() < cap) out[base + used + lane_id()] = buf[lane_id()];
        __builtin_amdgcn_wave_barrier();
        used += cnt;
        cnt = 0;
    }
    // all lanes of the (converged) wave call this
    __device__ __forceinline__ void append(bool emit, const T& v, T* out, uint32_t cap, uint32_t* counter, const T& sentinel) {
        const uint64_t m = __ballot(emit);
        if (m == 0) return;
        const uint32_t n = (uint32_t)__popcll(m);
        const uint32_t rank = (uint32_t)__popcll(m & lanemask_lt());
        if (n > CAP / 2) {
            flush(out, cap, counter, sentinel);   // keeps the list in append order per wave
            need(n, out, cap, counter, sentinel);
            if (emit && base + used + rank < cap) out[base + used + rank] = v;
            used += n;
            return;
        }
        if (cnt + n > CAP) flush(out, cap, counter, sentinel);
        if (emit) buf[cnt + rank] = v;
        cnt += n;
    }
    __device__ __forceinline__ void finish(T* out, uint32_t cap, uint32_t* counter, const T& sentinel) {
        flush(out, cap, counter, sentinel);
        pad_rest(out, cap, sentinel);
    }
};

// Chunked writer for the plane-organised domain anchor list (TokParams::dom_list): slots are reserved like in
// ChunkWriter (one atomic per ANCHOR_CHUNK slots), the caller stores the planes of its slot.
struct DomWriter {
    uint32_t next = 0xFFFFFFFFu, left = 0;   // the wave's current chunk: next free slot, slots left in it (0xFFFFFFFF: no chunk yet)
    __device__ __forceinline__ void pad_rest(uint32_t* out, uint32_t cap) {
        if (next == 0xFFFFFFFFu) return;
        for (uint32_t k = lane_id(); k < left; k += 64)
            if (next + k < cap) out[dom_plane_index(next + k, 0)] = 0xFFFFFFFFu;
        next += left;
        left = 0;
    }
    // all lanes of the converged wave call this; returns the lane's slot, or 0xFFFFFFFF (not emitting / list full).
    // counter_fn() yields the list counter: it is called only when a new chunk is needed, so the caller can fetch the pointer there instead
    // of holding it in registers; chunk_fn() the size of such a chunk (a multiple of DOM_TILE, at most ANCHOR_CHUNK).
    // static_fn() yields the first slot of the ANCHOR_CHUNK slots this wave owns WITHOUT a reservation, or 0xFFFFFFFF: every wave of k_anchor needs
    // its first chunk in its first blocks, all of them at once — 4096 returning atomics on one counter line, served one after the other while every
    // wave waits for its own (6 % of k_anchor on a web-server log, measured by handing the first chunks out without the atomic). When the previous
    // batch was dense the host presets the counter to waves x ANCHOR_CHUNK (TokParams::dom_static) and wave w starts in chunk w. The later
    // reservations of a wave are spread over the kernel and cost nothing measurable, so they can be small (TokParams::dom_chunk: an eighth of a
    // wave's share of the previous batch): what a wave leaves unused of its last chunk are slots k_validate_dom reads for nothing.
    template <class F, class G, class H>
    __device__ __forceinline__ uint32_t reserve(bool emit, uint32_t* out, uint32_t cap, F counter_fn, G static_fn, H chunk_fn) {
        const uint64_t m = __ballot(emit);
        if (m == 0) return 0xFFFFFFFFu;
        const uint32_t n = (uint32_t)__popcll(m);
        if (left < n) {
            const uint32_t own = next == 0xFFFFFFFFu ? static_fn() : 0xFFFFFFFFu;
            pad_rest(out, cap);
            if (own != 0xFFFFFFFFu) {
                next = own;
                left = ANCHOR_CHUNK;
            } else {
                const uint32_t c = chunk_fn();
                uint32_t b = 0;
                uint32_t* const counter = counter_fn();
                if (lane_id() == 0) b = atomicAdd(counter, c);
                next = __builtin_amdgcn_readfirstlane(b);
                left = c;
            }
        }
        uint32_t slot = 0xFFFFFFFFu;
        if (emit) {
            slot = next + (uint32_t)__popcll(m & lanemask_lt());
            if (slot >= cap) slot = 0xFFFFFFFFu;
        }
        next += n;
        left -= n;
        return slot;
    }
};

struct LogView {
    const uint8_t* p;
    uint32_t len;
    __device__ __forceinline__ uint32_t at(uint32_t i) const { return p[i]; }
};

// Rust core::str::from_utf8 acceptance
__device__ inline bool d_valid_utf8(const uint8_t* s, uint32_t n) {
    uint32_t i = 0;
    while (i < n) {
        uint32_t c = s[i];
        if (c < 0x80) { ++i; continue; }
        uint32_t l = (c >= 0xC2 && c <= 0xDF) ? 2 : (c >= 0xE0 && c <= 0xEF) ? 3 : (c >= 0xF0 && c <= 0xF4) ? 4 : 0;
        if (l == 0 || i + l > n) return false;
        uint32_t c1 = s[i + 1];
        uint32_t lo = 0x80, hi = 0xBF;
        if (c == 0xE0) lo = 0xA0;
        if (c == 0xED) hi = 0x9F;
        if (c == 0xF0) lo = 0x90;
        if (c == 0xF4) hi = 0x8F;
        if (c1 < lo || c1 > hi) return false;
        for (uint32_t k = 2; k < l; ++k) if ((s[i + k] & 0xC0) != 0x80) return false;
        i += l;
    }
    return true;
}

// Byte-class masks of 8 log bytes at once (SWAR; bit 7 of each byte of the result is the class bit, all other bits 0).
// Every addend keeps each byte below 0x100, so no carry crosses a byte.
struct ByteMasks { uint64_t dc, dot, dash, high; };
// The class arithmetic keeps every byte lane below 0x100, so no carry ever crosses a byte — nor the middle of a 64-bit word: the two
// halves are computed on their own with 32-bit adds. (Written on uint64_t the compiler emits v_add_co / v_addc_co pairs, a carry chain
// through VCC that costs twice as much on the vector pipe; it also fuses a 64-bit expression that merely spells the two halves back
// into that form, hence the separate function.)
struct ByteMasks32 { uint32_t dc, dot, dash, high; };
__device__ __forceinline__ ByteMasks32 domain_masks32(uint32_t x) {
    constexpr uint32_t H = 0x80808080u, L7 = 0x7F7F7F7Fu;
    const uint32_t t = x & L7, l = t | 0x20202020u;
    const uint32_t dig = (t + 0x50505050u) & ~(t + 0x46464646u);   // '0'..'9'
    const uint32_t alp = (l + 0x1F1F1F1Fu) & ~(l + 0x05050505u);   // 'a'..'z' after case folding
    const uint32_t ndot = (t ^ 0x2E2E2E2Eu) + L7, ndash = (t ^ 0x2D2D2D2Du) + L7;  // bit 7 set iff different
    ByteMasks32 m;
    m.high = x & H;
    m.dot = ~ndot & ~x & H;
    m.dash = ~ndash & ~x & H;
    m.dc = (((dig | alp) & ~x) | m.dot | m.dash | m.high) & H;  // DOMAIN_CHAR_LOOKUP (ext:1597-1629)
    return m;
}
__device__ __forceinline__ ByteMasks domain_masks(uint64_t x) {
    const ByteMasks32 lo = domain_masks32((uint32_t)x), hi = domain_masks32((uint32_t)(x >> 32));
    ByteMasks m;
    m.dc = (uint64_t)lo.dc | ((uint64_t)hi.dc << 32);
    m.dot = (uint64_t)lo.dot | ((uint64_t)hi.dot << 32);
    m.dash = (uint64_t)lo.dash | ((uint64_t)hi.dash << 32);
    m.high = (uint64_t)lo.high | ((uint64_t)hi.high << 32);
    return m;
}

// ------------------------------------------------------------------------------------------------ IPv4
// IPv4 dotted-quad rules (ext:813-869, 1120-1179) on a 20-byte window held in registers: q = bytes [dot-4, dot+12),
// t4 = bytes [dot+12, dot+16), `dot` = position of the first dot of the run. Emits iff the maximal [0-9.] run around
// the dot is a valid dotted quad delimited by boundaries (N1).
__device__ __forceinline__ bool d_ipv4_from_window(uint4 q, uint32_t t4, uint32_t dot, uint32_t& start, uint32_t& end, uint32_t& addr) {
    // first octet: digits at dot-1, dot-2, dot-3 (right to left), then a boundary
    const uint32_t b3 = q.x >> 24, b2 = (q.x >> 16) & 0xFF, b1 = (q.x >> 8) & 0xFF, b0 = q.x & 0xFF;
    if (!d_is_digit(b3)) return false;
    const bool g2 = d_is_digit(b2), g1 = g2 && d_is_digit(b1);
    const uint32_t n1 = 1u + g2 + g1;
    const uint32_t before = g1 ? b0 : g2 ? b1 : b2;
    if (!d_is_boundary(before)) return false;
    const uint32_t first = g1 ? b1 : g2 ? b2 : b3;
    uint32_t a = g1 ? (b1 - 48) * 100 + (b2 - 48) * 10 + (b3 - 48) : g2 ? (b2 - 48) * 10 + (b3 - 48) : (b3 - 48);
    if (a > 255 || (n1 > 1 && first == '0')) return false;
    // r0..r2 = the 12 bytes after the dot
    uint32_t r0 = __builtin_amdgcn_alignbyte(q.z, q.y, 1), r1 = __builtin_amdgcn_alignbyte(q.w, q.z, 1), r2 = __builtin_amdgcn_alignbyte(t4, q.w, 1);
    uint32_t pos = dot + 1;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const uint32_t c0 = r0 & 0xFF, c1 = (r0 >> 8) & 0xFF, c2 = (r0 >> 16) & 0xFF, c3 = r0 >> 24;
        if (!d_is_digit(c0)) return false;
        const bool h1 = d_is_digit(c1), h2 = h1 && d_is_digit(c2);
        const uint32_t n = 1u + h1 + h2;
        const uint32_t v = h2 ? (c0 - 48) * 100 + (c1 - 48) * 10 + (c2 - 48) : h1 ? (c0 - 48) * 10 + (c1 - 48) : (c0 - 48);
        if (v > 255 || (h1 && c0 == '0')) return false;
        const uint32_t sep = h2 ? c3 : h1 ? c2 : c1;
        a = (a << 8) | v;
        pos += n;
        if (o < 2) {
            if (sep != '.') return false;
            pos += 1;
            // drop n + 1 bytes: first n (1..3), then one more
            r0 = __builtin_amdgcn_alignbyte(r1, r0, n); r1 = __builtin_amdgcn_alignbyte(r2, r1, n); r2 = __builtin_amdgcn_alignbyte(0u, r2, n);
            r0 = __builtin_amdgcn_alignbyte(r1, r0, 1); r1 = __builtin_amdgcn_alignbyte(r2, r1, 1); r2 = r2 >> 8;
        } else if (!d_is_boundary(sep)) {
            return false;
        }
    }
    start = dot - n1; end = pos; addr = a;
    return true;
}
// The same with two wide global loads; callers make sure [dot-4, dot+16) lies inside the buffer.
__device__ __forceinline__ bool val_ipv4_fast(const uint8_t* log, uint32_t dot, uint32_t& start, uint32_t& end, uint32_t& addr) {
    uint4 q;
    uint32_t t4;
    __builtin_memcpy(&q, log + dot - 4, 16);   // unaligned global_load_dwordx4 (unaligned access mode is on for gfx9+)
    __builtin_memcpy(&t4, log + dot + 12, 4);
    return d_ipv4_from_window(q, t4, dot, start, end, addr);
}

// IPv4 (ext:813-869, 1120-1179). `dot` is the first dot of a maximal [0-9.] run whose first octet has 1-3 digits and
// is preceded by a boundary or the buffer start (anchor rule). The whole run must parse as a dotted quad and be
// followed by a boundary or the end of the buffer.
__device__ inline bool val_ipv4(const LogView& lg, uint32_t dot, uint32_t& start, uint32_t& end, uint32_t& addr) {
    uint32_t s = dot;
    while (s > 0 && dot - s < 3 && d_is_digit(lg.at(s - 1))) --s;
    if (s == dot) return false;
    if (s > 0 && !d_is_boundary(lg.at(s - 1))) return false;
    uint32_t pos = s, a = 0;
    for (int idx = 0; idx < 4; ++idx) {
        uint32_t v = 0, digits = 0, first = 0;
        while (pos < lg.len && digits < 3) {
            uint32_t c = lg.at(pos);
            if (!d_is_digit(c)) break;
            if (digits == 0) first = c;
            v = v * 10 + (c - '0');
            ++pos;
            ++digits;
        }
        if (digits == 0 || v > 255) return false;
        if (digits > 1 && first == '0') return false;
        a = (a << 8) | v;
        if (idx < 3) {
            if (pos >= lg.len || lg.at(pos) != '.') return false;
            ++pos;
        }
    }
    if (pos < lg.len && !d_is_boundary(lg.at(pos))) return false;
    start = s; end = pos; addr = a;
    return true;
}

// ------------------------------------------------------------------------------------------------ PSL
__device__ inline bool psl_contains(const DevDb& db, uint64_t h, const uint8_t* s, uint32_t n) {
    uint32_t slot = (uint32_t)h & db.psl_mask;
    for (;;) {
        PslSlot e = db.psl_slots[slot];
        if (e.len == 0) return false;
        if (e.hash == h && e.len == n) {
            const uint8_t* q = db.psl_pool + e.off;
            bool eq = true;
            for (uint32_t k = 0; k < n; ++k) if (q[k] != s[k]) { eq = false; break; }
            if (eq) return true;
        }
        slot = (slot + 1) & db.psl_mask;
    }
}
// find_valid_tld_suffix_bytes(..).is_some() (ext:1671-1692) over log[lo,hi): dots right-to-left, hash grows leftwards.
__device__ inline bool psl_suffix_exists(const DevDb& db, const uint8_t* log, uint32_t lo, uint32_t hi) {
    uint64_t rh = psl_hash_init();
    for (uint32_t q = hi; q-- > lo;) {
        uint32_t c = log[q];
        if (hi - q - 1 > db.max_suffix_len) return false;   // longer than every suffix in the list
        if (c == '.') {
            if (psl_contains(db, psl_hash_finish(rh), log + q + 1, hi - q - 1)) return true;
        }
        rh = psl_hash_step(rh, (uint8_t)c);
    }
    return false;
}
__device__ __forceinline__ uint32_t tld_hash_step(uint32_t h, uint32_t c) { return (h ^ c) * 16777619u; }
__device__ __forceinline__ uint32_t tld_hash_bit(uint32_t h) { return (h ^ (h >> 15)) & (TLD_BLOOM_BITS - 1); }

}  // namespace mxy
