// Types shared between the host pipeline and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mxy {

// MATCHY_ITEM_TYPE_* (reference: crates/matchy/include/matchy/matchy.h:233-288)
enum ItemType : uint8_t {
    IT_DOMAIN = 0, IT_EMAIL = 1, IT_IPV4 = 2, IT_IPV6 = 3, IT_MD5 = 4, IT_SHA1 = 5, IT_SHA256 = 6,
    IT_SHA384 = 7, IT_SHA512 = 8, IT_BITCOIN = 9, IT_ETHEREUM = 10, IT_MONERO = 11, IT_COUNT = 12
};
// MATCHY_EXTRACT_* (matchy.h:188-228)
enum ExtractFlags : uint32_t {
    EX_DOMAINS = 1, EX_EMAILS = 2, EX_IPV4 = 4, EX_IPV6 = 8, EX_HASHES = 16, EX_BITCOIN = 32, EX_ETHEREUM = 64,
    EX_MONERO = 128, EX_ALL = 255
};

// One validated candidate (what Extractor::extract_from_chunk yields), 16 bytes.
struct Candidate {
    uint32_t start;      // byte offset in the scanned buffer
    uint32_t len_type;   // length in bits 0..23, ItemType in bits 24..31
    uint32_t v4;         // IPv4 address (host order) for IT_IPV4, else 0
    uint32_t pad;        // string candidates of databases with globs: what the producer already knows — 0 nothing (the lookup walks the
                         // automaton), CAND_NO_GLOB no glob can match, CAND_GLOB some glob may (straight to the glob pass)
};
constexpr uint32_t CAND_NO_GLOB = 1, CAND_GLOB = 2;

// Anchors that need the rare-path validators (IPv6, e-mail, hash / crypto tokens), 8 bytes.
enum RareKind : uint32_t { RARE_V6 = 0, RARE_AT = 1, RARE_TOK = 2, HEAVY_B58 = 3, HEAVY_BECH32 = 4, HEAVY_ETH = 5, HEAVY_XMR = 6, RARE_DOM = 7 };
struct RareAnchor {
    uint32_t pos;       // RARE_V6: index of the 2nd ':' of a "::"; RARE_AT: index of '@'; RARE_TOK: token start
    uint32_t len_kind;  // RareKind in bits 0..7, token length in bits 8..31
};

// One database hit, 24 bytes. For IP hits `a` is the data-section offset; for pattern hits `a` is the literal
// pattern id (or 0xFFFFFFFF) and [ids_off, ids_off + n_globs) indexes the glob-id side buffer.
struct Hit {
    uint32_t cand;       // index into the candidate array
    uint32_t a;
    uint32_t ids_off;
    uint16_t n_globs;
    uint8_t kind;        // 2 = IP, 3 = pattern (QueryResult discriminants used by the oracle too)
    uint8_t prefix_len;
    uint32_t start;      // copy of the candidate span/type so the host needs no second gather
    uint32_t len_type;
};

struct LitSlot { uint64_t hash; uint32_t str_off; uint32_t pattern_id; };   // str_off 0xFFFFFFFF = empty
struct PslSlot { uint64_t hash; uint32_t off; uint32_t len; };              // len 0 = empty

// Device-resident database image (all pointers are device pointers).
struct DevDb {
    // IP tree, re-laid out as one uint2 {left,right} per node (records widened to 32 bit, host byte order)
    const uint2* ip_nodes;
    uint32_t ip_bm24_permille; // share of set bits in ip_bm24 (how much of the IPv4 space the /24 filter lets through)
    uint32_t ip_bm24_any;      // 0: no bit of ip_bm24 is set (the database answers no IPv4 address: k_anchor lists no IPv4 candidate)
    const uint32_t* ip_bm24; // 2^24 bits: bit v set iff the first 24 IPv4 levels for prefix v do not end in "not found"
    const uint2* ip_l1;      // 65536 entries: outcome of the first 16 IPv4 levels (x = kind | prefix << 8, y = node / data offset)
    const uint2* ip_l24;     // 2^24 entries, same encoding after 24 levels, or null (small trees: the 128 MB table is not built);
                             // kind 3: the /24 has a leaf table, y = its index
    const uint2* ip_leaf;    // leaf tables: 256 entries per undecided /24 (x = 1 not found | 2 | prefix << 8 found, y = data offset)
    uint32_t node_count;
    uint32_t ip_version;     // 4 or 6
    uint32_t v4_start_node;  // node reached after the 96 zero-bit steps of tree.rs:258-277 (v6 trees)
    uint32_t has_ip;
    // literal hash, re-hashed into one power-of-two open-addressing table keyed by the stored XXH64
    const LitSlot* lit_slots;
    uint32_t lit_mask;
    uint32_t has_literal;
    // bitmap over name_hash31() of the literal keys of <= 31 bytes (bit h & lit_bm_mask): clear => no key is this name.
    // k_validate drops (but counts) domain candidates that cannot hit when the database has no glob section.
    const uint32_t* lit_bm;
    uint32_t lit_bm_mask;
    const uint8_t* lit_pool;  // LHSH string pool: {u16 len, bytes, NUL}
    uint32_t lit_pool_size;
    uint32_t lit_max_len;     // longest key in the pool
    // paraglob buffer as stored on disk + a dense literal-id -> pattern-id list (from the ACLH table)
    const uint8_t* pg;
    uint32_t pg_len;
    uint32_t has_glob;
    uint32_t ac_start, ac_size;
    uint32_t patterns_off, pattern_count;
    uint32_t wild_off, wild_count;
    uint32_t glob_seg_off;
    uint32_t glob_max_segs;   // longest pattern in segments: bounds the star nesting of the matcher
    // dense DFA of the Aho-Corasick automaton (DbImage::build_ac_dfa), or null: next state = dfa[state * dfa_k +
    // dfa_cls[byte]] (bit 31: that state has output literals), dfa_node[state] = its node offset in the AC section
    const uint32_t* dfa;
    const uint8_t* dfa_cls;
    const uint32_t* dfa_node;
    uint32_t dfa_k;
    uint32_t dfa_states;          // number of states; they are numbered breadth-first, so the first rows are the shallowest
    // Suffix filter (databases whose globs are all of the form *LITERAL with one common first byte, e.g. "*.evil.com": what
    // indicator feeds hold): a glob can only match a name that ENDS with its literal, and the literal then starts at the d-th
    // occurrence of its first byte counted from the end of the name, d = occurrences of that byte in the literal. sfx_bm is a
    // bitmap over name_hash31 of the literals (<= 31 bytes; ASCII-folded for case-insensitive databases), sfx_first the byte,
    // sfx_dots bit d-1 set for every d that occurs. k_validate_dom tests <= 4 suffixes per name instead of walking the automaton.
    const uint32_t* sfx_bm;       // null: no suffix filter for this database
    uint32_t sfx_mask;
    uint32_t sfx_first;
    uint32_t sfx_dots;
    const uint32_t* lit2pat_off;  // [n_ac_lits + 1]
    const uint32_t* lit2pat;
    uint32_t n_ac_lits;
    uint32_t ac_alnum;            // 0: every literal of the automaton holds a byte that is no ASCII letter or digit, so a text of letters and digits only
                                  // (hex hashes, Base58 / Bech32 / 0x addresses: the long tokens) cannot contain one — no walk, no glob pass for it
    // public-suffix table
    const PslSlot* psl_slots;
    uint32_t psl_mask;
    const uint8_t* psl_pool;
    const uint32_t* tld_bloom;    // TLD_BLOOM_WORDS words: bloom over the LAST labels of all suffixes; behind them TLD_BLOOM_WORDS / 2 words: k_anchor's prefilter
    uint32_t max_tld_len;
    uint32_t max_suffix_len;      // longest suffix in bytes: a hash walk that has gone further cannot find one any more
    // exact open-addressing table (1 << TLD_TAB_BITS slots) of the last labels of <= 7 bytes: x = bytes 0..3, y = bytes
    // 4..6 | flags << 24 (0x80 occupied, 0x01 the label alone is a public suffix); slot = tld_tab_slot(x, y & 0xFFFFFF)
    const uint2* tld_tab;
    uint32_t tld_first[8];        // 256-bit set: bytes that start the last label of at least one suffix
    // case-insensitive databases (metadata match_mode = 1): Unicode lower-case data for literal queries with non-ASCII
    // characters (Rust str::to_lowercase: per-character mapping + the Final_Sigma rule); ASCII is folded inline
    uint32_t ci;
    const uint32_t* lc_map;       // lc_n entries of 3 words: code point, then len | utf8[0..2] << 8, then utf8[3..6]
    uint32_t lc_n;
    const uint2* lc_ign;          // Case_Ignorable ranges {first, last}, sorted
    uint32_t lc_n_ign;
    const uint2* lc_cased;        // Cased ranges
    uint32_t lc_n_cased;
};

constexpr uint32_t DFA_LDS_ENTRIES = 8192;      // 32 KiB of transition rows per workgroup of k_lookup
constexpr uint32_t DFA_LDS_ENTRIES_GLOB = 2048; // 8 KiB per workgroup of the glob pass of k_lookup
constexpr uint32_t DFA_LDS_ENTRIES_VAL = 2048;  // 8 KiB per workgroup of k_validate_dom (AC prefilter)
constexpr uint32_t TLD_BLOOM_BITS = 32768;
constexpr uint32_t TLD_BLOOM_WORDS = TLD_BLOOM_BITS / 32;

// List counters and statistics of one scan. Every counter that many waves bump with a returning atomic has a 128-byte line
// of its own: atomics on one line are served one after the other (~12 ns each), and with all counters in one line the
// list reservations of k_anchor's 4096 waves queued behind each other.
struct ScanCounters {
    alignas(128) unsigned long long lines;        // '\n' bytes
    uint32_t cand_true;              // candidates really written (n_cand counts chunk-allocated slots incl. padding)
    uint32_t hits_true;
    uint32_t error;                  // bit0: a candidate matches more than 65535 glob patterns (16-bit id count of the hit record), bit2: candidate > 16 MiB (24-bit length)
    uint32_t reserved0;
    alignas(128) uint32_t n_cand;    // candidates appended by the validation kernels (may exceed capacity → overflow)
    alignas(128) uint32_t n_cand_a;  // IPv4 candidates appended by k_anchor: a list of their own (TokParams::cands_a), so that their
                                     // lookups can start when k_anchor ends, beside the validation kernels
    alignas(128) uint32_t n_cand_m;  // candidates of k_validate over tokens / IPv6 / e-mail anchors and of k_rare when those run on the third
                                     // stream with a list (and a lookup pass) of their own
    uint32_t n_cand_d;               // ... of k_validate over the undecided domains (looked up beside the lookups of k_validate_dom's candidates)
    uint32_t n_cand_r;               // ... and of k_rare (its list is looked up after it, behind the lookups of n_cand_m)
    alignas(128) uint32_t n_dom;     // domain anchors (first byte of a label that follows a dot)
    alignas(128) uint32_t n_rare;    // IPv6 / e-mail anchors (k_anchor)
    alignas(128) uint32_t n_rare_dom;   // domain anchors k_validate_dom could not decide (general walk in k_validate): a list of their own,
                                        // so that k_validate can take k_anchor's rare anchors and the tokens while k_validate_dom runs
    alignas(128) uint32_t n_tok;     // long-token anchors (hash / crypto candidates)
    alignas(128) uint32_t n_heavy;   // tokens that need a checksum validator (Base58Check, Bech32, EIP-55, Monero)
    alignas(128) uint32_t n_hits;
    uint32_t n_ids;
    alignas(128) uint32_t n_final;   // dense final hit records written by pack_record
    uint32_t n_final_ids;            // entries of the pattern-id / data-offset side arrays
    uint32_t n_c4;                   // compact IPv4 records (PackParams::c4_out)
    // Forked scans without globs: the LAST kernel of every side-stream chain counts its finished workgroups in arrive_wgs[chain]; the
    // workgroup that completes a chain bumps chains_done, and k_finish — launched on the scan's stream without any event wait — polls
    // it (a cross-stream event join costs ~20 us between the last kernel and k_finish; the poll ends within a microsecond of the last arrival)
    alignas(128) uint32_t chains_done;
    uint32_t arrive_wgs[3];
    alignas(128) uint32_t n_glob_work;   // candidates whose text reaches an output state of the AC automaton (glob work list)
    uint32_t n_glob_work_d;          // ... queued by k_validate_dom itself (TokParams::glob_work_d): the glob pass over these starts when that kernel ends
    uint32_t n_spill;                // candidates handed to k_lookup_spill (more glob results / deeper star nesting than a lane of the glob pass holds)
};

// Final hit record, bit-identical to matchy_scan_hit_t in include/matchy_amd.h (checked by static_assert in capi.cpp).
struct FinalHit {
    uint32_t start;
    uint32_t len_type;    // length | item type << 24
    uint32_t value;       // IP: data offset; pattern: index of the first id
    uint8_t kind, prefix_len;
    uint16_t n_ids;
};

// pack_record (called from k_lookup): dense FinalHit records, pattern ids resolved to data offsets.
struct PackParams {
    const Hit* hits;
    uint32_t hit_cap;
    const uint32_t* ids;        // glob ids written by k_lookup
    uint32_t ids_cap;
    const uint32_t* lit_offsets;   // literal pattern id -> data offset (0xFFFFFFFF = no mapping), n_lit entries
    uint32_t n_lit;
    const uint32_t* glob_offsets;  // glob pattern id -> data offset, n_glob entries (PatternDataMappings)
    uint32_t n_glob;
    FinalHit* out;
    uint32_t out_cap;
    uint32_t* out_ids;
    long long* out_offs;
    uint32_t out_ids_cap;
    // optional mirror in pinned host memory (device-visible): records below these capacities are also written there, so
    // that a scan whose results fit needs no device-to-host copy of the hit records at all
    FinalHit* host_out;
    uint32_t host_cap;
    uint32_t* host_ids;
    long long* host_offs;
    uint32_t host_ids_cap;
    // MATCHY_SCAN_FETCH_COMPACT: IPv4 results leave as 8-byte records (c4_pack) in arrays of their own — device copy + pinned host mirror like
    // `out` / `host_out`, slots from ScanCounters::n_c4 — instead of as FinalHit records. nullptr: every result is a FinalHit.
    uint2* c4_out;
    uint2* host_c4;
    uint32_t c4_cap, host_c4_cap;
    ScanCounters* counters;
};

// Compact IPv4 result, bit-identical to matchy_scan_ip4_hit_t (include/matchy_amd.h): start | data offset in bits 0..21, text length - 7
// in bits 22..25, prefix length in bits 26..31. Only offered for databases whose data section is at most 4 MiB (Scanner::compact_possible).
constexpr uint32_t C4_DATA_BITS = 22;
__host__ __device__ inline uint2 c4_pack(uint32_t start, uint32_t len, uint32_t data_off, uint32_t prefix) {
    return make_uint2(start, data_off | ((len - 7u) << C4_DATA_BITS) | (prefix << (C4_DATA_BITS + 4)));
}

struct TokParams {
    const uint8_t* log;
    uint32_t len;
    uint32_t flags;           // ExtractFlags
    uint32_t min_labels;
    uint32_t debug;           // MATCHY_AMD_DEBUG: free for kernel experiments (unused in the shipped kernels)
    uint32_t filter_v4;       // 1: IPv4 candidates whose /24 has no database entry are counted but not listed (lookup scans)
    uint32_t filter_lit;      // 1: domain candidates whose XXH64 is not in DevDb::lit_bm are counted but not listed ...
    uint32_t filter_ac;       // 1: ... unless their text reaches an output state of the glob automaton (databases with globs)
    uint32_t n_segs;
    uint32_t seg_bytes;       // bytes of log per wavefront work item: a multiple of SEG_ALIGN chosen from the batch length
    // One launch of k_anchor covers the byte range [seg_base, scan_end) of the batch: segment s starts at seg_base + s * seg_bytes.
    // A whole batch is seg_base = 0, scan_end = len + 1 (position `len` closes a trailing token). A scan that is cut into slices
    // (Scanner::scan_device: the tail of one slice runs beside k_anchor of the next) launches k_anchor once per slice; seg_base
    // is a multiple of SEG_ALIGN, positions stay absolute, look-back and look-ahead across a cut read the neighbouring bytes.
    uint32_t seg_base;
    uint32_t scan_end;
    Candidate* cands;         // candidates of the validation kernels (domains, e-mail, IPv6, hashes, addresses)
    uint32_t cand_cap;
    uint32_t* n_cand;         // the counter of `cands` (ScanCounters::n_cand, or n_cand_m for the list of the third stream)
    Candidate* cands_a;       // IPv4 candidates of k_anchor (ScanCounters::n_cand_a)
    uint32_t cand_a_cap;
    uint32_t cand_chunk;      // slots k_anchor reserves per atomic for its IPv4 candidates (64: sparse list, 1024: one per line)
    RareAnchor* rare;         // IPv6 / e-mail anchors
    uint32_t rare_cap;
    uint32_t rare_chunk;      // 0, or slots per reservation in k_anchor when the previous batch's list was long (SparseWriter; tok_chunk below alike)
    RareAnchor* rare_dom;     // undecided domain anchors (written by k_validate_dom)
    uint32_t rare_dom_cap;
    uint32_t dom_chunk;       // slots k_anchor reserves per atomic for the domain list (ANCHOR_CHUNK, or less behind the preset first chunks: dom_static)
    // databases with globs, forked scans: k_validate_dom queues the candidates it flags CAND_GLOB on a work list of its own (indices into
    // `cands`, counter ScanCounters::n_glob_work_d), and the glob pass over them runs beside the lean pass over the rest. nullptr: off
    uint32_t* glob_work_d;
    uint32_t glob_work_d_cap;
    uint32_t vmode;           // k_validate: bit 0 = the rare list (IPv6 / e-mail anchors), bit 1 = the rare_dom list, bit 2 = the long tokens
    RareAnchor* tok;          // long-token anchors
    uint32_t tok_cap;
    uint32_t tok_chunk;
    RareAnchor* heavy;        // tokens that passed the cheap prefilters of k_validate and need k_rare
    uint32_t heavy_cap;
    // Domain anchors that survive k_anchor's prefilter, with 32 bytes of context copied from its LDS window so that
    // k_validate reads them coalesced instead of gathering log lines: per 64-slot tile 9 planes of 64 dwords
    // (dom_plane_index): plane 0 = anchor position j (bit 31 set: no context, 0xFFFFFFFF: unused slot),
    // planes 1..8 = log[j-24, j+8). dom_cap counts slots.
    uint32_t* dom_list;
    uint32_t dom_cap;
    uint32_t dom_static;      // 0, or ANCHOR_CHUNK: wave w of k_anchor owns chunk w of the list without a reservation; ScanCounters::n_dom was preset to
                              // waves x ANCHOR_CHUNK before the launch (Scanner::scan_device / k_finish)
    ScanCounters* counters;
    // inline_v4 = 1 (lookup scans of a database whose /24 bitmap thins the IPv4 candidates): k_anchor looks its IPv4 candidates up
    // itself — a few dozen at a time, whenever a wave has collected them — and writes the hit records through `pk`; no candidate
    // list, no lookup kernel for them, and the records cross the bus while the streaming pass runs instead of behind it.
    uint32_t inline_v4;
    PackParams pk;
};

// k_finish (lookup_kernels.hip): the last kernel of a scan. Copies the counter blocks of the scan's slices into pinned host memory
// and zeroes them on the device for the next scan — one small kernel instead of a device-to-host copy behind the scan and a memset
// in front of the next one.
// expect_chains > 0: the kernel first waits (polling ScanCounters::chains_done of block 0, with a time-out that sets error bit 8) until that
// many side-stream chains have reported their end.
void launch_finish(ScanCounters* dev, ScanCounters* host_pinned, int n_blocks, uint32_t expect_chains, uint32_t next_n_dom, hipStream_t stream);

// Called by every launch wrapper right after its hipLaunchKernelGGL: a launch the runtime rejects (LDS or register budget of
// another target, bad grid) would otherwise show up as a scan without hits. Throws mxy::HipError (engine.cpp).
void check_launch(const char* kernel);

struct LookupParams {
    const uint8_t* log;
    uint32_t len;
    const Candidate* cands;
    uint32_t cand_cap;
    const uint32_t* n_in;     // number of entries of `cands` (device memory); null = counters->n_cand
    Hit* hits;
    uint32_t hit_cap;
    uint32_t* ids;
    uint32_t ids_cap;
    // two-pass glob lookup: the lean pass (ac_filter = 1) lists the candidates that touch an AC output state in
    // glob_work; the glob pass (from_work = 1) handles exactly those. Both 0: one pass over all candidates.
    uint32_t* glob_work;
    uint32_t glob_work_cap;
    uint32_t ac_filter, from_work;
    const uint32_t* n_work;   // from_work: entries of glob_work (device memory); null = counters->n_glob_work
    uint32_t early_glob;      // lean pass: candidates flagged CAND_GLOB were queued by their producer (TokParams::glob_work_d) — skip them
    // forked scans of glob databases: the undecided domains have a candidate list (and a glob pass) of their own; a candidate of that
    // list that spills is listed with bit 31 set (spill_tag = 1), and the spill pass reads it from cands_alt
    const Candidate* cands_alt;
    uint32_t cand_alt_cap;
    uint32_t spill_tag;
    // glob candidates that exceed the per-lane storage of the glob pass: candidate indices, and per-thread scratch of the
    // spill pass (spill_words words per thread: one bit per pattern id, then the star stack)
    uint32_t* spill;
    uint32_t spill_cap;
    uint32_t* spill_scratch;
    uint32_t spill_words;
    uint32_t spill_blocks;
    // bulk scans: direct = 1 -> every hit is written as its final record at once (pack_record) instead of going through
    // the hit list and k_pack
    uint32_t direct;
    PackParams pk;
    ScanCounters* counters;
    // side-stream chains of a forked scan (see ScanCounters::chains_done): this launch is the last kernel of chain `arrive_chain`
    // (1..3; 0 = none) and reports its end in `arrive` (slice 0's counters)
    uint32_t arrive_chain;
    ScanCounters* arrive;
};


// Output lists are filled through wave-private chunks (one atomic per chunk, not per append); unused slots of a
// chunk hold a sentinel (anchor 0xFFFFFFFF, Candidate.len_type 0xFFFFFFFF, RareAnchor kind 0xFF, Hit.kind 0xFF).
constexpr uint32_t ANCHOR_CHUNK = 1024, RARE_CHUNK = 64, CAND_CHUNK = 512, HIT_CHUNK = 256;
constexpr uint32_t DOM_PLANES = 9;
// The domain anchor list is organised in tiles of 64 slots (one wave store per plane): tile t holds plane 0 of its 64 slots,
// then plane 1, ... — the planes of one slot are 256 bytes apart, which fits the immediate offset of a store instruction.
constexpr uint32_t DOM_TILE = 64;
static_assert(ANCHOR_CHUNK % DOM_TILE == 0, "chunks are whole tiles");
__host__ __device__ inline size_t dom_plane_index(uint32_t slot, uint32_t plane) {
    return (size_t)(slot / DOM_TILE) * (DOM_PLANES * DOM_TILE) + (size_t)plane * DOM_TILE + (slot % DOM_TILE);
}
// A wavefront of k_anchor works through the log in segments. Every segment end flushes the wave's anchor rings, however few
// anchors they hold, so long segments are cheaper per byte: the host makes one segment per resident wave (engine.cpp).
// Segment starts must fall on window boundaries.
constexpr uint32_t SEG_ALIGN = 8192, SEG_MIN = 8192, SEG_MAX = 1u << 26;
constexpr uint32_t MAX_GLOB_RESULTS = 32;
constexpr uint32_t MAX_GLOB_STARS = 24;

}  // namespace mxy
