#include "engine.h"
#include "unicode_lower.h"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <unordered_set>

#include "hashes.h"

namespace mxy {

void check_launch(const char* kernel) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw HipError{std::string("launch of ") + kernel + ": " + hipGetErrorString(e)};
}

// ------------------------------------------------------------------------------------------------ PSL
namespace {

std::string lib_dir() {
    Dl_info info;
    if (dladdr((void*)&lib_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t s = p.rfind('/');
        if (s != std::string::npos) return p.substr(0, s);
    }
    return ".";
}

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint8_t tmp[65536];
    size_t n;
    out.clear();
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) out.insert(out.end(), tmp, tmp + n);
    fclose(f);
    return true;
}

uint32_t tld_hash(const uint8_t* s, size_t n) {  // must match tld_hash_step/tld_hash_bit in device_common.h
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < n; ++i) h = (h ^ s[i]) * 16777619u;
    return (h ^ (h >> 15)) & (TLD_BLOOM_BITS - 1);
}

PslHost* load_psl() {
    std::vector<std::string> candidates;
    if (const char* env = getenv("MATCHY_AMD_PSL")) candidates.push_back(env);
    std::string d = lib_dir();
    candidates.push_back(d + "/../data/psl.bin");
    candidates.push_back(d + "/data/psl.bin");
    candidates.push_back(d + "/psl.bin");
    std::vector<uint8_t> buf;
    bool ok = false;
    for (auto& c : candidates) if (read_file(c, buf)) { ok = true; break; }
    if (!ok) throw std::runtime_error("matchy_amd: cannot find the public-suffix container psl.bin (set MATCHY_AMD_PSL)");
    if (buf.size() < 16 || memcmp(buf.data(), "PSLB", 4) != 0) throw std::runtime_error("matchy_amd: bad psl.bin header");
    uint32_t count, bytes;
    memcpy(&count, &buf[8], 4);
    memcpy(&bytes, &buf[12], 4);
    auto* h = new PslHost();
    std::string prev;
    size_t p = 16;
    for (uint32_t i = 0; i < count; ++i) {
        if (p + 2 > buf.size()) throw std::runtime_error("matchy_amd: truncated psl.bin");
        uint8_t shared = buf[p], rest = buf[p + 1];
        p += 2;
        if (shared > prev.size() || p + rest > buf.size()) throw std::runtime_error("matchy_amd: corrupt psl.bin");
        std::string s = prev.substr(0, shared) + std::string((const char*)&buf[p], rest);
        p += rest;
        h->suffixes.push_back(s);
        prev.swap(s);
    }
    size_t cap = 16;
    while (cap < h->suffixes.size() * 3) cap <<= 1;
    h->mask = (uint32_t)(cap - 1);
    h->slots.assign(cap, PslSlot{0, 0, 0});
    // two filters in one array: [0, TLD_BLOOM_WORDS) over the last labels of all suffixes, byte-wise hash (the general walk of
    // k_validate); behind it TLD_BLOOM_WORDS / 2 words for k_anchor's prefilter — labels that fit its 8-byte window, TWO bits per label
    // from the two-multiply hash (bits 0..13 and 14..27). (Until round 3 the prefilter folded the first filter, which also held its
    // bits, to half its size: a quarter of all short labels passed, "html" among them.)
    h->bloom.assign(TLD_BLOOM_WORDS + TLD_BLOOM_WORDS / 2, 0);
    for (const std::string& s : h->suffixes) {
        if (s.empty()) continue;
        uint64_t rh = psl_hash_init();
        for (size_t k = s.size(); k-- > 0;) rh = psl_hash_step(rh, (uint8_t)s[k]);
        uint64_t hh = psl_hash_finish(rh);
        uint32_t slot = (uint32_t)hh & h->mask;
        while (h->slots[slot].len != 0) slot = (slot + 1) & h->mask;
        h->slots[slot] = PslSlot{hh, (uint32_t)h->pool.size(), (uint32_t)s.size()};
        h->pool.insert(h->pool.end(), s.begin(), s.end());
        size_t dot = s.rfind('.');
        const uint8_t* last = (const uint8_t*)s.data() + (dot == std::string::npos ? 0 : dot + 1);
        size_t ll = s.size() - (dot == std::string::npos ? 0 : dot + 1);
        uint32_t bit = tld_hash(last, ll);
        h->bloom[bit >> 5] |= 1u << (bit & 31);
        if (ll >= 1 && ll <= 8) {  // second hash used by k_anchor's prefilter (labels that fit its 8-byte window)
            uint8_t k8[8] = {0};
            memcpy(k8, last, ll);
            uint32_t lo8, hi8;
            memcpy(&lo8, k8, 4); memcpy(&hi8, k8 + 4, 4);
            const uint32_t h8 = tld_hash8(lo8, hi8);
            const uint32_t ba = h8 & (TLD_BLOOM_BITS / 2 - 1), bb = (h8 >> 14) & (TLD_BLOOM_BITS / 2 - 1);
            h->bloom[TLD_BLOOM_WORDS + (ba >> 5)] |= 1u << (ba & 31);
            h->bloom[TLD_BLOOM_WORDS + (bb >> 5)] |= 1u << (bb & 31);
        }
        h->max_tld_len = std::max<uint32_t>(h->max_tld_len, (uint32_t)ll);
        h->max_suffix_len = std::max<uint32_t>(h->max_suffix_len, (uint32_t)s.size());
        if (ll) h->tld_first[last[0] >> 5] |= 1u << (last[0] & 31);
    }
    // exact table of short last labels; flag 1 = the label by itself is a suffix ("com"), unlike e.g. "ck" (only "*.ck")
    h->tld_tab.assign(1u << TLD_TAB_BITS, make_uint2(0u, 0u));
    std::unordered_set<std::string> whole(h->suffixes.begin(), h->suffixes.end());
    for (const std::string& s : h->suffixes) {
        size_t dot = s.rfind('.');
        std::string l = dot == std::string::npos ? s : s.substr(dot + 1);
        if (l.empty() || l.size() > 7) continue;
        uint8_t k[8] = {0};
        memcpy(k, l.data(), l.size());
        uint32_t lo, hi;
        memcpy(&lo, k, 4);
        memcpy(&hi, k + 4, 4);
        const uint32_t flags = 0x80u | (whole.count(l) ? 1u : 0u);
        uint32_t slot = tld_tab_slot(lo, hi);
        for (;;) {
            uint2& e = h->tld_tab[slot];
            if ((e.y >> 24) == 0) { e = make_uint2(lo, hi | (flags << 24)); break; }
            if (e.x == lo && (e.y & 0xFFFFFFu) == hi) break;  // same label from another suffix
            slot = (slot + 1) & ((1u << TLD_TAB_BITS) - 1);
        }
    }
    return h;
}

}  // namespace

const PslHost& PslHost::get() {
    static PslHost* inst = load_psl();
    return *inst;
}

// ------------------------------------------------------------------------------------------------ device image
void DeviceDb::upload(const DbImage& img, int dev) {
    device = dev;
    MXY_HIP(hipSetDevice(dev));
    const PslHost& psl = PslHost::get();
    std::vector<uint2> nodes;
    uint32_t v4_start;
    img.build_ip_nodes(nodes, v4_start);
    ip_nodes.upload(nodes);
    view.ip_nodes = ip_nodes.p;
    view.node_count = img.node_count;
    view.ip_version = (uint32_t)img.ip_version;
    view.v4_start_node = v4_start;
    view.has_ip = img.has_ip;
    bytes_uploaded = nodes.size() * sizeof(uint2);
    {
        // First-level table for IPv4: the outcome of the first 16 steps of SearchTree::lookup_v4 (tree.rs:46-90) for
        // every 16-bit prefix, so a lookup is one table load plus the (usually 1-3) remaining levels instead of ~18
        // dependent node loads. Entry: x = kind | prefix_len << 8 (kind 0 continue at node y, 1 not found,
        // 2 found with data offset y).
        std::vector<uint2> l1(65536, make_uint2(1u, 0u));
        if (img.node_count > 0) {
            for (uint32_t v = 0; v < 65536; ++v) {
                uint32_t node = v4_start;
                uint2 e = make_uint2(0u, 0u);
                bool done = false;
                for (int bi = 0; bi < 16 && !done; ++bi) {
                    const uint2 nd = nodes[node];
                    const uint32_t rec = ((v >> (15 - bi)) & 1) ? nd.y : nd.x;
                    if (rec == img.node_count) { e = make_uint2(1u, 0u); done = true; }
                    else if (rec < img.node_count) node = rec;
                    else {
                        const uint32_t off = rec - img.node_count;
                        e = off < 16 ? make_uint2(1u, 0u) : make_uint2(2u | ((uint32_t)(bi + 1) << 8), off - 16);
                        done = true;
                    }
                }
                if (!done) e = make_uint2(0u, node);
                l1[v] = e;
            }
        }
        ip_l1.upload(l1);
        view.ip_l1 = ip_l1.p;
        bytes_uploaded += l1.size() * sizeof(uint2);
        // /24 occupancy bitmap (2 MiB): bit v is clear iff lookup_v4 of every address under prefix v ends in "not found"
        // within the first 24 levels. The streaming kernel drops such candidates right after validation (they are still
        // counted), so only addresses that can hit reach the lookups.
        // First 24 levels as a direct table (2^24 entries, 128 MB of the 288 GB), and below a /24 that is not decided yet a LEAF
        // table with the outcome for each of its 256 addresses (2 KB per such /24: 40 K host addresses are 80 MB, 900 K are
        // 1.8 GB — what 288 GB of HBM are for): every IPv4 lookup is two dependent loads instead of 1 + up to 16. Entry kind 3 in
        // the /24 table: y = leaf index. Leaf tables are capped at 16 GB (MATCHY_AMD_LEAF_MB); /24s beyond the cap keep kind 0
        // (continue at node y: the reference's walk). All three are built on the device (ip_tables.hip): one walk per /24 and
        // per leaf address over the tree that was uploaded above. Tiny trees (< 4096 nodes; MATCHY_AMD_L24=0 / 1 forces) do
        // without the 128 MB table: 16-level table + walk, bitmap from the host.
        bool want24 = img.node_count >= 4096;
        if (const char* e = getenv("MATCHY_AMD_L24")) want24 = atoi(e) != 0 && img.node_count > 0;
        std::vector<uint32_t> bm(1u << 19, 0u);
        // The tables are accelerators, not requirements: when their memory cannot be had (a smaller GPU, a process that holds several
        // databases) the open degrades — no leaf tables (undecided /24s continue at their node: the reference's walk), then no /24
        // table at all (16-level table + walk, below) — instead of failing. The leaf tables take at most a quarter of what is free.
        bool have24 = false;
        if (want24) {
            try {
                ip_l24.alloc((size_t)1 << 24);
                ip_bm24.alloc(bm.size());
                DevBuf<uint32_t> ctr;
                ctr.alloc(2);
                MXY_HIP(hipMemset(ctr.p, 0, 8));
                launch_ip_l24(ip_nodes.p, img.node_count, v4_start, ip_l24.p, ip_bm24.p, ctr.p, nullptr);
                uint32_t undecided = 0;
                MXY_HIP(hipMemcpy(&undecided, ctr.p, 4, hipMemcpyDeviceToHost));
                size_t cap_leaf = ((size_t)16 << 30) / (256 * sizeof(uint2));
                size_t mem_free = 0, mem_total = 0;
                if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess) cap_leaf = std::min(cap_leaf, mem_free / 4 / (256 * sizeof(uint2)));
                else (void)hipGetLastError();
                if (const char* e = getenv("MATCHY_AMD_LEAF_MB")) cap_leaf = ((size_t)atoll(e) << 20) / (256 * sizeof(uint2));
                size_t n_leaf = std::min<size_t>(undecided, cap_leaf);
                DevBuf<uint32_t> leaf_node;
                if (n_leaf) {
                    try {
                        ip_leaf.alloc(n_leaf * 256);
                        leaf_node.alloc(n_leaf);
                    } catch (const HipError&) {
                        (void)hipGetLastError();
                        ip_leaf.free();
                        n_leaf = 0;   // the /24 entries of the undecided prefixes stay "continue at node"
                    }
                }
                if (n_leaf) {
                    launch_ip_leaf(ip_nodes.p, img.node_count, ip_l24.p, ctr.p + 1, (uint32_t)n_leaf, leaf_node.p, ip_leaf.p, nullptr);
                    MXY_HIP(hipDeviceSynchronize());   // leaf_node is released below
                    view.ip_leaf = ip_leaf.p;
                    bytes_uploaded += n_leaf * 256 * sizeof(uint2);
                }
                view.ip_l24 = ip_l24.p;
                bytes_uploaded += ((size_t)1 << 24) * sizeof(uint2);
                MXY_HIP(hipMemcpy(bm.data(), ip_bm24.p, bm.size() * 4, hipMemcpyDeviceToHost));   // for the statistics below
                have24 = true;
            } catch (const HipError&) {
                (void)hipGetLastError();
                ip_l24.free(); ip_leaf.free(); ip_bm24.free();
                view.ip_l24 = nullptr; view.ip_leaf = nullptr;
                std::fill(bm.begin(), bm.end(), 0u);
            }
        }
        if (!have24) {
            if (img.node_count > 0) {
                auto set_range = [&](uint32_t first, uint32_t count) {
                    for (uint32_t v = first; v < first + count;) {
                        if ((v & 31) == 0 && v + 32 <= first + count) { bm[v >> 5] = 0xFFFFFFFFu; v += 32; }
                        else { bm[v >> 5] |= 1u << (v & 31); ++v; }
                    }
                };
                struct Frame { uint32_t node, prefix, depth; };
                std::vector<Frame> st;
                st.push_back({v4_start, 0u, 0u});
                while (!st.empty()) {
                    const Frame f = st.back();
                    st.pop_back();
                    const uint2 nd = nodes[f.node];
                    for (uint32_t bit = 0; bit < 2; ++bit) {
                        const uint32_t rec = bit ? nd.y : nd.x;
                        const uint32_t depth = f.depth + 1;                       // levels consumed
                        const uint32_t prefix = f.prefix | (bit << (24 - depth));  // left-aligned in 24 bits
                        if (rec == img.node_count) continue;
                        if (rec < img.node_count) {
                            if (depth == 24) set_range(prefix, 1);
                            else st.push_back({rec, prefix, depth});
                        } else if (rec - img.node_count >= 16) {
                            set_range(prefix, 1u << (24 - depth));
                        }
                    }
                }
            }
            ip_bm24.upload(bm);
        }
        view.ip_bm24 = ip_bm24.p;
        {
            uint64_t set = 0;
            for (uint32_t wv : bm) set += (uint64_t)__builtin_popcount(wv);
            view.ip_bm24_permille = (uint32_t)(set * 1000 / ((uint64_t)bm.size() * 32));
            view.ip_bm24_any = set ? 1u : 0u;
        }
        bytes_uploaded += bm.size() * 4;
    }
    if (img.has_literal) {
        std::vector<LitSlot> slots;
        uint32_t mask;
        img.build_lit_table(slots, mask);
        lit_slots.upload(slots);
        const uint8_t* pool = img.bytes.data() + img.lh_off + img.lh_strings_offset;
        std::vector<uint8_t> pv(pool, pool + img.lh_strings_size);
        lit_pool.upload(pv);
        {
            // >= 64 bits per literal (false-positive rate <= 1.6 %), at least 4 Mbit: L2-resident for typical databases
            size_t bits = (size_t)1 << 22;
            size_t live = 0;
            for (const LitSlot& sl : slots) live += sl.str_off != 0xFFFFFFFFu;
            while (bits < live * 64 && bits < ((size_t)1 << 31)) bits <<= 1;
            std::vector<uint32_t> bm(bits / 32, 0u);
            const uint32_t bmask = (uint32_t)(bits - 1);
            // names of <= 31 bytes are tested whole (k_validate_dom decides names that fit its 32-byte context; longer ones take
            // the general path, which lists every valid name); keys of 32 bytes and more — file hashes — enter with their first 32
            // bytes and their length (k_validate tests every hex token of a hash length that way before it lists it)
            for (const LitSlot& sl : slots) {
                if (sl.str_off == 0xFFFFFFFFu || (size_t)sl.str_off + 2 > pv.size()) continue;
                const size_t kl = (size_t)pv[sl.str_off] | ((size_t)pv[sl.str_off + 1] << 8);
                if ((size_t)sl.str_off + 2 + kl > pv.size()) continue;
                uint64_t lane[4] = {0, 0, 0, 0};
                memcpy(lane, pv.data() + sl.str_off + 2, std::min<size_t>(kl, 32));
                const uint32_t b = name_hash31(lane[0], lane[1], lane[2], lane[3], (uint32_t)kl) & bmask;
                bm[b >> 5] |= 1u << (b & 31);
            }
            lit_bm.upload(bm);
            view.lit_bm = lit_bm.p; view.lit_bm_mask = bmask;
            bytes_uploaded += bm.size() * 4;
        }
        // longest stored key: a query that is longer after lower-casing cannot match any literal
        uint32_t max_len = 0;
        for (const LitSlot& sl : slots)
            if (sl.str_off != 0xFFFFFFFFu && (size_t)sl.str_off + 2 <= pv.size()) max_len = std::max<uint32_t>(max_len, (uint32_t)pv[sl.str_off] | ((uint32_t)pv[sl.str_off + 1] << 8));
        view.lit_max_len = max_len;
        view.lit_slots = lit_slots.p; view.lit_mask = mask; view.has_literal = 1;
        view.lit_pool = lit_pool.p; view.lit_pool_size = img.lh_strings_size;
        bytes_uploaded += slots.size() * sizeof(LitSlot) + pv.size();
    }
    if (img.has_glob) {
        const uint8_t* pgp = img.bytes.data() + img.pg_off;
        std::vector<uint8_t> pv(pgp, pgp + img.pg_len);
        pv.resize((pv.size() + 3) & ~(size_t)3, 0);
        pg.upload(pv);
        std::vector<uint32_t> off, ids;
        img.build_lit2pat(off, ids);
        if (ids.empty()) ids.push_back(0);
        lit2pat_off.upload(off);
        lit2pat.upload(ids);
        auto r32 = [&](size_t o) { uint32_t v; memcpy(&v, pgp + o, 4); return v; };
        view.pg = pg.p; view.pg_len = (uint32_t)img.pg_len; view.has_glob = 1;
        view.ac_start = r32(20); view.ac_size = r32(24);
        view.patterns_off = r32(36); view.pattern_count = r32(32);
        uint32_t unaligned = r32(40) + r32(44);
        view.wild_off = unaligned + (8 - unaligned % 8) % 8;  // recomputed like the reader (pg:1089-1093)
        view.wild_count = r32(60);
        view.glob_seg_off = r32(104);
        view.glob_max_segs = 1;
        for (uint32_t pid = 0; pid < view.pattern_count; ++pid) view.glob_max_segs = std::max<uint32_t>(view.glob_max_segs, r32((size_t)view.glob_seg_off + (size_t)pid * 8 + 4) & 0xFFFFu);
        view.lit2pat_off = lit2pat_off.p; view.lit2pat = lit2pat.p; view.n_ac_lits = (uint32_t)off.size() - 1;
        {
            // one dependent load per text byte instead of a node + edge-list walk with failure links: the automaton as a
            // dense table in HBM (there is room: a million literals need a few GB)
            std::vector<uint32_t> nx, noff;
            std::vector<uint8_t> cls;
            uint32_t k = 0;
            size_t limit = (size_t)8 << 30;
            if (const char* env = getenv("MATCHY_AMD_DFA_MAX_MB")) limit = (size_t)atoll(env) << 20;
            bool alnum_literal = true;
            view.ac_alnum = 1;
            if (img.build_ac_dfa(nx, cls, k, noff, limit, &alnum_literal)) {
                view.ac_alnum = alnum_literal ? 1u : 0u;
                // case-insensitive: the automaton holds lower-cased literals and the text is ASCII-lower-cased while it is
                // walked (paraglob_offset.rs:1198-1206) — here by giving 'A'..'Z' the classes of 'a'..'z'
                if (img.match_mode == 1) for (int c = 'A'; c <= 'Z'; ++c) cls[c] = cls[c + 32];
                dfa.upload(nx); dfa_node.upload(noff); dfa_cls.upload(cls);
                view.dfa = dfa.p; view.dfa_node = dfa_node.p; view.dfa_cls = dfa_cls.p; view.dfa_k = k;
                view.dfa_states = (uint32_t)noff.size();
                bytes_uploaded += nx.size() * 4 + noff.size() * 4 + 256;
            }
        }
        bytes_uploaded += pv.size() + (off.size() + ids.size()) * 4;
        // Suffix filter (DevDb::sfx_bm): every pattern is star + literal of >= 3 bytes (shorter ones can never match: they get no
        // automaton literal, Q8), all literals begin with the same byte, at most 4 different counts of that byte per literal.
        if (view.dfa && view.wild_count == 0 && view.pattern_count && !getenv("MATCHY_AMD_NO_SUFFIX_FILTER")) {
            const bool ci_db = img.match_mode == 1;
            bool ok = true;
            int first = -1;
            uint32_t dots = 0;
            std::vector<std::pair<uint32_t, uint32_t>> lits;   // (offset, length) in the paraglob buffer
            for (uint32_t pid = 0; pid < view.pattern_count && ok; ++pid) {
                const size_t io = (size_t)view.glob_seg_off + (size_t)pid * 8;
                const uint32_t firsth = r32(io), count = r32(io + 4) & 0xFFFFu;
                if (count != 2 || (size_t)firsth + 24 > img.pg_len) { ok = false; break; }
                const uint8_t* h0 = pgp + firsth;
                const uint8_t* h1 = pgp + firsth + 12;
                uint32_t dlen, doff;
                memcpy(&dlen, h1 + 4, 4); memcpy(&doff, h1 + 8, 4);
                if (h0[0] != 1 || h1[0] != 0 || (size_t)doff + dlen > img.pg_len) { ok = false; break; }
                if (dlen < 3) continue;   // never matches
                const uint8_t* l = pgp + doff;
                const int fb = ci_db ? (int)ascii_lower1(l[0]) : (int)l[0];
                if (first < 0) first = fb;
                if (fb != first) { ok = false; break; }
                uint32_t d = 0;
                for (uint32_t k = 0; k < dlen; ++k) d += (ci_db ? ascii_lower1(l[k]) : l[k]) == (uint32_t)first;
                if (d > 32) { ok = false; break; }
                dots |= 1u << (d - 1);
                if (dlen <= 31) lits.push_back({doff, dlen});   // a longer literal cannot end a name of <= 31 bytes (the ones decided here)
            }
            if (ok && first >= 0 && __builtin_popcount(dots) <= 4) {
                size_t bits = (size_t)1 << 16;
                while (bits < lits.size() * 64 && bits < ((size_t)1 << 31)) bits <<= 1;
                std::vector<uint32_t> bm(bits / 32, 0u);
                const uint32_t bmask = (uint32_t)(bits - 1);
                for (auto& ol : lits) {
                    uint64_t lane[4] = {0, 0, 0, 0};
                    memcpy(lane, pgp + ol.first, ol.second);
                    if (ci_db) for (uint64_t& w : lane) w = ascii_lower8(w);
                    const uint32_t b = name_hash31(lane[0], lane[1], lane[2], lane[3], ol.second) & bmask;
                    bm[b >> 5] |= 1u << (b & 31);
                }
                sfx_bm.upload(bm);
                view.sfx_bm = sfx_bm.p; view.sfx_mask = bmask; view.sfx_first = (uint32_t)first; view.sfx_dots = dots;
                bytes_uploaded += bm.size() * 4;
            }
        }
    }
    view.ci = img.match_mode == 1 ? 1u : 0u;
    if (view.ci) {
        const LowerTable& lt = LowerTable::get();
        std::vector<uint32_t> m;
        m.reserve(lt.map.size() * 3);
        for (const LowerMapEntry& e : lt.map) {
            m.push_back(e.cp);
            m.push_back((uint32_t)e.len | ((uint32_t)e.utf8[0] << 8) | ((uint32_t)e.utf8[1] << 16) | ((uint32_t)e.utf8[2] << 24));
            m.push_back((uint32_t)e.utf8[3] | ((uint32_t)e.utf8[4] << 8) | ((uint32_t)e.utf8[5] << 16) | ((uint32_t)e.utf8[6] << 24));
        }
        std::vector<uint2> ri, rc;
        for (const CpRange& r : lt.ignorable) ri.push_back(make_uint2(r.first, r.last));
        for (const CpRange& r : lt.cased) rc.push_back(make_uint2(r.first, r.last));
        lc_map.upload(m); lc_ign.upload(ri); lc_cased.upload(rc);
        view.lc_map = lc_map.p; view.lc_n = (uint32_t)lt.map.size();
        view.lc_ign = lc_ign.p; view.lc_n_ign = (uint32_t)ri.size();
        view.lc_cased = lc_cased.p; view.lc_n_cased = (uint32_t)rc.size();
        bytes_uploaded += m.size() * 4 + (ri.size() + rc.size()) * 8;
    }
    {
        // pattern id -> data offset tables for pack_record
        std::vector<uint32_t> lo, go;
        if (img.has_literal) {
            if (!img.lit_data_offsets.empty()) lo = img.lit_data_offsets;
            else {
                uint32_t mx = 0;
                for (auto& kv : img.lit_data_map) mx = std::max(mx, kv.first + 1);
                lo.assign(mx, 0xFFFFFFFFu);
                for (auto& kv : img.lit_data_map) lo[kv.first] = kv.second;
            }
        }
        if (img.has_glob) {
            go.resize(img.pdm_count);
            if (img.pdm_count) memcpy(go.data(), img.bytes.data() + img.pdm_off, img.pdm_count * 4);
        }
        if (lo.empty()) lo.push_back(0xFFFFFFFFu);
        if (go.empty()) go.push_back(0);
        n_lit_offsets = img.has_literal ? (uint32_t)lo.size() : 0;
        n_glob_offsets = img.has_glob ? (uint32_t)img.pdm_count : 0;
        lit_offsets.upload(lo);
        glob_offsets.upload(go);
    }
    psl_slots.upload(psl.slots);
    psl_pool.upload(psl.pool);
    bloom.upload(psl.bloom);
    view.psl_slots = psl_slots.p; view.psl_mask = psl.mask; view.psl_pool = psl_pool.p; view.tld_bloom = bloom.p;
    view.max_tld_len = psl.max_tld_len;
    view.max_suffix_len = psl.max_suffix_len;
    tld_tab.upload(psl.tld_tab);
    view.tld_tab = tld_tab.p;
    for (int k = 0; k < 8; ++k) view.tld_first[k] = psl.tld_first[k];
    bytes_uploaded += psl.slots.size() * sizeof(PslSlot) + psl.pool.size() + psl.bloom.size() * 4;
}

// ------------------------------------------------------------------------------------------------ scanner
Scanner::Scanner(std::shared_ptr<const DbImage> img, std::shared_ptr<DeviceDb> ddb, uint32_t extract_flags, uint32_t min_labels)
    : img_(std::move(img)), ddb_(std::move(ddb)), flags_(extract_flags), min_labels_(min_labels ? min_labels : 2) {
    MXY_HIP(hipSetDevice(ddb_->device));
    hipDeviceProp_t prop;
    MXY_HIP(hipGetDeviceProperties(&prop, ddb_->device));
    n_cu_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    counters_.alloc(MAX_SLICES);
    MXY_HIP(hipMemset(counters_.p, 0, sizeof(ScanCounters) * MAX_SLICES));
    counters_clean_ = true;
    MXY_HIP(hipHostMalloc((void**)&host_slices_, sizeof(ScanCounters) * MAX_SLICES, hipHostMallocDefault));
    for (auto& e : ev_) MXY_HIP(hipEventCreate(&e));
    MXY_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
    MXY_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
    MXY_HIP(hipStreamCreateWithFlags(&aux_stream_, hipStreamNonBlocking));
    MXY_HIP(hipEventCreateWithFlags(&ev_join2_, hipEventDisableTiming));
    MXY_HIP(hipStreamCreateWithFlags(&aux2_stream_, hipStreamNonBlocking));
}

Scanner::~Scanner() {
    if (pinned_) (void)hipHostFree(pinned_);
    if (mirror_) (void)hipHostFree(mirror_);
    if (mirror_c4_) (void)hipHostFree(mirror_c4_);
    if (pinned_c4_) (void)hipHostFree(pinned_c4_);
    if (host_slices_) (void)hipHostFree(host_slices_);
    for (auto& e : ev_) if (e) (void)hipEventDestroy(e);
    for (auto& e : ev_anchor_) if (e) (void)hipEventDestroy(e);
    for (auto& e : ev_misc_) if (e) (void)hipEventDestroy(e);
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_join_) (void)hipEventDestroy(ev_join_);
    if (aux_stream_) (void)hipStreamDestroy(aux_stream_);
    if (ev_join2_) (void)hipEventDestroy(ev_join2_);
    if (ev_join3_) (void)hipEventDestroy(ev_join3_);
    if (ev_dom_) (void)hipEventDestroy(ev_dom_);
    if (ev_v1_) (void)hipEventDestroy(ev_v1_);
    if (aux2_stream_) (void)hipStreamDestroy(aux2_stream_);
    if (dom_stream_) (void)hipStreamDestroy(dom_stream_);
    if (host_stream_) (void)hipStreamDestroy(host_stream_);
}

// work lists of one slice for `len` bytes of log
void Scanner::Work::ensure(uint32_t len) {
    const size_t want_c = std::max<size_t>(4096, (size_t)len / 24);
    const size_t want_r = std::max<size_t>(1024, (size_t)len / 256);
    const size_t want_h = std::max<size_t>(1024, want_c / 4);
    if (cands.n < want_c) cands.alloc(want_c);
    if (hits.n < want_h) hits.alloc(want_h);
    if (ids.n < want_h) ids.alloc(want_h);
    if (cands_a.n < want_c) cands_a.alloc(want_c);
    if (cands_m.n < want_c / 8) cands_m.alloc(want_c / 8);
    if (cands_r.n < want_r) cands_r.alloc(want_r);
    if (cands_d.n < want_r) cands_d.alloc(want_r);
    if (rare.n < want_r) rare.alloc(want_r);
    if (rare_dom.n < want_r) rare_dom.alloc(want_r);
    if (tok.n < want_r) tok.alloc(want_r);
    if (heavy.n < want_r) heavy.alloc(want_r);
    // domain anchor list: slots of DOM_PLANES dwords, whole 1024-slot chunks (see TokParams::dom_list)
    const size_t want_d = ((std::max<size_t>(8192, (size_t)len / 96) + ANCHOR_CHUNK - 1) / ANCHOR_CHUNK) * ANCHOR_CHUNK;
    if (dom_slots < want_d) { dom_list.alloc(want_d * DOM_PLANES); dom_slots = want_d; }
}

void Scanner::ensure_final(size_t recs, size_t ids) {
    if (final_.n < recs) final_.alloc(recs);
    if (final_ids_.n < recs + ids) { final_ids_.alloc(recs + ids); final_offs_.alloc(recs + ids); }
}

void Scanner::ensure_capacity(uint32_t len) {
    work_[0].ensure(len);
    ensure_final(work_[0].hits.n, work_[0].ids.n);
}

void Scanner::ensure_mirror(uint32_t recs, uint32_t ids) {
    if (mirror_ && mirror_cap_ >= recs && mirror_ids_cap_ >= ids) return;
    if (mirror_) (void)hipHostFree(mirror_);
    mirror_cap_ = std::max(recs, mirror_cap_); mirror_ids_cap_ = std::max(ids, mirror_ids_cap_);
    const size_t bytes = (size_t)mirror_cap_ * sizeof(FinalHit) + (size_t)mirror_ids_cap_ * 12 + 64;
    MXY_HIP(hipHostMalloc(&mirror_, bytes, hipHostMallocDefault));
}

void Scanner::ensure_mirror_c4(uint32_t recs) {
    if (mirror_c4_ && mirror_c4_cap_ >= recs) return;
    if (mirror_c4_) (void)hipHostFree(mirror_c4_);
    mirror_c4_ = nullptr;
    mirror_c4_cap_ = std::max(recs, mirror_c4_cap_);
    MXY_HIP(hipHostMalloc((void**)&mirror_c4_, (size_t)mirror_c4_cap_ * sizeof(uint2), hipHostMallocDefault));
}

namespace {
// grid multipliers (workgroups per CU) of k_anchor / k_validate_dom / k_lookup: what is resident at once (grid-stride kernels; a
// larger grid only adds a partially filled second round). MATCHY_AMD_GRID=a,v,l overrides for experiments.
// k_anchor: one full round of resident workgroups; k_validate_dom: one; k_lookup: 2 per CU measured best (the string
// lookups are chains of dependent loads: more waves in flight only add contention on the random table accesses).
void grid_multipliers(bool filter_ac, int (&gm)[3]) {
    static const int occ_a = anchor_blocks_per_cu(), occ_v = validate_blocks_per_cu(false), occ_v_ac = validate_blocks_per_cu(true);
    // k_validate_dom: at most 4 workgroups per CU — the fifth that registers and LDS would allow leaves no LDS for the kernels
    // that run beside it (k_validate over tokens / IPv6 / e-mail anchors: 26 KB per workgroup), which then start when
    // k_validate_dom's workgroups retire (measured: tail 0.250 -> 0.232 ms)
    gm[0] = occ_a; gm[1] = std::min(filter_ac ? occ_v_ac : occ_v, 4); gm[2] = 2;
    if (const char* g = getenv("MATCHY_AMD_GRID")) {   // 0 keeps the default of that kernel
        int o[3] = {0, 0, 0};
        (void)sscanf(g, "%d,%d,%d", &o[0], &o[1], &o[2]);
        for (int k = 0; k < 3; ++k) if (o[k] > 0) gm[k] = o[k];
    }
    for (int& m : gm) m = std::max(1, std::min(m, 64));
}
}  // namespace

// workgroups for a list kernel: the default while the previous scan's list was short, else one workgroup per `per_wg` entries of it,
// at most max_per_cu per CU (workgroups beyond the end of a list leave at once)
int Scanner::grid_for(uint32_t n_hint, uint32_t per_wg, int dflt, int max_per_cu) const {
    static const bool fixed = getenv("MATCHY_AMD_FIXED_GRIDS") != nullptr;
    if (fixed) return dflt;
    const uint64_t want = ((uint64_t)n_hint + per_wg - 1) / per_wg;
    return (int)std::max<uint64_t>((uint64_t)dflt, std::min<uint64_t>(want, (uint64_t)n_cu_ * (uint64_t)max_per_cu));
}

// Kernel parameters of slice `sl` for the byte range [lo, hi) of the batch (hi = len + 1 for the range that ends the batch).
// The lists and counters are the slice's own; the final record arrays (and the counters that hand out their slots: slice 0's)
// are shared by all slices.
void Scanner::slice_params(int sl, const uint8_t* dptr, uint32_t len, uint32_t lo, uint32_t hi, bool lookup, bool host_mirror, SliceLaunch& L) {
    Work& w = work_[sl];
    ScanCounters* ctr = counters_.p + sl;
    TokParams& tp = L.tp;
    tp = TokParams{};
    tp.log = dptr; tp.len = len; tp.flags = flags_; tp.min_labels = min_labels_;
    tp.filter_v4 = lookup ? 1u : 0u;
    // without a glob section a string candidate can only hit through the literal table
    // with a glob section also through a glob, and every glob with a literal part needs an output state of the AC
    // automaton on the way (pure-wildcard patterns match without one: no prefilter then)
    const bool ac_ok = ddb_->view.has_glob && ddb_->view.dfa && ddb_->view.wild_count == 0;
    tp.filter_ac = (lookup && ac_ok) ? 1u : 0u;
    tp.filter_lit = (lookup && (!ddb_->view.has_glob || ac_ok)) ? 1u : 0u;
    if (const char* dbg = getenv("MATCHY_AMD_DEBUG")) tp.debug = (uint32_t)atoi(dbg);
    tp.cands = w.cands.p; tp.cand_cap = (uint32_t)w.cands.n; tp.n_cand = &ctr->n_cand;
    tp.cands_a = w.cands_a.p; tp.cand_a_cap = (uint32_t)w.cands_a.n;
    // IPv4 candidates are listed sparsely when the /24 bitmap of the database filters most of the address space
    tp.cand_chunk = (lookup && ddb_->view.ip_bm24_permille <= 250) ? 64u : 1024u;
    tp.rare = w.rare.p; tp.rare_cap = (uint32_t)w.rare.n;
    tp.rare_dom = w.rare_dom.p; tp.rare_dom_cap = (uint32_t)w.rare_dom.n;
    tp.vmode = 7u;
    tp.tok = w.tok.p; tp.tok_cap = (uint32_t)w.tok.n;
    tp.heavy = w.heavy.p; tp.heavy_cap = (uint32_t)w.heavy.n;
    tp.dom_list = w.dom_list.p; tp.dom_cap = (uint32_t)w.dom_slots;
    tp.counters = ctr;
    grid_multipliers(tp.filter_ac != 0, L.gm);
    {
        // k_anchor: one round of resident workgroups and ONE segment per wave, all of the same size — the segments are
        // handed out statically, so anything else leaves some waves with one segment more than the others (with three or
        // four segments per wave that was 19 % of the kernel), and every segment end flushes the wave's anchor rings.
        // Short ranges get SEG_MIN segments and fewer waves.
        const uint64_t waves = (uint64_t)n_cu_ * L.gm[0] * 4;
        const uint64_t span = (uint64_t)hi - lo;
        uint64_t sb = ((span + waves - 1) / waves + SEG_ALIGN - 1) / SEG_ALIGN * SEG_ALIGN;
        if (const char* e = getenv("MATCHY_AMD_SEG_KB")) sb = (uint64_t)atoi(e) * 1024 / SEG_ALIGN * SEG_ALIGN;
        tp.seg_bytes = (uint32_t)std::min<uint64_t>(SEG_MAX, std::max<uint64_t>(SEG_MIN, sb));
        tp.n_segs = (uint32_t)((span + tp.seg_bytes - 1) / tp.seg_bytes);
        tp.seg_base = lo; tp.scan_end = hi;
    }
    L.grid_anchor = (int)std::min<uint32_t>((tp.n_segs + 3) / 4, (uint32_t)n_cu_ * L.gm[0]);
    if (L.grid_anchor < 1) L.grid_anchor = 1;
    {
        // starting chunk of k_anchor's sparse lists when the previous batch filled them (SparseWriter in k_anchor.hip): a quarter of what a
        // wave wrote then, in whole 64-slot steps, at most 4032 (the chunk state holds 12 bits)
        const uint32_t waves = (uint32_t)L.grid_anchor * 4u;
        auto chunk_for = [&](uint32_t n_prev) -> uint32_t {
            if (n_prev < 262144u) return 0u;
            return std::min<uint32_t>(4032u, std::max<uint32_t>(64u, (n_prev / waves / 4u) & ~63u));
        };
        tp.tok_chunk = chunk_for(hint_.n_tok);
        tp.rare_chunk = chunk_for(hint_.n_rare);
    }
    LookupParams& lp = L.lp;
    lp = LookupParams{};
    if (lookup) {
        lp.log = dptr; lp.len = len; lp.cands = w.cands.p; lp.cand_cap = (uint32_t)w.cands.n;
        lp.hits = w.hits.p; lp.hit_cap = (uint32_t)w.hits.n; lp.ids = w.ids.p; lp.ids_cap = (uint32_t)w.ids.n;
        if (ddb_->view.has_glob) {
            if (w.glob_work.n < w.cands.n / 8) w.glob_work.alloc(w.cands.n / 8);
            lp.glob_work = w.glob_work.p; lp.glob_work_cap = (uint32_t)w.glob_work.n;
            if (early_glob_ && sl == 0 && ac_ok) {
                if (w.glob_work_d.n < w.cands.n / 8) w.glob_work_d.alloc(w.cands.n / 8);
                tp.glob_work_d = w.glob_work_d.p; tp.glob_work_d_cap = (uint32_t)w.glob_work_d.n;
                lp.early_glob = 1u;
            }
            setup_spill(sl, lp);
        }
        lp.counters = ctr;
        PackParams pp{};
        pp.hits = w.hits.p; pp.hit_cap = (uint32_t)w.hits.n; pp.ids = w.ids.p; pp.ids_cap = (uint32_t)w.ids.n;
        pp.lit_offsets = ddb_->lit_offsets.p; pp.n_lit = ddb_->n_lit_offsets;
        pp.glob_offsets = ddb_->glob_offsets.p; pp.n_glob = ddb_->n_glob_offsets;
        pp.out = final_.p; pp.out_cap = (uint32_t)final_.n;
        pp.out_ids = final_ids_.p; pp.out_offs = final_offs_.p; pp.out_ids_cap = (uint32_t)final_ids_.n;
        if (host_mirror) {
            uint8_t* mb = (uint8_t*)mirror_;
            pp.host_out = (FinalHit*)mb; pp.host_cap = mirror_cap_;
            pp.host_ids = (uint32_t*)(mb + (size_t)mirror_cap_ * sizeof(FinalHit));
            pp.host_offs = (long long*)(mb + (size_t)mirror_cap_ * sizeof(FinalHit) + (((size_t)mirror_ids_cap_ * 4 + 7) & ~(size_t)7));
            pp.host_ids_cap = mirror_ids_cap_;
        }
        if (compact_) {
            pp.c4_out = c4_.p; pp.c4_cap = (uint32_t)c4_.n;
            if (host_mirror) { pp.host_c4 = mirror_c4_; pp.host_c4_cap = mirror_c4_cap_; }
        }
        pp.counters = counters_.p;   // n_final / n_final_ids of slice 0 hand out the slots of the shared record arrays
        // k_lookup writes the final records itself (the PCIe writes of the mirror overlap the lookups); only the
        // single-query path (lookup_one) reads the raw hit list
        lp.direct = 1u;
        lp.pk = pp;
        // sparse IPv4 candidates (the /24 bitmap lets a few per cent through): k_anchor looks them up itself while it streams
        // (k_anchor.hip v4_lookup_flush) — their hit records, most of the result traffic of a log scan, then cross the bus under
        // the streaming pass instead of behind it. Dense lists (a database that answers most addresses) keep the lookup kernel:
        // there the walks are the work, and a kernel of lanes that do nothing else hides their latency better.
        static const bool env_no_inline = getenv("MATCHY_AMD_INLINE_V4") && atoi(getenv("MATCHY_AMD_INLINE_V4")) == 0;
        tp.inline_v4 = (!env_no_inline && tp.filter_v4 && tp.cand_chunk == 64u) ? 1u : 0u;
        tp.pk = pp;
    }
    // a database without any IPv4 answer lists no IPv4 candidate (the /24 bitmap is empty): nothing to look up
    L.ip_pass = lookup && !tp.inline_v4 && (ddb_->view.ip_bm24_any || !tp.filter_v4);
    L.la = lp;
    if (L.ip_pass) {
        L.la.cands = w.cands_a.p; L.la.cand_cap = (uint32_t)w.cands_a.n; L.la.n_in = &ctr->n_cand_a;
        L.la.glob_work = nullptr; L.la.glob_work_cap = 0;
    }
    // Forked scans of a database without globs: the third stream (tokens, IPv6 / e-mail anchors, k_rare) gets a candidate list and a
    // lookup pass of its own, so the scan's stream does not have to join it in front of its lookups (an event wait between two
    // kernels costs ~20 us of the chain even when the event is long done). With a glob section the lookups share the glob work
    // list and its counter: one list, joined as before.
    L.split_misc = lookup && !ddb_->view.has_glob;
    L.lm = lp;
    L.lr = lp;
    L.ld = lp;
    if (lp.early_glob) {
        // forked scan of a glob database: the undecided domains get a list and ONE glob pass over all of it (no work list: a few thousand
        // candidates) on a stream of their own; what spills there is listed with bit 31 set and read from cands_alt by the spill pass
        lp.cands_alt = w.cands_d.p; lp.cand_alt_cap = (uint32_t)w.cands_d.n;
        L.ld = lp;
        L.ld.cands = w.cands_d.p; L.ld.cand_cap = (uint32_t)w.cands_d.n; L.ld.n_in = &ctr->n_cand_d;
        L.ld.glob_work = nullptr; L.ld.glob_work_cap = 0; L.ld.early_glob = 0; L.ld.spill_tag = 1u;
    }
    if (L.split_misc) {
        L.ld.cands = w.cands_d.p; L.ld.cand_cap = (uint32_t)w.cands_d.n; L.ld.n_in = &ctr->n_cand_d;
        L.lm.cands = w.cands_m.p; L.lm.cand_cap = (uint32_t)w.cands_m.n; L.lm.n_in = &ctr->n_cand_m;
        L.lr.cands = w.cands_r.p; L.lr.cand_cap = (uint32_t)w.cands_r.n; L.lr.n_in = &ctr->n_cand_r;
    }
    // one workgroup on every other CU: enough lanes to keep the result traffic on the bus, and the validation kernels beside it
    // keep nearly all of their resident waves (128 / 256 / 512 workgroups measured 1.199 / 1.213 / 1.241 ms per headline batch)
    // (dense lists — a database that answers most addresses, C5 — are latency-bound trie walks for every line: full grid)
    static const int ip_wgs = getenv("MATCHY_AMD_IPGRID") ? atoi(getenv("MATCHY_AMD_IPGRID")) : 0;
    L.ip_dense = tp.cand_chunk != 64u;
    L.ip_grid = ip_wgs > 0 ? ip_wgs : L.ip_dense ? n_cu_ * L.gm[2] : std::max(1, n_cu_ / 2);
}

// How a device-resident batch of `len` bytes is cut into slices: cuts[0] = 0 < cuts[1] < ... < cuts[n] = len + 1, inner cuts on
// multiples of one full round of k_anchor segments (waves x SEG_ALIGN bytes), so that every wave of a slice gets a segment of
// the same size. MATCHY_AMD_SLICES = "n" (equal slices) or "a,b,c,..." (relative sizes) overrides the default.
int Scanner::plan_slices(uint32_t len, int want, uint32_t (&cuts)[MAX_SLICES + 1]) {
    int gm[3];
    grid_multipliers(false, gm);
    // an explicit slice count (matchy_scanner_set_slices: tests, experiments) is honoured down to SEG_ALIGN-sized slices
    const uint64_t unit = want > 0 ? (uint64_t)SEG_ALIGN : (uint64_t)n_cu_ * gm[0] * 4 * SEG_ALIGN;
    double share[MAX_SLICES];
    int n = 0;
    static const char* env = getenv("MATCHY_AMD_SLICES");
    if (want > 0) {
        n = std::min(want, (int)MAX_SLICES);
        for (int k = 0; k < n; ++k) share[k] = 1.0;
    } else if (env && *env) {
        const char* q = env;
        while (*q && n < MAX_SLICES) {
            char* e = nullptr;
            const double v = strtod(q, &e);
            if (e == q) break;
            share[n++] = v > 0 ? v : 1.0;
            q = *e == ',' ? e + 1 : e;
        }
        if (n == 1) { n = std::max(1, std::min((int)share[0], (int)MAX_SLICES)); for (int k = 0; k < n; ++k) share[k] = 1.0; }
    } else {
        // default: ONE slice. Measured on the headline batch (profiles/r03_slices_sweep.txt): with k_anchor's segments handed out
        // statically — one per resident wave — the tail kernels of slice i take wave slots that k_anchor of slice i + 1 needs at
        // its start, the workgroups that start late finish late, and the whole scan gets slower (2 slices 1.32 ms, 4 slices 1.53 ms
        // against 1.16 ms for one). The mechanism stays available (matchy_scanner_set_slices, MATCHY_AMD_SLICES).
        n = 1;
        share[0] = 1.0;
    }
    // a slice wants at least two rounds of segments (an explicitly requested one: one segment)
    n = (int)std::min<uint64_t>((uint64_t)std::max(n, 1), std::max<uint64_t>(1, ((uint64_t)len + 1) / (want > 0 ? unit : 2 * unit)));
    double total = 0;
    for (int k = 0; k < n; ++k) total += share[k];
    cuts[0] = 0;
    double acc = 0;
    for (int k = 1; k < n; ++k) {
        acc += share[k - 1];
        uint64_t c = (uint64_t)((double)len * acc / total) / unit * unit;
        c = std::max<uint64_t>(c, (uint64_t)cuts[k - 1] + unit);
        cuts[k] = (uint32_t)c;
    }
    cuts[n] = len + 1;
    for (int k = 1; k <= n; ++k) if (cuts[k] <= cuts[k - 1]) return (cuts[1] = len + 1, 1);   // degenerate: one slice
    return n;
}

void Scanner::scan_device(const uint8_t* dptr, uint32_t len, bool lookup, hipStream_t stream, bool host_mirror, bool fork, int slices, bool compact) {
    if (len >= 0x7FFF0000u) throw HipError{"scan_device: chunk too large (must be < 2^31 bytes)"};
    if (((uintptr_t)dptr & 15) != 0) throw HipError{"scan_device: device pointer must be 16-byte aligned"};
    MXY_HIP(hipSetDevice(ddb_->device));
    last_ptr_ = dptr; last_len_ = len; last_lookup_ = lookup; last_mirror_ = host_mirror; last_fork_ = fork; last_slices_ = slices;
    last_compact_ = compact;
    compact_ = compact && lookup && compact_possible();
    last_forked_ = false;
    spill_done_ = false;
    expect_chains_ = 0;
    // MATCHY_AMD_NO_FORK=1 keeps everything on one stream.
    static const bool env_no_fork = getenv("MATCHY_AMD_NO_FORK") != nullptr;
    const bool no_fork = env_no_fork || !fork;
    uint32_t cuts[MAX_SLICES + 1] = {0, len + 1};
    const int ns = (no_fork || !lookup) ? 1 : plan_slices(len, slices, cuts);
    n_slices_ = ns;
    // buffers first: what they may allocate (lists, mirror, glob work list) must not sit between the launches
    if (ns == 1) ensure_capacity(len);
    else {
        size_t recs = 0, ids = 0;
        for (int k = 0; k < ns; ++k) { work_[k].ensure(cuts[k + 1] - cuts[k]); recs += work_[k].hits.n; ids += work_[k].ids.n; }
        ensure_final(recs, ids);
    }
    mirror_used_ = false;
    if (lookup && host_mirror) {
        static const uint32_t mirror0 = getenv("MATCHY_AMD_MIRROR_RECS") ? (uint32_t)atoi(getenv("MATCHY_AMD_MIRROR_RECS")) : (1u << 20);
        ensure_mirror(std::max(mirror0, 16u), std::max(mirror0 / 16, 16u));
        if (compact_) ensure_mirror_c4(std::max(mirror0, 16u));
        mirror_used_ = true;
    }
    if (compact_ && c4_.n < final_.n) c4_.alloc(final_.n);
    // databases with globs, one slice, forked: k_validate_dom queues the candidates it flags for the glob pass itself (MATCHY_AMD_NO_EARLY_GLOB=1: off)
    static const bool env_no_early = getenv("MATCHY_AMD_NO_EARLY_GLOB") != nullptr;
    early_glob_ = !no_fork && ns == 1 && lookup && ddb_->view.has_glob && !env_no_early;
    for (int k = 0; k < ns; ++k) slice_params(k, dptr, len, cuts[k], cuts[k + 1], lookup, host_mirror, launch_[k]);
    early_glob_ = early_glob_ && launch_[0].lp.early_glob != 0;
    // the counter blocks are zero already when the last scan ended with fetch() (k_finish copies them out and clears them)
    if (!counters_clean_) { MXY_HIP(hipMemsetAsync(counters_.p, 0, sizeof(ScanCounters) * MAX_SLICES, stream)); dom_preset_ = 0; }
    counters_clean_ = false;
    {
        // First chunks of the domain list without a reservation (DomWriter::reserve): when the previous batch needed more than one chunk per
        // wave on average, wave w of k_anchor owns chunk w and the counter starts behind those chunks — k_finish of the previous scan has
        // written that value already when the batches are alike; otherwise it is set here.
        static const bool env_no_static = getenv("MATCHY_AMD_NO_DOM_STATIC") != nullptr;
        const uint64_t total = (uint64_t)launch_[0].grid_anchor * 4u * ANCHOR_CHUNK;
        uint32_t want = 0;
        if (ns == 1 && !env_no_static && total <= launch_[0].tp.dom_cap && (uint64_t)hint_.n_dom * 2 >= total * 3) want = (uint32_t)total;
        if (want != dom_preset_) MXY_HIP(hipMemsetD32Async((hipDeviceptr_t)&counters_.p->n_dom, (int)want, 1, stream));
        dom_preset_ = want;
        dom_want_ = want;
        launch_[0].tp.dom_static = want ? ANCHOR_CHUNK : 0u;
        // behind the first chunk: an eighth of what a wave wrote last time, in whole tiles, 256..ANCHOR_CHUNK slots (less padding for k_validate_dom
        // to read; these reservations are spread over the kernel)
        static const int env_chunk = getenv("MATCHY_AMD_DOM_CHUNK") ? atoi(getenv("MATCHY_AMD_DOM_CHUNK")) : 0;
        for (int k = 0; k < ns; ++k) launch_[k].tp.dom_chunk = ANCHOR_CHUNK;
        if (want) {
            const uint32_t share = hint_.n_dom / ((uint32_t)launch_[0].grid_anchor * 4u) / 8u;
            launch_[0].tp.dom_chunk = env_chunk > 0 ? (uint32_t)env_chunk : std::min<uint32_t>(ANCHOR_CHUNK, std::max<uint32_t>(256u, share & ~63u));
        }
    }
    const bool rare_possible = (flags_ & (EX_HASHES | EX_BITCOIN | EX_ETHEREUM | EX_MONERO)) != 0;
    static const int misc_wgs = getenv("MATCHY_AMD_MISC_GRID") ? atoi(getenv("MATCHY_AMD_MISC_GRID")) : 0;
    const DevDb& view = ddb_->view;
    if (ns > 1) {
        // Sliced scan. The scan's stream carries the k_anchor launches, slice after slice. Behind k_anchor of slice i three
        // side streams take over, as in the one-slice fork below: dom_stream_ k_validate_dom -> k_validate (undecided
        // domains) -> k_lookup; aux_stream_ k_lookup_ip; aux2_stream_ k_validate (tokens, IPv6 / e-mail) -> k_rare. They
        // run BESIDE k_anchor of slice i + 1 (whose workgroups leave the CUs one by one), the slices of one side stream in
        // stream order. Only the tail of the last slice is not covered by a k_anchor.
        last_forked_ = true;
        if (!dom_stream_) {
            MXY_HIP(hipStreamCreateWithFlags(&dom_stream_, hipStreamNonBlocking));
            MXY_HIP(hipEventCreateWithFlags(&ev_join3_, hipEventDisableTiming));
            MXY_HIP(hipEventCreateWithFlags(&ev_dom_, hipEventDisableTiming));
        }
        if (!ev_anchor_[0]) {
            for (auto& e : ev_anchor_) MXY_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto& e : ev_misc_) MXY_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        if (profile_) MXY_HIP(hipEventRecord(ev_[0], stream));
        for (int k = 0; k < ns; ++k) {
            SliceLaunch& L = launch_[k];
            launch_anchor(L.tp, view, L.grid_anchor, stream);
            hipEvent_t done = ev_anchor_[k];
            if (profile_ && k == ns - 1) { MXY_HIP(hipEventRecord(ev_[1], stream)); done = ev_[1]; }
            else MXY_HIP(hipEventRecord(done, stream));
            if (L.ip_pass) {
                MXY_HIP(hipStreamWaitEvent(aux_stream_, done, 0));
                launch_lookup_ip(L.la, view, L.ip_grid, L.ip_dense, aux_stream_);
            }
            MXY_HIP(hipStreamWaitEvent(aux2_stream_, done, 0));
            TokParams t1 = L.tp;
            t1.vmode = 5u;
            if (L.split_misc) { t1.cands = work_[k].cands_m.p; t1.cand_cap = (uint32_t)work_[k].cands_m.n; t1.n_cand = &(counters_.p + k)->n_cand_m; }
            launch_validate_misc(t1, view, misc_wgs > 0 ? misc_wgs : std::max(1, n_cu_ / 2), aux2_stream_);
            if (L.split_misc) launch_lookup(L.lm, view, std::max(1, n_cu_ / 2), aux2_stream_);
            if (rare_possible) {
                TokParams tr = t1;
                tr.vmode = L.tp.vmode;
                if (L.split_misc) { tr.cands = work_[k].cands_r.p; tr.cand_cap = (uint32_t)work_[k].cands_r.n; tr.n_cand = &(counters_.p + k)->n_cand_r; }
                launch_rare(tr, view, n_cu_ * 4, aux2_stream_);
                if (L.split_misc) launch_lookup(L.lr, view, std::max(1, n_cu_ / 8), aux2_stream_);
            }
            MXY_HIP(hipEventRecord(ev_misc_[k], aux2_stream_));
            MXY_HIP(hipStreamWaitEvent(dom_stream_, done, 0));
            launch_validate_dom(L.tp, view, n_cu_ * L.gm[1], dom_stream_);
            TokParams t2 = L.tp;
            t2.vmode = 2u;
            launch_validate_misc(t2, view, misc_wgs > 0 ? misc_wgs : n_cu_, dom_stream_);
            if (!L.split_misc) MXY_HIP(hipStreamWaitEvent(dom_stream_, ev_misc_[k], 0));
            launch_lookup(L.lp, view, n_cu_ * L.gm[2], dom_stream_);
        }
        MXY_HIP(hipEventRecord(ev_join3_, dom_stream_));
        MXY_HIP(hipStreamWaitEvent(stream, ev_join3_, 0));
        if (launch_[0].split_misc) MXY_HIP(hipStreamWaitEvent(stream, ev_misc_[ns - 1], 0));
        bool any_ip = false;
        for (int k = 0; k < ns; ++k) any_ip |= launch_[k].ip_pass;
        if (any_ip) {
            MXY_HIP(hipEventRecord(ev_join_, aux_stream_));
            MXY_HIP(hipStreamWaitEvent(stream, ev_join_, 0));
        }
        if (profile_) MXY_HIP(hipEventRecord(ev_[4], stream));
        return;
    }
    SliceLaunch& L = launch_[0];
    const TokParams& tp = L.tp;
    if (profile_) MXY_HIP(hipEventRecord(ev_[0], stream));
    launch_anchor(tp, view, L.grid_anchor, stream);
    if (profile_) MXY_HIP(hipEventRecord(ev_[1], stream));
    // Fork. Everything k_anchor lists is complete now, and two parts of the rest do not depend on k_validate_dom:
    //  * the trie lookups of the IPv4 candidates (and the PCIe writes of their hit records, which is most of the result traffic
    //    of a log scan) -> k_lookup_ip on a stream of its own; both lookup passes append to the same final arrays (atomic slot
    //    reservation);
    //  * k_validate over the long tokens and k_anchor's IPv6 / e-mail anchors, and k_rare behind it -> a third stream.
    // The scan's stream keeps k_validate_dom, k_validate over the domain anchors k_validate_dom left undecided, and the lookups
    // of the validation kernels' candidates; it joins the third stream before those lookups and the second after them.
    last_forked_ = !no_fork;
    if (!no_fork) {
        hipEvent_t fork = profile_ ? ev_[1] : ev_fork_;
        if (!profile_) MXY_HIP(hipEventRecord(ev_fork_, stream));
        if (L.ip_pass) {
            MXY_HIP(hipStreamWaitEvent(aux_stream_, fork, 0));
            launch_lookup_ip(L.la, view, L.ip_grid, L.ip_dense, aux_stream_);
            MXY_HIP(hipEventRecord(ev_join_, aux_stream_));
        }
        MXY_HIP(hipStreamWaitEvent(aux2_stream_, fork, 0));
        TokParams t1 = tp;
        t1.vmode = 5u;
        if (L.split_misc) { t1.cands = work_[0].cands_m.p; t1.cand_cap = (uint32_t)work_[0].cands_m.n; t1.n_cand = &counters_.p->n_cand_m; }
        // split lists: the lookups of k_validate's candidates do not wait for k_rare (checksum validators: a chain of their own,
        // almost always over next to nothing), whose few candidates get a list and a lookup launch of their own — on the second
        // stream when the IPv4 lookups do not need it (they run inside k_anchor), beside the third stream's lookups. The long tokens
        // — k_rare takes what they leave — then get a launch of their own in front of the rare anchors (IPv6 / e-mail: ten times as
        // many, 0.12 ms), so that k_rare runs beside those instead of behind them.
        const bool rare_own = L.split_misc && rare_possible && !L.ip_pass;
        // Long lists (dense logs): these passes wait for memory most of their time, and beyond a few workgroups per CU more of them only
        // queue up in front of it. Measured per pass on 1.85 GB batches (tools/sweep_misc_grid.sh, profiles/r05_log_shapes.txt): long tokens
        // 8 -> 6 per CU: step 2.61 -> 2.41 ms (endpoint log with hashes); '@' / "::" anchors 8 -> 4: 2.37 -> 2.31 (JSON lines); domains
        // k_validate_dom left undecided 8 -> 2: 2.36 -> 2.22 (proxy log with URLs). Short lists keep their default grids.
        constexpr int TOK_WGS_PER_CU = 6, RARE_WGS_PER_CU = 4, RARE_DOM_WGS_PER_CU = 2;
        // Both lists long (the previous batch had hundreds of thousands of long tokens AND of IPv6 / e-mail anchors: application logs in
        // JSON lines): the long tokens do not run in front of the rare anchors on the third stream but beside them on the second, in front
        // of k_rare (which takes what they leave: stream order instead of an event), and list their hashes in k_rare's candidate list — the
        // lookup behind k_rare takes both. Two latency-bound passes over millions of entries each then overlap instead of queueing (the
        // JSON-lines shape: tail 1.18 -> see profiles/r04_log_shapes.txt). With short lists (a web-server log) the two kernels would only
        // take vector-ALU time from k_validate_dom beside them (+6 us measured): the order stays as it is there.
        static const int env_tok_aside = getenv("MATCHY_AMD_TOK_ASIDE") ? atoi(getenv("MATCHY_AMD_TOK_ASIDE")) : -1;
        const bool tok_aside = rare_own && (env_tok_aside >= 0 ? env_tok_aside != 0 : (hint_.n_tok >= 262144u && hint_.n_rare >= 262144u));
        if (tok_aside) {
            MXY_HIP(hipStreamWaitEvent(aux_stream_, fork, 0));
            TokParams tt = t1;
            tt.vmode = 4u;
            tt.cands = work_[0].cands_r.p; tt.cand_cap = (uint32_t)work_[0].cands_r.n; tt.n_cand = &counters_.p->n_cand_r;
            launch_validate_misc(tt, view, misc_wgs > 0 ? misc_wgs : grid_for(hint_.n_tok, 1024, std::max(1, n_cu_ / 2), TOK_WGS_PER_CU), aux_stream_);
            t1.vmode = 1u;
        } else if (rare_own) {
            TokParams tt = t1;
            tt.vmode = 4u;
            launch_validate_misc(tt, view, misc_wgs > 0 ? misc_wgs : grid_for(hint_.n_tok, 1024, std::max(1, n_cu_ / 2), TOK_WGS_PER_CU), aux2_stream_);   // 6 KB of LDS, workgroups beyond the list leave at once
            if (!ev_v1_) MXY_HIP(hipEventCreateWithFlags(&ev_v1_, hipEventDisableTiming));
            MXY_HIP(hipEventRecord(ev_v1_, aux2_stream_));
            t1.vmode = 1u;
        }
        // beside k_validate_dom: half the CUs, so that kernel keeps most of its resident waves (more when the previous batch's lists were long)
        launch_validate_misc(t1, view, misc_wgs > 0 ? misc_wgs : grid_for(rare_own ? hint_.n_rare : std::max(hint_.n_rare, hint_.n_tok), 1024, std::max(1, n_cu_ / 2), rare_own ? RARE_WGS_PER_CU : 8), aux2_stream_);
        // Split lists: the side chains do not join the scan's stream through events — the last kernel of each (a k_lookup launch) reports
        // its end in ScanCounters::chains_done, which k_finish polls (arrive_chain 1: third stream, 2: k_rare's, 3: the fourth stream)
        static const bool env_join = getenv("MATCHY_AMD_EVENT_JOIN") != nullptr;
        const bool arrive = L.split_misc && !env_join;
        uint32_t chains = 0;
        if (L.split_misc) {
            LookupParams lm = L.lm;
            if (arrive && (rare_own || !rare_possible)) { lm.arrive_chain = 1; lm.arrive = counters_.p; ++chains; }
            static const int lm_wgs = getenv("MATCHY_AMD_LMGRID") ? atoi(getenv("MATCHY_AMD_LMGRID")) : 0;   // experiments (tools/sweep_tail_knobs.sh)
            launch_lookup(lm, view, lm_wgs > 0 ? lm_wgs : grid_for(hint_.n_cand_m, 512, std::max(1, n_cu_ / 2), 4), aux2_stream_);
        }
        if (rare_possible) {   // one wave per SIMD (297 VGPRs)
            TokParams tr = t1;
            tr.vmode = tp.vmode;
            if (L.split_misc) { tr.cands = work_[0].cands_r.p; tr.cand_cap = (uint32_t)work_[0].cands_r.n; tr.n_cand = &counters_.p->n_cand_r; }
            hipStream_t rs = rare_own ? aux_stream_ : aux2_stream_;
            if (rare_own && !tok_aside) MXY_HIP(hipStreamWaitEvent(aux_stream_, ev_v1_, 0));
            launch_rare(tr, view, grid_for(hint_.n_heavy, 128, n_cu_ * 4, 16), rs);
            if (L.split_misc) {
                LookupParams lr = L.lr;
                if (arrive) { lr.arrive_chain = rare_own ? 2u : 1u; lr.arrive = counters_.p; ++chains; }
                launch_lookup(lr, view, grid_for(hint_.n_cand_r, 512, std::max(1, n_cu_ / 8), 4), rs);   // (tok_aside: the hint follows the list that now holds the hashes too)
            }
            if (rare_own && !arrive) MXY_HIP(hipEventRecord(ev_join_, aux_stream_));
        }
        if (!arrive) MXY_HIP(hipEventRecord(ev_join2_, aux2_stream_));
        launch_validate_dom(tp, view, n_cu_ * L.gm[1], stream);
        TokParams t2 = tp;
        t2.vmode = 2u;
        if (L.split_misc) {
            // Behind k_validate_dom two chains: the few domain anchors it left undecided (general walk + lookups of THEIR candidates,
            // a list of their own) stay on the scan's stream — no event between producer and consumer —, the lookups of
            // k_validate_dom's candidates go to a fourth stream beside them. That stream then takes the other side streams' end
            // events in, so the scan's stream joins ONE event (every event wait on it is ~10 us in front of k_finish).
            if (!dom_stream_) {
                MXY_HIP(hipStreamCreateWithFlags(&dom_stream_, hipStreamNonBlocking));
                MXY_HIP(hipEventCreateWithFlags(&ev_join3_, hipEventDisableTiming));
                MXY_HIP(hipEventCreateWithFlags(&ev_dom_, hipEventDisableTiming));
            }
            MXY_HIP(hipEventRecord(ev_dom_, stream));
            MXY_HIP(hipStreamWaitEvent(dom_stream_, ev_dom_, 0));
            {
                LookupParams lpm = L.lp;
                if (arrive) { lpm.arrive_chain = 3; lpm.arrive = counters_.p; ++chains; }
                static const int lp_wgs = getenv("MATCHY_AMD_LPGRID") ? atoi(getenv("MATCHY_AMD_LPGRID")) : 0;
                // one workgroup per CU: every workgroup ends with a pair of returning atomics on the two record counters, and with 512
                // of them those queue up behind each other (64 / 128 / 192 / 256 / 512 workgroups: tail 0.236 / 0.218 / 0.217 / 0.218 / 0.227 ms;
                // not fewer than one per CU: a database that most names hit makes this the kernel with the work)
                launch_lookup(lpm, view, lp_wgs > 0 ? lp_wgs : grid_for(hint_.n_cand, 512, n_cu_, 4), dom_stream_);
            }
            if (arrive) expect_chains_ = chains;
            else {
                MXY_HIP(hipStreamWaitEvent(dom_stream_, ev_join2_, 0));
                if (rare_own) MXY_HIP(hipStreamWaitEvent(dom_stream_, ev_join_, 0));
                MXY_HIP(hipEventRecord(ev_join3_, dom_stream_));
            }
            t2.cands = work_[0].cands_d.p; t2.cand_cap = (uint32_t)work_[0].cands_d.n; t2.n_cand = &counters_.p->n_cand_d;
            launch_validate_misc(t2, view, misc_wgs > 0 ? misc_wgs : grid_for(hint_.n_rare_dom, 512, n_cu_, RARE_DOM_WGS_PER_CU), stream);
            static const int ld_wgs = getenv("MATCHY_AMD_LDGRID") ? atoi(getenv("MATCHY_AMD_LDGRID")) : 0;
            launch_lookup(L.ld, view, ld_wgs > 0 ? ld_wgs : grid_for(hint_.n_cand_d, 512, std::max(1, n_cu_ / 8), 4), stream);
        } else {
            if (early_glob_) {
                // Databases with globs keep one candidate list, but the candidates k_validate_dom flags for the glob pass — most of that
                // pass's work — are on a work list of their own already (TokParams::glob_work_d): the glob pass over them starts here, on the
                // fourth stream, beside k_validate<2> and the lean pass over everything else (which skips them and defers what IT finds to
                // the usual work list for a second, small glob pass behind it).
                if (!dom_stream_) {
                    MXY_HIP(hipStreamCreateWithFlags(&dom_stream_, hipStreamNonBlocking));
                    MXY_HIP(hipEventCreateWithFlags(&ev_join3_, hipEventDisableTiming));
                    MXY_HIP(hipEventCreateWithFlags(&ev_dom_, hipEventDisableTiming));
                }
                MXY_HIP(hipEventRecord(ev_dom_, stream));
                MXY_HIP(hipStreamWaitEvent(dom_stream_, ev_dom_, 0));
                // Both side chains report their ends to k_finish (arrival counters, as in the scans without globs: an event join in front of
                // k_finish costs ~10-20 us each).
                static const bool env_join_g = getenv("MATCHY_AMD_EVENT_JOIN") != nullptr;
                uint32_t chains = 0;
                LookupParams lg = L.lp;
                lg.glob_work = work_[0].glob_work_d.p; lg.glob_work_cap = (uint32_t)work_[0].glob_work_d.n;
                lg.n_work = &counters_.p->n_glob_work_d;
                if (!env_join_g) { lg.arrive_chain = 3; lg.arrive = counters_.p; ++chains; }
                launch_lookup_early_glob(lg, view, n_cu_ * L.gm[2], dom_stream_);
                if (env_join_g) MXY_HIP(hipEventRecord(ev_join3_, dom_stream_));
                glob_join3_ = env_join_g;
                // ... and the undecided domains (k_validate<2>: general walk) with the glob pass over THEIR candidates leave the scan's
                // stream for the second one, when the IPv4 lookups do not need it: the lean pass does not have to wait for them
                static const bool env_no_aside = getenv("MATCHY_AMD_NO_V2_ASIDE") != nullptr;
                glob_v2_aside_ = !L.ip_pass && !env_no_aside;
                if (glob_v2_aside_) {
                    MXY_HIP(hipStreamWaitEvent(aux_stream_, ev_dom_, 0));
                    t2.cands = work_[0].cands_d.p; t2.cand_cap = (uint32_t)work_[0].cands_d.n; t2.n_cand = &counters_.p->n_cand_d;
                    launch_validate_misc(t2, view, misc_wgs > 0 ? misc_wgs : grid_for(hint_.n_rare_dom, 512, n_cu_, 8), aux_stream_);
                    LookupParams ld = L.ld;
                    if (!env_join_g) { ld.arrive_chain = 2; ld.arrive = counters_.p; ++chains; }
                    launch_lookup(ld, view, grid_for(hint_.n_cand_d, 512, std::max(2, n_cu_ / 8), 4), aux_stream_);
                    if (env_join_g) MXY_HIP(hipEventRecord(ev_join_, aux_stream_));
                }
                expect_chains_ = chains;
            }
            if (!(early_glob_ && glob_v2_aside_)) launch_validate_misc(t2, view, misc_wgs > 0 ? misc_wgs : grid_for(hint_.n_rare_dom, 512, n_cu_, RARE_DOM_WGS_PER_CU), stream);
            // no timing events inside the forked tail: every packet between two kernels of the chain is ~6-8 us of it, and with
            // kernels running side by side the intervals would not be kernel times anyway (ScanTiming: validate_ms = the whole tail)
            MXY_HIP(hipStreamWaitEvent(stream, ev_join2_, 0));
        }
    } else {
        launch_validate_dom(tp, view, n_cu_ * L.gm[1], stream);
        launch_validate_misc(tp, view, misc_wgs > 0 ? misc_wgs : grid_for(std::max(std::max(hint_.n_tok, hint_.n_rare), hint_.n_rare_dom), 1024, n_cu_, 8), stream);   // vmode 3: every list
        if (profile_) MXY_HIP(hipEventRecord(ev_[2], stream));
        if (rare_possible) launch_rare(tp, view, grid_for(hint_.n_heavy, 128, n_cu_ * 4, 16), stream);
        if (profile_) MXY_HIP(hipEventRecord(ev_[3], stream));
    }
    if (lookup) {
        if (L.ip_pass && no_fork) launch_lookup_ip(L.la, view, L.ip_grid, L.ip_dense, stream);
        if (no_fork || !L.split_misc) launch_lookup(L.lp, view, n_cu_ * L.gm[2], stream);   // split lists: launched on the fourth stream above
        if (L.ip_pass && !no_fork) MXY_HIP(hipStreamWaitEvent(stream, ev_join_, 0));
        if (L.split_misc && !no_fork && !expect_chains_) MXY_HIP(hipStreamWaitEvent(stream, ev_join3_, 0));
        if (early_glob_ && !no_fork && glob_join3_) {
            MXY_HIP(hipStreamWaitEvent(stream, ev_join3_, 0));
            if (glob_v2_aside_) MXY_HIP(hipStreamWaitEvent(stream, ev_join_, 0));
        }
    }
    if (profile_) MXY_HIP(hipEventRecord(ev_[4], stream));
}

void Scanner::ensure_pinned(size_t bytes) {
    if (pinned_bytes_ >= bytes) return;
    if (pinned_) (void)hipHostFree(pinned_);
    pinned_bytes_ = bytes + bytes / 4 + (1 << 16);
    MXY_HIP(hipHostMalloc(&pinned_, pinned_bytes_, hipHostMallocDefault));
}

size_t sort_hits_temp_bytes(uint32_t n);
hipError_t sort_hits(const FinalHit* fin, uint32_t n, unsigned long long* keys, uint32_t* vals, void* temp, size_t temp_bytes, FinalHit* out,
                     hipStream_t stream);

// Wait for the stream. poll: the scan behind it is about a millisecond (device-resident input), and the runtime's blocking wait
// takes ~10 us longer to notice the end — poll the stream for the first milliseconds, then block. Host-buffer scans wait for tens
// of milliseconds of copies with other threads busy in the runtime (the command line's reader and second scanner): they block.
static void wait_stream(hipStream_t stream, bool poll) {
    static const bool env_block = getenv("MATCHY_AMD_BLOCKING_WAIT") != nullptr;
    if (poll && !env_block) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t e = hipStreamQuery(stream);
            if (e == hipSuccess) return;
            if (e != hipErrorNotReady) throw HipError{std::string("hipStreamQuery: ") + hipGetErrorString(e)};
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(4)) break;
        }
    }
    MXY_HIP(hipStreamSynchronize(stream));
}

void Scanner::fetch(ScanOutput& out, bool want_cands, hipStream_t stream, HitMode hit_mode, bool sorted) {
    const bool trace = !single_ && getenv("MATCHY_AMD_TRACE");
    MXY_HIP(hipSetDevice(ddb_->device));   // regrown buffers must land on this scanner's device whatever thread calls
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    const int ns = n_slices_;
    for (int attempt = 0;; ++attempt) {
        // k_finish: the counter blocks go to pinned host memory and are cleared on the device for the next scan
        launch_finish(counters_.p, host_slices_, ns, expect_chains_, dom_want_, stream);
        // side chains that report to k_finish end behind the last event scan_device recorded: the interval ends behind k_finish then
        if (profile_ && expect_chains_) MXY_HIP(hipEventRecord(ev_[4], stream));
        expect_chains_ = 0;   // a rescan sets it again; the spill pass below runs on this stream
        wait_stream(stream, last_fork_);
        if (host_slices_[0].error & 8u) {
            // k_finish gave up polling for the side chains (they are slow, not lost): the counters it copied are a snapshot and the
            // device copies were left alone. Join the side streams here, then take the counters again without a poll.
            counters_clean_ = false;
            if (trace) fprintf(stderr, "[matchy_amd] k_finish gave up polling for the side chains: joining the side streams on the host\n");
            if (aux_stream_) MXY_HIP(hipStreamSynchronize(aux_stream_));
            if (aux2_stream_) MXY_HIP(hipStreamSynchronize(aux2_stream_));
            if (dom_stream_) MXY_HIP(hipStreamSynchronize(dom_stream_));
            launch_finish(counters_.p, host_slices_, ns, 0u, dom_want_, stream);
            wait_stream(stream, false);
        }
        counters_clean_ = true;
        // slice 0 holds n_final / n_final_ids of all slices; statistics are summed
        ScanCounters& c = host_counters_;
        c = host_slices_[0];
        for (int k = 1; k < ns; ++k) {
            const ScanCounters& s = host_slices_[k];
            c.lines += s.lines; c.cand_true += s.cand_true; c.hits_true += s.hits_true; c.error |= s.error;
        }
        bool over = c.n_final > final_.n || c.n_final_ids > final_ids_.n || (compact_ && c.n_c4 > c4_.n);
        for (int k = 0; k < ns; ++k) {
            const ScanCounters& s = host_slices_[k];
            const Work& w = work_[k];
            over = over || s.n_cand > w.cands.n || s.n_cand_a > w.cands_a.n || s.n_cand_m > w.cands_m.n || s.n_cand_r > w.cands_r.n || s.n_cand_d > w.cands_d.n || s.n_rare > w.rare.n || s.n_rare_dom > w.rare_dom.n || s.n_tok > w.tok.n ||
                   s.n_heavy > w.heavy.n || (w.glob_work.n && s.n_glob_work > w.glob_work.n) || (early_glob_ && s.n_glob_work_d > w.glob_work_d.n) || s.n_hits > w.hits.n || s.n_ids > w.ids.n ||
                   s.n_dom > w.dom_slots || (w.spill.n && s.n_spill > w.spill.n);
        }
        if (!over) {
            // candidates the glob pass could not hold (more results / deeper star nesting than a lane stores): normally none.
            // If there are, the spill pass runs now — its scratch (one bit per pattern id per thread) is allocated only here — and
            // appends their records to the same arrays; the counters are read again afterwards.
            uint32_t n_spill = 0;
            for (int k = 0; k < ns; ++k) n_spill += work_[k].spill.n ? host_slices_[k].n_spill : 0u;
            if (n_spill == 0 || spill_done_ || !last_lookup_) break;
            const uint32_t threads = spill_threads();
            const size_t words = launch_[0].lp.spill_words;
            if (spill_scratch_.n < words * SPILL_BLOCKS * threads) spill_scratch_.alloc(words * SPILL_BLOCKS * threads);
            // the spill pass appends to the lists the counters describe: put them back (k_finish cleared them)
            MXY_HIP(hipMemcpyAsync(counters_.p, host_slices_, sizeof(ScanCounters) * ns, hipMemcpyHostToDevice, stream));
            counters_clean_ = false;
            for (int k = 0; k < ns; ++k) {
                if (!work_[k].spill.n || !host_slices_[k].n_spill) continue;
                LookupParams lp = launch_[k].lp;
                lp.spill_scratch = spill_scratch_.p;
                launch_lookup_spill(lp, ddb_->view, stream);
            }
            spill_done_ = true;
            if (trace) fprintf(stderr, "[matchy_amd] %u candidates to the spill pass\n", n_spill);
            continue;
        }
        if (trace) fprintf(stderr, "[matchy_amd] work buffers overflow (attempt %d): regrow and rescan\n", attempt);
        if (single_) throw HipError{"lookup_one: work buffers overflow"};
        if (attempt >= 5) throw HipError{"scan: work buffers still overflow after regrowing"};
        // grow and run again: the kernels count past the capacity without writing, so the counts are exact demands
        auto grown = [](uint32_t n) { return (size_t)n + n / 4 + 1024; };
        size_t recs = 0, ids = 0;
        for (int k = 0; k < ns; ++k) {
            const ScanCounters& s = host_slices_[k];
            Work& w = work_[k];
            if (s.n_cand > w.cands.n) w.cands.alloc(grown(s.n_cand));
            if (s.n_cand_a > w.cands_a.n) w.cands_a.alloc(grown(s.n_cand_a));
            if (s.n_cand_m > w.cands_m.n) w.cands_m.alloc(grown(s.n_cand_m));
            if (s.n_cand_r > w.cands_r.n) w.cands_r.alloc(grown(s.n_cand_r));
            if (s.n_cand_d > w.cands_d.n) w.cands_d.alloc(grown(s.n_cand_d));
            if (s.n_rare > w.rare.n) w.rare.alloc(grown(s.n_rare));
            if (s.n_rare_dom > w.rare_dom.n) w.rare_dom.alloc(grown(s.n_rare_dom));
            if (s.n_tok > w.tok.n) w.tok.alloc(grown(s.n_tok));
            if (s.n_heavy > w.heavy.n) w.heavy.alloc(grown(s.n_heavy));
            if (w.glob_work.n && s.n_glob_work > w.glob_work.n) w.glob_work.alloc(grown(s.n_glob_work));
            if (early_glob_ && s.n_glob_work_d > w.glob_work_d.n) w.glob_work_d.alloc(grown(s.n_glob_work_d));
            if (w.spill.n && s.n_spill > w.spill.n) w.spill.alloc(grown(s.n_spill));
            if (s.n_dom > w.dom_slots) {
                w.dom_slots = (((size_t)s.n_dom + s.n_dom / 4 + ANCHOR_CHUNK) / ANCHOR_CHUNK) * ANCHOR_CHUNK;
                w.dom_list.alloc(w.dom_slots * DOM_PLANES);
            }
            if (s.n_hits > w.hits.n || w.hits.n < w.cands.n / 4) w.hits.alloc(std::max<size_t>(grown(s.n_hits), w.cands.n / 4));
            if (s.n_ids > w.ids.n) w.ids.alloc(grown(s.n_ids));
            recs += w.hits.n; ids += w.ids.n;
        }
        if (final_.n < recs || c.n_final > final_.n) final_.alloc(std::max<size_t>(recs, grown(c.n_final)));
        if (final_ids_.n < recs + ids || c.n_final_ids > final_ids_.n) {
            const size_t want = std::max<size_t>(recs + ids, grown(c.n_final_ids));
            final_ids_.alloc(want); final_offs_.alloc(want);
        }
        if (compact_ && c.n_c4 > c4_.n) c4_.alloc(grown(c.n_c4));
        scan_device(last_ptr_, last_len_, last_lookup_, stream, last_mirror_, last_fork_, last_slices_, last_compact_);
    }
    {
        ListHint hh;
        for (int k = 0; k < ns; ++k) {
            const ScanCounters& q = host_slices_[k];
            hh.n_tok = std::max(hh.n_tok, q.n_tok); hh.n_rare = std::max(hh.n_rare, q.n_rare); hh.n_rare_dom = std::max(hh.n_rare_dom, q.n_rare_dom);
            hh.n_heavy = std::max(hh.n_heavy, q.n_heavy); hh.n_cand = std::max(hh.n_cand, q.n_cand); hh.n_cand_m = std::max(hh.n_cand_m, q.n_cand_m);
            hh.n_cand_r = std::max(hh.n_cand_r, q.n_cand_r); hh.n_cand_d = std::max(hh.n_cand_d, q.n_cand_d);
            hh.n_dom = std::max(hh.n_dom, q.n_dom);
        }
        hint_ = hh;
    }
    const ScanCounters& c = host_counters_;
    if (c.error & 1) throw HipError{"scan: a candidate matches more than 65535 glob patterns (the hit record counts pattern ids in 16 bits)"};
    if (c.error & 4) throw HipError{"scan: a candidate is longer than 16 MiB (24-bit length field)"};
    const double t_counters = since();
    if (trace) {
        ScanCounters t = host_slices_[0];
        for (int k = 1; k < ns; ++k) {
            const ScanCounters& s = host_slices_[k];
            t.n_dom += s.n_dom; t.n_rare += s.n_rare; t.n_rare_dom += s.n_rare_dom; t.n_tok += s.n_tok; t.n_heavy += s.n_heavy; t.n_cand_a += s.n_cand_a;
            t.n_cand += s.n_cand; t.n_hits += s.n_hits; t.n_ids += s.n_ids; t.n_glob_work += s.n_glob_work;
        }
        fprintf(stderr, "[matchy_amd] slices=%d lines=%llu n_dom=%u n_rare=%u+%u n_tok=%u n_heavy=%u n_cand=%u+%u (true %u) n_hits=%u (true %u) n_ids=%u glob_work=%u final=%u\n",
                ns, c.lines, t.n_dom, t.n_rare, t.n_rare_dom, t.n_tok, t.n_heavy, t.n_cand_a, t.n_cand, c.cand_true, t.n_hits, c.hits_true, t.n_ids, t.n_glob_work, c.n_final);
    }
    out.lines = c.lines; out.n_cand = single_ ? c.n_cand : c.cand_true;
    out.n_hits = !last_lookup_ ? 0 : (single_ ? c.hits_true : c.n_final + (compact_ ? c.n_c4 : 0u));
    if (profile_) {
        MXY_HIP(hipEventElapsedTime(&timing_.anchor_ms, ev_[0], ev_[1]));
        if (last_forked_) {   // one interval for everything behind k_anchor (kernels on three streams)
            MXY_HIP(hipEventElapsedTime(&timing_.validate_ms, ev_[1], ev_[4]));
            timing_.rare_ms = 0; timing_.lookup_ms = 0;
        } else {
            MXY_HIP(hipEventElapsedTime(&timing_.validate_ms, ev_[1], ev_[2]));
            MXY_HIP(hipEventElapsedTime(&timing_.rare_ms, ev_[2], ev_[3]));
            MXY_HIP(hipEventElapsedTime(&timing_.lookup_ms, ev_[3], ev_[4]));
        }
        MXY_HIP(hipEventElapsedTime(&timing_.total_ms, ev_[0], ev_[4]));
        timing_.slices = ns;
    }
    out.hits.clear(); out.ids.clear(); out.cands.clear();
    out.fin = nullptr; out.fin_ids = nullptr; out.fin_offs = nullptr; out.n_fin = 0; out.n_fin_ids = 0;
    out.c4 = nullptr; out.n_c4 = 0;
    const bool had_mirror = mirror_used_;
    Work& w0 = work_[0];   // raw hits and candidate lists are read by one-slice scans only (single queries, extraction)
    const bool get_raw = last_lookup_ && hit_mode == HITS_RAW && c.n_hits;
    const bool get_fin = last_lookup_ && hit_mode == HITS_FINAL && c.n_final;
    if ((get_raw || want_cands) && ns != 1) throw HipError{"fetch: raw hits / candidates of a sliced scan"};
    // D2H into pinned memory (pageable destinations run at a fraction of the PCIe rate)
    if (get_raw) {
        ensure_pinned((size_t)c.n_hits * sizeof(Hit));
        MXY_HIP(hipMemcpyAsync(pinned_, w0.hits.p, (size_t)c.n_hits * sizeof(Hit), hipMemcpyDeviceToHost, stream));
        if (c.n_ids) {
            out.ids.resize(c.n_ids);
            MXY_HIP(hipMemcpyAsync(out.ids.data(), w0.ids.p, (size_t)c.n_ids * 4, hipMemcpyDeviceToHost, stream));
        }
    }
    if (get_fin && !sorted && mirror_used_ && c.n_final <= mirror_cap_ && c.n_final_ids <= mirror_ids_cap_) {
        // k_lookup (pack_record) has already written the records into pinned host memory
        uint8_t* mb = (uint8_t*)mirror_;
        out.fin = (const FinalHit*)mb; out.n_fin = c.n_final;
        out.fin_ids = (const uint32_t*)(mb + (size_t)mirror_cap_ * sizeof(FinalHit));
        out.fin_offs = (const long long*)(mb + (size_t)mirror_cap_ * sizeof(FinalHit) + (((size_t)mirror_ids_cap_ * 4 + 7) & ~(size_t)7));
        out.n_fin_ids = c.n_final_ids;
    } else if (get_fin) {
        if (mirror_used_ && !sorted) {  // did not fit: take the copy path now, have a larger mirror next time
            const uint32_t want_r = std::max(mirror_cap_, c.n_final + c.n_final / 2), want_i = std::max(mirror_ids_cap_, c.n_final_ids + c.n_final_ids / 2);
            mirror_used_ = false;
            ensure_mirror(want_r, want_i);
        }
        const size_t hb = (size_t)c.n_final * sizeof(FinalHit), ib = (((size_t)c.n_final_ids * 4) + 7) & ~(size_t)7, ob = (size_t)c.n_final_ids * 8;
        ensure_pinned(hb + ib + ob);
        uint8_t* base = (uint8_t*)pinned_;
        const FinalHit* src = final_.p;
        if (sorted) {
            const uint32_t n = c.n_final;
            if (sort_keys_.n < 2 * (size_t)n) { sort_keys_.alloc(2 * (size_t)n + 2048); sort_vals_.alloc(2 * (size_t)n + 2048); final_sorted_.alloc((size_t)n + 1024); }
            const size_t tb = sort_hits_temp_bytes(n);
            if (sort_tmp_.n < tb) sort_tmp_.alloc(tb + tb / 4 + 4096);
            MXY_HIP(sort_hits(final_.p, n, sort_keys_.p, sort_vals_.p, sort_tmp_.p, tb, final_sorted_.p, stream));
            src = final_sorted_.p;
        }
        MXY_HIP(hipMemcpyAsync(base, src, hb, hipMemcpyDeviceToHost, stream));
        if (c.n_final_ids) {
            MXY_HIP(hipMemcpyAsync(base + hb, final_ids_.p, (size_t)c.n_final_ids * 4, hipMemcpyDeviceToHost, stream));
            MXY_HIP(hipMemcpyAsync(base + hb + ib, final_offs_.p, ob, hipMemcpyDeviceToHost, stream));
        }
        out.fin = (const FinalHit*)base; out.n_fin = c.n_final;
        out.fin_ids = (const uint32_t*)(base + hb); out.fin_offs = (const long long*)(base + hb + ib); out.n_fin_ids = c.n_final_ids;
    }
    if (last_lookup_ && hit_mode == HITS_FINAL && compact_ && c.n_c4) {
        if (had_mirror && !sorted && c.n_c4 <= mirror_c4_cap_) out.c4 = mirror_c4_;   // written by the kernels
        else {
            if (had_mirror) { mirror_used_ = false; ensure_mirror_c4(c.n_c4 + c.n_c4 / 2); }   // a larger mirror next time
            if (pinned_c4_n_ < c.n_c4) {
                if (pinned_c4_) (void)hipHostFree(pinned_c4_);
                pinned_c4_ = nullptr;
                pinned_c4_n_ = (size_t)c.n_c4 + c.n_c4 / 4 + 1024;
                MXY_HIP(hipHostMalloc((void**)&pinned_c4_, pinned_c4_n_ * sizeof(uint2), hipHostMallocDefault));
            }
            MXY_HIP(hipMemcpyAsync(pinned_c4_, c4_.p, (size_t)c.n_c4 * sizeof(uint2), hipMemcpyDeviceToHost, stream));
            out.c4 = pinned_c4_;
        }
        out.n_c4 = c.n_c4;
    }
    if (want_cands && c.n_cand + c.n_cand_a) {   // k_anchor's IPv4 list, then the validation kernels' list
        out.cands.resize((size_t)c.n_cand_a + c.n_cand);
        if (c.n_cand_a) MXY_HIP(hipMemcpyAsync(out.cands.data(), w0.cands_a.p, (size_t)c.n_cand_a * sizeof(Candidate), hipMemcpyDeviceToHost, stream));
        if (c.n_cand) MXY_HIP(hipMemcpyAsync(out.cands.data() + c.n_cand_a, w0.cands.p, (size_t)c.n_cand * sizeof(Candidate), hipMemcpyDeviceToHost, stream));
    }
    wait_stream(stream, last_fork_);
    if (trace) fprintf(stderr, "[matchy_amd] fetch: counters after %.3f ms, records after %.3f ms\n", t_counters, since());
    // drop the padding slots of partially filled chunks
    if (get_raw) {
        const Hit* ph = (const Hit*)pinned_;
        out.hits.reserve(c.hits_true);
        for (size_t r = 0; r < c.n_hits; ++r) if (ph[r].kind != 0xFF) out.hits.push_back(ph[r]);
    }
    if (!out.cands.empty()) {
        size_t w = 0;
        for (size_t r = 0; r < out.cands.size(); ++r) if (out.cands[r].len_type != 0xFFFFFFFFu) out.cands[w++] = out.cands[r];
        out.cands.resize(w);
    }
    for (auto& t : out.by_type) t = 0;
    if (want_cands) for (const Candidate& cd : out.cands) { uint32_t ty = cd.len_type >> 24; if (ty < IT_COUNT) out.by_type[ty]++; }
}

// Spill pass of the glob lookup (k_lookup_spill): list of candidate indices per slice; the per-thread scratch (one bit per
// pattern id and a star stack as deep as the longest pattern: ~128 MB for a million patterns) is allocated by fetch() when a
// scan has actually spilled.
void Scanner::setup_spill(int sl, LookupParams& lp) {
    if (!ddb_->view.has_glob) return;
    Work& w = work_[sl];
    const size_t words = (ddb_->view.pattern_count + 31) / 32 + 2 * ((size_t)ddb_->view.glob_max_segs + 1);
    const size_t want = std::max<size_t>(1024, w.cands.n / 256);
    if (w.spill.n < want) w.spill.alloc(want);
    lp.spill = w.spill.p; lp.spill_cap = (uint32_t)w.spill.n;
    lp.spill_scratch = nullptr; lp.spill_words = (uint32_t)words; lp.spill_blocks = SPILL_BLOCKS;
}

void Scanner::lookup_one(const std::string& text, Candidate c, ScanOutput& out) {
    MXY_HIP(hipSetDevice(ddb_->device));
    ensure_capacity(4096);
    Work& w = work_[0];
    if (staging_.n < text.size() + 16) staging_.alloc(text.size() + 4096);
    if (!text.empty()) MXY_HIP(hipMemcpy(staging_.p, text.data(), text.size(), hipMemcpyHostToDevice));
    c.start = 0;
    MXY_HIP(hipMemcpy(w.cands.p, &c, sizeof(c), hipMemcpyHostToDevice));
    ScanCounters z{};
    z.n_cand = 1;
    MXY_HIP(hipMemcpy(counters_.p, &z, sizeof(z), hipMemcpyHostToDevice));
    counters_clean_ = false;
    LookupParams lp{};
    lp.log = staging_.p; lp.len = (uint32_t)text.size(); lp.cands = w.cands.p; lp.cand_cap = (uint32_t)w.cands.n;
    lp.hits = w.hits.p; lp.hit_cap = (uint32_t)w.hits.n; lp.ids = w.ids.p; lp.ids_cap = (uint32_t)w.ids.n;
    lp.counters = counters_.p;
    setup_spill(0, lp);
    launch_lookup(lp, ddb_->view, 1, nullptr);
    launch_[0].lp = lp;   // fetch() launches the spill pass from here if the candidate spilled
    n_slices_ = 1; spill_done_ = false;
    last_lookup_ = true; last_ptr_ = staging_.p; last_len_ = (uint32_t)text.size();
    bool prof = profile_;
    profile_ = false;
    single_ = true;
    try { fetch(out, false, nullptr, HITS_RAW); } catch (...) { single_ = false; profile_ = prof; throw; }
    single_ = false;
    profile_ = prof;
}

void Scanner::scan_host(const uint8_t* data, size_t len, bool lookup, bool want_cands, ScanOutput& out, std::vector<uint64_t>* cand_bases,
                        std::vector<FinalHit>* fin, std::vector<uint32_t>* fin_ids, std::vector<long long>* fin_offs) {
    // Cut into < 2^30-byte pieces at newlines (N4 in SURVEY §8a: no candidate class admits '\n').
    const size_t MAXC = (size_t)1 << 30;
    // the calling thread may be another one than last time (its current device is thread state): allocations below must land
    // on this scanner's device
    MXY_HIP(hipSetDevice(ddb_->device));
    // copies and kernels of a host-buffer scan go to a stream of this scanner's own, so that several scanners on one
    // device (one per host thread, `matchy match --devices 0,0`) overlap one batch's transfer with another's kernels
    if (!host_stream_) MXY_HIP(hipStreamCreateWithFlags(&host_stream_, hipStreamNonBlocking));
    out = ScanOutput();
    size_t pos = 0;
    std::vector<Candidate> all_cands;
    if (cand_bases) cand_bases->clear();
    if (fin) { fin->clear(); fin_ids->clear(); fin_offs->clear(); }
    do {
        size_t n = std::min(MAXC, len - pos);
        if (pos + n < len) {
            const void* nl = memrchr(data + pos, '\n', n);
            if (!nl) throw HipError{"scan_host: a single line exceeds 1 GiB"};
            n = (const uint8_t*)nl - (data + pos) + 1;
        }
        if (staging_.n < n + 16) staging_.alloc(n + 16 + n / 8);
        // Pageable host memory (a mapped file, a heap buffer) reaches the device through the runtime's own pinned staging at
        // ~20 GB/s; pinned for the duration of the copy (hipHostRegister: ~5 ms per GB, tools/ubench/h2d_rate2.cpp) the DMA
        // engine reads the pages themselves at the bus rate (~57 GB/s). Only whole pages inside the piece are registered —
        // neighbouring pieces (other scanners working on the same mapping) never share a page of their registered ranges —
        // the few bytes in front of and behind them travel as they did. Memory that is pinned already, small pieces and
        // MATCHY_AMD_NO_REGISTER=1 take the plain copy.
        const bool trace_h = getenv("MATCHY_AMD_TRACE") != nullptr;
        const auto th0 = std::chrono::steady_clock::now();
        auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
        const uint8_t* src = data + pos;
        const uint8_t* reg_lo = nullptr;
        size_t reg_len = 0;
        bool mine = false;   // this call pinned the range (and unpins it)
        static const bool no_reg = getenv("MATCHY_AMD_NO_REGISTER") != nullptr;
        if (!no_reg && n >= ((size_t)4 << 20)) {
            const uintptr_t a = ((uintptr_t)src + 4095) & ~(uintptr_t)4095, b = ((uintptr_t)src + n) & ~(uintptr_t)4095;
            // pins::acquire: the whole range is covered by a pin of this library (the caller's, or another scanner's transient one —
            // its count goes up, so it stays pinned until this copy is done too), or it is pinned now. Anything else (memory someone
            // else pinned, a partial overlap, a failed registration) takes the one plain copy below.
            if (b > a && pins::acquire(a, b)) { reg_lo = (const uint8_t*)a; reg_len = b - a; mine = true; }
        }
        struct Unreg { uintptr_t a, b; ~Unreg() { if (b > a) pins::release(a, b); } } unreg{mine ? (uintptr_t)reg_lo : 0, mine ? (uintptr_t)reg_lo + reg_len : 0};
        const double t_reg = ms_since(th0);
        if (reg_lo) {
            const size_t head = (size_t)(reg_lo - src), tail = n - head - reg_len;
            if (head) MXY_HIP(hipMemcpyAsync(staging_.p, src, head, hipMemcpyHostToDevice, host_stream_));
            MXY_HIP(hipMemcpyAsync(staging_.p + head, reg_lo, reg_len, hipMemcpyHostToDevice, host_stream_));
            if (tail) MXY_HIP(hipMemcpyAsync(staging_.p + head + reg_len, reg_lo + reg_len, tail, hipMemcpyHostToDevice, host_stream_));
        } else if (n) MXY_HIP(hipMemcpyAsync(staging_.p, src, n, hipMemcpyHostToDevice, host_stream_));
        scan_device(staging_.p, (uint32_t)n, lookup, host_stream_);
        ScanOutput part;
        const double t_launch = ms_since(th0);
        fetch(part, want_cands, host_stream_, lookup && fin ? HITS_FINAL : HITS_NONE, true);   // synchronises the stream: the copy is done
        if (trace_h) {
            static const auto t_proc = std::chrono::steady_clock::now();
            fprintf(stderr, "[matchy_amd] t=%.1f ms scan_host piece %zu B: registered %zu B in %.3f ms, launches done after %.3f ms, results after %.3f ms\n",
                    ms_since(t_proc), n, reg_len, t_reg, t_launch, ms_since(th0));
        }
        out.lines += part.lines; out.n_cand += part.n_cand; out.n_hits += part.n_hits;
        for (int t = 0; t < IT_COUNT; ++t) out.by_type[t] += part.by_type[t];
        if (fin && part.n_fin) {
            const uint32_t id_shift = (uint32_t)fin_ids->size();
            for (size_t i = 0; i < part.n_fin; ++i) {
                FinalHit h = part.fin[i];
                h.start += (uint32_t)pos;   // len < 4 GiB is checked by the caller
                if (h.kind == 3) h.value += id_shift;
                fin->push_back(h);
            }
            fin_ids->insert(fin_ids->end(), part.fin_ids, part.fin_ids + part.n_fin_ids);
            fin_offs->insert(fin_offs->end(), part.fin_offs, part.fin_offs + part.n_fin_ids);
        }
        for (const Candidate& c : part.cands) { all_cands.push_back(c); if (cand_bases) cand_bases->push_back(pos); }
        pos += n;
    } while (pos < len);
    out.cands.swap(all_cands);
}

namespace pins {
namespace {
struct Entry { uintptr_t a, b; int refs; bool caller; };
std::mutex g_mutex;
std::vector<Entry> g_entries;
}  // namespace
bool acquire(uintptr_t a, uintptr_t b) {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (Entry& e : g_entries) {
        if (e.a <= a && b <= e.b) { ++e.refs; return true; }
        if (a < e.b && e.a < b) return false;   // partial overlap with a pinned range: registering would fail, a pinned-path copy would run over unpinned pages
    }
    const hipError_t err = hipHostRegister((void*)a, b - a, hipHostRegisterDefault);
    if (err != hipSuccess) { (void)hipGetLastError(); return false; }   // pinned by someone else (hipHostMalloc, a foreign registration) or not pinnable
    g_entries.push_back(Entry{a, b, 1, false});
    return true;
}
void release(uintptr_t a, uintptr_t b) {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (size_t i = 0; i < g_entries.size(); ++i) {
        Entry& e = g_entries[i];
        if (!(e.a <= a && b <= e.b) || e.refs <= 0) continue;
        if (--e.refs == 0 && !e.caller) {
            (void)hipHostUnregister((void*)e.a);
            (void)hipGetLastError();
            g_entries.erase(g_entries.begin() + (long)i);
        }
        return;
    }
}
int add_caller(const void* ptr, size_t bytes) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return -1; }
    g_entries.push_back(Entry{(uintptr_t)ptr, (uintptr_t)ptr + bytes, 1, true});   // the caller's own reference
    return 0;
}
void remove_caller(const void* ptr) {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (size_t i = 0; i < g_entries.size(); ++i) {
        if (g_entries[i].a != (uintptr_t)ptr || !g_entries[i].caller) continue;
        // the caller says the range is no longer in use (it must not unregister under its own scans): drop it whatever the count
        (void)hipHostUnregister(const_cast<void*>(ptr));
        (void)hipGetLastError();
        g_entries.erase(g_entries.begin() + (long)i);
        return;
    }
    (void)hipHostUnregister(const_cast<void*>(ptr));   // not registered through this library: as before
    (void)hipGetLastError();
}
}  // namespace pins

}  // namespace mxy
