// Host pipeline around the HIP kernels: PSL table, device-resident database image, scan sessions.
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "db_image.h"
#include "scan_types.h"

namespace mxy {

// kernel launch wrappers (k_anchor.hip, validate_kernels.hip, lookup_kernels.hip)
int anchor_blocks_per_cu();
int validate_blocks_per_cu(bool ac);
void launch_anchor(const TokParams& p, const DevDb& db, int grid, hipStream_t stream);
void launch_validate_dom(const TokParams& p, const DevDb& db, int grid, hipStream_t stream);
void launch_validate_misc(const TokParams& p, const DevDb& db, int grid, hipStream_t stream);
void launch_rare(const TokParams& p, const DevDb& db, int grid, hipStream_t stream);
void launch_lookup(const LookupParams& p, const DevDb& db, int grid, hipStream_t stream);
void launch_lookup_early_glob(const LookupParams& p, const DevDb& db, int grid, hipStream_t stream);
void launch_lookup_ip(const LookupParams& p, const DevDb& db, int grid, bool dense, hipStream_t stream);
// k_lookup_spill over p.spill (launched by Scanner::fetch when a scan has spilled candidates); threads = spill_threads() per block
void launch_lookup_spill(const LookupParams& p, const DevDb& db, hipStream_t stream);
uint32_t spill_threads();
// ip_tables.hip: the /24 table + /24 bitmap, and the leaf tables of the undecided /24s, built on the device from the uploaded tree
void launch_ip_l24(const uint2* nodes, uint32_t node_count, uint32_t v4_start, uint2* l24, uint32_t* bm24, uint32_t* n_undecided, hipStream_t s);
void launch_ip_leaf(const uint2* nodes, uint32_t node_count, uint2* l24, uint32_t* next, uint32_t n_leaf, uint32_t* leaf_node, uint2* leaf, hipStream_t s);

struct HipError { std::string what; };
#define MXY_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) throw ::mxy::HipError{std::string(#expr) + ": " + hipGetErrorString(_e)}; \
    } while (0)

// Public-suffix set loaded from the PSLB container (tools/gen_psl.py), shared by all handles.
struct PslHost {
    std::vector<std::string> suffixes;
    std::vector<PslSlot> slots;
    std::vector<uint8_t> pool;
    std::vector<uint32_t> bloom;
    std::vector<uint2> tld_tab;   // exact table of the last labels of <= 7 bytes (see DevDb::tld_tab)
    uint32_t mask = 0, max_tld_len = 0, max_suffix_len = 0;
    uint32_t tld_first[8] = {0};
    static const PslHost& get();  // throws std::runtime_error if the container cannot be found
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    void alloc(size_t count) {
        free();
        if (count) MXY_HIP(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
    }
    void upload(const std::vector<T>& v) { alloc(v.size()); if (!v.empty()) MXY_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); }
    void free() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    ~DevBuf() { free(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// Device-resident image of one database on one GPU (uploaded once, at first use).
struct DeviceDb {
    int device = 0;
    DevDb view{};
    DevBuf<uint2> ip_nodes, ip_l1, ip_l24, ip_leaf;
    DevBuf<uint32_t> ip_bm24, lit_bm, sfx_bm;
    DevBuf<uint2> tld_tab;
    DevBuf<uint32_t> dfa, dfa_node;
    DevBuf<uint8_t> dfa_cls;
    DevBuf<LitSlot> lit_slots;
    DevBuf<uint8_t> lit_pool, pg, psl_pool;
    DevBuf<uint32_t> lit2pat_off, lit2pat, bloom;
    DevBuf<uint32_t> lit_offsets, glob_offsets;  // pattern id -> data-section offset (pack_record)
    DevBuf<PslSlot> psl_slots;
    DevBuf<uint32_t> lc_map;
    DevBuf<uint2> lc_ign, lc_cased;
    uint32_t n_lit_offsets = 0, n_glob_offsets = 0;
    size_t bytes_uploaded = 0;
    void upload(const DbImage& img, int dev);
};

// HIP-event intervals of the last scan. One stream: k_anchor | k_validate_dom + k_validate | k_rare | lookups incl. record packing.
// Forked scan (three streams): anchor_ms as before, validate_ms = everything behind k_anchor, rare_ms = lookup_ms = 0.
// Sliced scan: anchor_ms = first k_anchor launch to the end of the last (the other slices' tails run beside), validate_ms = what is
// left behind the last k_anchor.
struct ScanTiming { float anchor_ms = 0, validate_ms = 0, rare_ms = 0, lookup_ms = 0, total_ms = 0; int slices = 1; };

enum HitMode { HITS_NONE = 0, HITS_FINAL = 1, HITS_RAW = 2 };

struct ScanOutput {
    std::vector<Candidate> cands;  // filled only when requested
    std::vector<Hit> hits;         // HITS_RAW (single-query path)
    std::vector<uint32_t> ids;
    // HITS_FINAL: dense records produced by pack_record, BORROWED from the scanner's pinned buffers
    // (valid until the next scan on that scanner)
    const FinalHit* fin = nullptr;
    const uint32_t* fin_ids = nullptr;
    const long long* fin_offs = nullptr;
    size_t n_fin = 0, n_fin_ids = 0;
    // compact IPv4 records of a scan with `compact` set (they are NOT in fin then), BORROWED like fin
    const uint2* c4 = nullptr;
    size_t n_c4 = 0;
    uint64_t lines = 0;
    uint32_t n_cand = 0, n_hits = 0;
    uint64_t by_type[IT_COUNT] = {0};
};

// Process-wide registry of the host ranges this library pinned (hipHostRegister): the caller's (matchy_amd_host_register, kept until
// matchy_amd_host_unregister) and the transient ones Scanner::scan_host pins around its copies. Reference-counted, so that a range two
// scanners copy from at the same time is unpinned when the LAST copy is done — never under a DMA another thread has in flight — and a
// range is only ever treated as pinned when an entry of the registry covers ALL of it. Memory pinned by someone else (hipHostMalloc,
// a foreign hipHostRegister) is not in the registry: scan_host copies from it with one plain hipMemcpyAsync and leaves the choice of
// path to the runtime, which knows.
namespace pins {
// returns true when [a, b) is pinned for the caller until release(a, b): covered by an entry (its count goes up) or registered now.
bool acquire(uintptr_t a, uintptr_t b);
void release(uintptr_t a, uintptr_t b);
// caller-owned entries (matchy_amd_host_register / _unregister); 0 = success
int add_caller(const void* ptr, size_t bytes);
void remove_caller(const void* ptr);
}

// A scan session: owns the per-launch work buffers on one device. Not thread-safe; create one per thread/stream.
class Scanner {
public:
    Scanner(std::shared_ptr<const DbImage> img, std::shared_ptr<DeviceDb> ddb, uint32_t extract_flags, uint32_t min_labels);
    ~Scanner();
    // Scan `len` bytes already resident in device memory (16-byte aligned). len < 2^31.
    // lookup=false stops after extraction. Results stay on the device until fetch().
    // host_mirror: pack_record also writes the final records into pinned host memory (unsorted fetches then need no copy)
    // fork: run the parts of the scan that do not depend on each other on the scanner's two extra streams (engine.cpp). For ONE
    // batch at a time on device-resident input; callers that keep several batches in flight (submit / wait) or whose time is the
    // host-to-device copy (scan_host) stay on one stream per scanner: the runtime multiplexes streams onto a few hardware queues,
    // and three scanners with three streams each ran 25 % slower than with one each.
    // slices > 1 (with fork): cut the batch into that many slices (see MAX_SLICES); 0 = the scanner's default for the batch size
    // compact: IPv4 results leave as 8-byte records (ScanOutput::c4, scan_types.h c4_pack) when compact_possible()
    void scan_device(const uint8_t* dptr, uint32_t len, bool lookup, hipStream_t stream, bool host_mirror = false, bool fork = false, int slices = 0,
                     bool compact = false);
    // the compact record holds data-section offsets of C4_DATA_BITS bits: every offset an IP result of this database can carry must fit
    bool compact_possible() const { return img_->max_ip_data_offset() < (1u << C4_DATA_BITS); }
    // Copies counters (and hits / candidates) back. Call after scan_device; synchronises the stream.
    // sorted: the final records are put into canonical order on the GPU (sort_hits.hip) before they are copied back
    void fetch(ScanOutput& out, bool want_cands, hipStream_t stream, HitMode hit_mode = HITS_FINAL, bool sorted = false);
    // One synthetic candidate (single-query API): `text` is uploaded, only the lookup kernel runs.
    // the dense hit records of the last lookup scan as the kernel left them in device memory (MATCHY_SCAN_FETCH_DEVICE)
    const FinalHit* device_final() const { return final_.p; }
    const uint32_t* device_final_ids() const { return final_ids_.p; }
    const long long* device_final_offs() const { return final_offs_.p; }
    size_t device_final_id_count() const { return host_counters_.n_final_ids; }
    void lookup_one(const std::string& text, Candidate c, ScanOutput& out);
    // Convenience: host buffer -> internal device buffer -> scan -> fetch (chunks of < 2^31 bytes).
    // `fin*` vectors receive owned copies of the final hits of all pieces, positions made absolute.
    void scan_host(const uint8_t* data, size_t len, bool lookup, bool want_cands, ScanOutput& out, std::vector<uint64_t>* cand_bases,
                   std::vector<FinalHit>* fin, std::vector<uint32_t>* fin_ids, std::vector<long long>* fin_offs);
    void set_profile(bool on) { profile_ = on; }
    // slices of the forked scan_device: 0 = default for the batch size (MATCHY_AMD_SLICES overrides), 1 = never cut, n = n equal slices
    void set_slices(int n) { slices_ = n < 0 ? 0 : n; }
    int slices() const { return slices_; }
    int last_slice_count() const { return n_slices_; }
    const ScanTiming& timing() const { return timing_; }
    uint32_t flags() const { return flags_; }
    int device() const { return ddb_->device; }
    const DbImage& image() const { return *img_; }

    // A device-resident batch of one synchronous scan is cut into up to MAX_SLICES slices (byte ranges on SEG_ALIGN boundaries, no
    // newline alignment needed: positions stay absolute and k_anchor looks across the cuts): k_anchor runs slice after slice on the
    // scan's stream, and everything behind it (validation, lookups, result traffic) runs per slice on three side streams, i.e.
    // BESIDE k_anchor of the following slices. Every slice has its own work lists and counters; the final records of all slices
    // go to the same arrays (slots reserved with atomics on the counters of slice 0).
    static constexpr int MAX_SLICES = 8;
    static constexpr uint32_t SPILL_BLOCKS = 16;

private:
    // work lists of one slice (grown on demand, reused between scans)
    struct Work {
        DevBuf<Candidate> cands, cands_a;   // candidates of the validation kernels / IPv4 candidates of k_anchor
        DevBuf<Candidate> cands_m, cands_r, cands_d; // forked scans of databases without globs: candidates of the third stream (tokens, IPv6, e-mail) / of k_rare
        DevBuf<RareAnchor> rare, rare_dom, tok, heavy;
        DevBuf<uint32_t> dom_list;
        size_t dom_slots = 0;
        DevBuf<Hit> hits;                   // single-query path and the spill pass only
        DevBuf<uint32_t> ids, glob_work;
        DevBuf<uint32_t> glob_work_d;       // work list k_validate_dom fills itself (forked scans of glob databases)
        DevBuf<uint32_t> spill;             // glob candidates beyond the per-lane storage of the glob pass (k_lookup_spill)
        void ensure(uint32_t len);
    };
    Work work_[MAX_SLICES];
    int n_slices_ = 1;                      // slices of the last scan_device
    int slices_ = 0;                        // set_slices()
    void ensure_capacity(uint32_t len);     // slice 0 + the final arrays, for a scan of `len` bytes in one slice
    void ensure_final(size_t recs, size_t ids);
    // kernel parameters and grids of one slice of the last scan_device (kept: the spill pass is launched from fetch())
    struct SliceLaunch {
        TokParams tp;
        LookupParams lp, la, lm, lr, ld;   // lookups of the validation kernels' candidates / of k_anchor's IPv4 candidates / of the third stream's lists
        bool split_misc = false;   // the third stream has a candidate list and a lookup pass of its own
        int gm[3] = {1, 1, 1};
        int grid_anchor = 1, ip_grid = 1;
        bool ip_pass = false, ip_dense = false;
    };
    SliceLaunch launch_[MAX_SLICES];
    void slice_params(int sl, const uint8_t* dptr, uint32_t len, uint32_t lo, uint32_t hi, bool lookup, bool host_mirror, SliceLaunch& L);
    int plan_slices(uint32_t len, int want, uint32_t (&cuts)[MAX_SLICES + 1]);
    void setup_spill(int sl, LookupParams& lp);
    bool spill_done_ = false;
    bool glob_join3_ = false, glob_v2_aside_ = false;   // early glob pass joined by event (MATCHY_AMD_EVENT_JOIN); k_validate<2> and its lookups on the second stream
    bool early_glob_ = false;       // last scan_device: the glob pass over k_validate_dom's flagged candidates runs beside the lean pass
    uint32_t expect_chains_ = 0;    // side-stream chains of the last scan_device that report their end to k_finish (0: event joins)
    // List sizes of the PREVIOUS scan of this scanner (fetch): the grids of the kernels behind the streaming pass are sized for the
    // sparse lists of a web-server log — half a workgroup per CU for the long tokens, an eighth for some lookups — and a log of another
    // shape (two file hashes per line: 18 M tokens) walked such a list with 500 iterations per lane. Batches of one input look alike,
    // so the next scan's grids follow the last scan's counts (grid_for); the first scan of a scanner runs with the defaults.
    struct ListHint { uint32_t n_tok = 0, n_rare = 0, n_rare_dom = 0, n_heavy = 0, n_cand = 0, n_cand_m = 0, n_cand_r = 0, n_cand_d = 0, n_dom = 0; } hint_;
    uint32_t dom_preset_ = 0;   // what ScanCounters::n_dom of slice 0 holds on the device while the counters are clean (k_finish presets it: TokParams::dom_static)
    uint32_t dom_want_ = 0;     // ... and what the scan in flight asked for
    int grid_for(uint32_t n_hint, uint32_t per_wg, int dflt, int max_per_cu) const;
    bool counters_clean_ = false;   // the device counter blocks are zero (k_finish of the last fetch left them so)
    int last_slices_ = 0;
    std::shared_ptr<const DbImage> img_;
    std::shared_ptr<DeviceDb> ddb_;
    hipStream_t host_stream_ = nullptr;   // scan_host: this scanner's own non-blocking stream
    uint32_t flags_, min_labels_;
    // the lookups of k_anchor's IPv4 candidates run on this stream beside the validation kernels (fork after k_anchor, join
    // before the counters are read back)
    hipStream_t aux_stream_ = nullptr;
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
    // a third stream: k_validate's part that does not depend on k_validate_dom (tokens, IPv6 / e-mail anchors) and k_rare behind it
    hipStream_t aux2_stream_ = nullptr;
    hipEvent_t ev_join2_ = nullptr;
    // sliced scans: the k_validate_dom -> k_validate -> k_lookup chain of every slice runs on a stream of its own too (the scan's
    // stream only carries the k_anchor launches), one "k_anchor of slice i is done" and one "misc stream of slice i is done" event
    // per slice
    hipStream_t dom_stream_ = nullptr;
    hipEvent_t ev_anchor_[MAX_SLICES] = {}, ev_misc_[MAX_SLICES] = {};
    hipEvent_t ev_join3_ = nullptr, ev_dom_ = nullptr, ev_v1_ = nullptr;
    DevBuf<FinalHit> final_;
    DevBuf<uint32_t> final_ids_;
    DevBuf<long long> final_offs_;
    DevBuf<unsigned long long> sort_keys_;
    DevBuf<uint32_t> sort_vals_;
    DevBuf<uint8_t> sort_tmp_;
    DevBuf<FinalHit> final_sorted_;
    DevBuf<ScanCounters> counters_;            // MAX_SLICES entries
    DevBuf<uint32_t> spill_scratch_;           // per-thread scratch of k_lookup_spill (allocated when a scan first spills)
    DevBuf<uint8_t> staging_;  // scan_host only
    // pinned mirror of the final records written by the lookup kernels themselves: FinalHit[mirror_cap_] | u32 ids[mirror_ids_cap_] | i64 offs[..]
    void* mirror_ = nullptr;
    uint32_t mirror_cap_ = 0, mirror_ids_cap_ = 0;
    bool mirror_used_ = false, last_mirror_ = false;
    void ensure_mirror(uint32_t recs, uint32_t ids);
    // compact IPv4 records (scan_device `compact`): device array, pinned mirror written by the kernels, pinned block of the copy path
    DevBuf<uint2> c4_;
    uint2* mirror_c4_ = nullptr;
    uint32_t mirror_c4_cap_ = 0;
    void ensure_mirror_c4(uint32_t recs);
    uint2* pinned_c4_ = nullptr;
    size_t pinned_c4_n_ = 0;
    bool compact_ = false, last_compact_ = false;   // of the last scan_device: in effect / as asked for
    void* pinned_ = nullptr;   // one pinned block: FinalHit[n] | u32 ids[m] | i64 offs[m]  (or Hit[n] for HITS_RAW)
    size_t pinned_bytes_ = 0;
    void ensure_pinned(size_t bytes);
    ScanCounters host_counters_{};             // the slices' counters summed (list counters: of slice 0 for one-slice scans)
    ScanCounters* host_slices_ = nullptr;      // pinned: MAX_SLICES counter blocks as read back
    uint32_t last_len_ = 0;
    const uint8_t* last_ptr_ = nullptr;
    bool last_lookup_ = false, last_fork_ = false, last_forked_ = false;
    bool single_ = false;
    bool profile_ = false;
    hipEvent_t ev_[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    ScanTiming timing_;
    int n_cu_ = 256;
};

}  // namespace mxy
