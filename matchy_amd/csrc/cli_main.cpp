// `matchy` command line for the MI355X build: the two subcommands of the reference CLI that sit on the hot path
// (crates/matchy/src/bin/matchy.rs:78-212):
//
//   matchy build <INPUT>... -o <FILE> [-f text|csv|json] [-t TYPE] [-d DESC] [--desc-lang LANG] [-i] [-v]
//   matchy match <DATABASE> <INPUT>... [--format json|summary] [-s] [--batch-bytes N] [--extractors LIST] [--device N | --devices LIST|all]
//   matchy query <DATABASE> <QUERY> [-q]                                  (bin/commands/query_cmd.rs)
//   matchy extract <INPUT>... [--format json|csv|text] [--types LIST] [--min-labels N] [-u] [-s] [--show-candidates]
//                                                                          (bin/commands/extract_cmd.rs)
//   matchy inspect <DATABASE> [-j] [-v]          matchy validate <DATABASE> [-l standard|strict] [-j]      (host only)
//
// `match` prints one JSON object per match on stdout (same records as the reference's parallel path,
// match_processor/parallel.rs:297-369: sorted keys, timestamp "0.000") and, with -s, the [INFO] statistics block on
// stderr (commands/match_cmd.rs:514-581). Every batch is scanned on the GPU through the C ABI (matchy_scanner_scan);
// there is no CPU scan path. Flags that only tune the reference's CPU thread pool (-j, --readers, --cache-size, -p,
// --debug-routing) are accepted and ignored; .gz inputs are decompressed on the host (zlib); -f/--follow polls the files.
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/matchy_amd.h"
#include "data_codec.h"
#include "db_builder.h"
#include "db_image.h"

using namespace mxy;

namespace {

int usage() {
    fprintf(stderr,
            "usage:\n"
            "  matchy build <INPUT>... -o <FILE> [-f text|csv|json] [-t TYPE] [-d DESC] [--desc-lang LANG] [-i] [-v]\n"
            "  matchy match <DATABASE> <INPUT>... [--format json|summary] [-s] [--batch-bytes N] [--extractors LIST] [--device N | --devices LIST|all] [-j N|auto] [-f]\n"
            "  matchy query <DATABASE> <QUERY> [-q]\n"
            "  matchy extract <INPUT>... [--format json|csv|text] [--types LIST] [--min-labels N] [-u] [-s] [--show-candidates]\n"
            "  matchy inspect <DATABASE> [-j] [-v]\n"
            "  matchy validate <DATABASE> [-l standard|strict] [-j]\n");
    return 2;
}

bool ends_with_ci(const std::string& s, const char* suf) {
    size_t n = strlen(suf);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; ++i) if (tolower((unsigned char)s[s.size() - n + i]) != suf[i]) return false;
    return true;
}

std::string fmt_num(unsigned long long v) {  // 1234567 -> "1,234,567" (bin/cli_utils.rs:143-153)
    std::string s = std::to_string(v), out;
    for (size_t i = 0; i < s.size(); ++i) {
        if (i && (s.size() - i) % 3 == 0) out.push_back(',');
        out.push_back(s[i]);
    }
    return out;
}

std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

// One CSV record (RFC 4180 quoting as the `csv` crate reads it by default: quotes, doubled quotes, embedded newlines).
bool read_csv_record(std::istream& in, std::vector<std::string>& out) {
    out.clear();
    std::string cell;
    bool quoted = false, any = false;
    int ch;
    while ((ch = in.get()) != EOF) {
        any = true;
        if (quoted) {
            if (ch == '"') {
                if (in.peek() == '"') { cell.push_back('"'); in.get(); }
                else quoted = false;
            } else cell.push_back((char)ch);
        } else if (ch == '"' && cell.empty()) quoted = true;
        else if (ch == ',') { out.push_back(cell); cell.clear(); }
        else if (ch == '\n') { if (!cell.empty() && cell.back() == '\r') cell.pop_back(); out.push_back(cell); return true; }
        else cell.push_back((char)ch);
    }
    if (!any) return false;
    out.push_back(cell);
    return true;
}

// CSV cell typing of the reference CLI (commands/build_cmd.rs:226-236): i64 -> Int32 (truncating), u64 -> Uint64,
// f64 -> Double, true/false -> Bool, else String.
DataValue csv_value(const std::string& v) {
    errno = 0;
    char* end = nullptr;
    if (!v.empty() && !isspace((unsigned char)v[0])) {
        long long i = strtoll(v.c_str(), &end, 10);
        if (errno == 0 && end && *end == 0 && (isdigit((unsigned char)v[0]) || ((v[0] == '-' || v[0] == '+') && v.size() > 1 && isdigit((unsigned char)v[1]))))
            return DataValue::Int32((int32_t)i);
        errno = 0;
        if (isdigit((unsigned char)v[0]) || (v[0] == '+' && v.size() > 1)) {
            unsigned long long u = strtoull(v.c_str(), &end, 10);
            if (errno == 0 && end && *end == 0) return DataValue::Uint64(u);
        }
        errno = 0;
        double f = strtod(v.c_str(), &end);
        if (errno == 0 && end && *end == 0 && end != v.c_str()) {
            // Rust's f64::from_str accepts decimal / exponent forms, "inf", "infinity", "nan" (any case); it has no hex floats
            bool hexish = v.find('x') != std::string::npos || v.find('X') != std::string::npos;
            if (!hexish) return DataValue::Double(f);
        }
    }
    if (v == "true" || v == "false") return DataValue::Bool(v == "true");
    return DataValue::String(v);
}

bool add_inputs(DatabaseBuilder& b, const std::vector<std::string>& inputs, const std::string& format, bool verbose, std::string& err) {
    size_t total = 0;
    for (const std::string& path : inputs) {
        std::ifstream in(path, std::ios::binary);
        if (!in) { err = "Failed to open input file: " + path; return false; }
        if (format == "text") {
            std::string line;
            bool first = true;
            while (std::getline(in, line)) {
                if (first) {
                    first = false;
                    size_t commas = 0;
                    for (char c : line) commas += c == ',';
                    if (commas >= 3)
                        fprintf(stderr, "Warning: %s looks like CSV but you specified --format text.\nIf this is a CSV file, use --format csv instead.\n", path.c_str());
                }
                std::string entry = trim(line);
                if (entry.empty() || entry[0] == '#') continue;
                if (!b.add_entry(entry, DataValue::Map())) { err = "Failed to add entry '" + entry + "': " + b.error(); return false; }
                ++total;
            }
        } else if (format == "csv") {
            std::vector<std::string> hdr, rec;
            if (!read_csv_record(in, hdr)) { err = "Failed to read CSV headers"; return false; }
            int entry_col = -1;
            for (size_t i = 0; i < hdr.size(); ++i) if (hdr[i] == "entry" || hdr[i] == "key") { entry_col = (int)i; break; }
            if (entry_col < 0) {
                err = "CSV must have an 'entry' or 'key' column. Found headers: ";
                for (size_t i = 0; i < hdr.size(); ++i) err += (i ? ", " : "") + hdr[i];
                return false;
            }
            size_t row = 1;
            while (read_csv_record(in, rec)) {
                ++row;
                if (rec.size() == 1 && rec[0].empty()) continue;  // blank line
                if ((size_t)entry_col >= rec.size()) { err = "Missing entry column at row " + std::to_string(row); return false; }
                DataValue data = DataValue::Map();
                for (size_t i = 0; i < hdr.size() && i < rec.size(); ++i)
                    if ((int)i != entry_col && !rec[i].empty()) data.map[hdr[i]] = csv_value(rec[i]);
                if (!b.add_entry(rec[entry_col], data)) {
                    err = "Failed to add entry '" + rec[entry_col] + "' at row " + std::to_string(row) + ": " + b.error();
                    return false;
                }
                ++total;
            }
        } else if (format == "json") {
            // [{"key": "...", "data": {...}}, ...] with the CLI's number typing (bin/cli_utils.rs:203-236)
            std::stringstream ss;
            ss << in.rdbuf();
            std::string text = ss.str(), perr;
            DataValue doc;
            if (!parse_json(text.data(), text.size(), NumberTyping::CLI, doc, perr)) { err = "Failed to parse JSON: " + perr; return false; }
            if (doc.type != DataValue::ARRAY) { err = "Failed to parse JSON: expected an array of {key, data} objects"; return false; }
            for (size_t i = 0; i < doc.arr.size(); ++i) {
                const DataValue& item = doc.arr[i];
                auto k = item.type == DataValue::MAP ? item.map.find("key") : item.map.end();
                if (item.type != DataValue::MAP || k == item.map.end() || k->second.type != DataValue::STRING) {
                    err = "Missing 'key' field at index " + std::to_string(i);
                    return false;
                }
                DataValue data = DataValue::Map();
                auto d = item.map.find("data");
                if (d != item.map.end()) {
                    if (d->second.type != DataValue::MAP) { err = "Expected JSON object for data at index " + std::to_string(i); return false; }
                    data = d->second;
                }
                if (!b.add_entry(k->second.str, data)) {
                    err = "Failed to add entry '" + k->second.str + "' at index " + std::to_string(i) + ": " + b.error();
                    return false;
                }
                ++total;
            }
        } else {
            err = "Unknown format: " + format + ". Use 'text', 'csv' or 'json'";
            return false;
        }
    }
    if (verbose) printf("  Total: %zu entries\n", total);
    return true;
}

int cmd_build(int argc, char** argv) {
    std::vector<std::string> inputs;
    std::string out, format = "text", dbtype, desc, lang = "en";
    bool verbose = false, case_insensitive = false;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&](const char* name) -> const char* { if (i + 1 >= argc) { fprintf(stderr, "error: %s needs a value\n", name); exit(2); } return argv[++i]; };
        if (a == "-o" || a == "--output") out = next("-o");
        else if (a == "-f" || a == "--format") format = next("-f");
        else if (a == "-t" || a == "--database-type") dbtype = next("-t");
        else if (a == "-d" || a == "--description") desc = next("-d");
        else if (a == "--desc-lang") lang = next("--desc-lang");
        else if (a == "-v" || a == "--verbose" || a == "--debug") verbose = true;
        else if (a == "-i" || a == "--case-insensitive") case_insensitive = true;
        else if (!a.empty() && a[0] == '-' && a != "-") { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); return 2; }
        else inputs.push_back(a);
    }
    if (inputs.empty() || out.empty()) return usage();
    DatabaseBuilder b(case_insensitive);
    if (!dbtype.empty()) b.set_database_type(dbtype);
    if (!desc.empty()) b.set_description(lang, desc);
    if (const char* e = getenv("MATCHY_BUILD_EPOCH")) b.set_build_epoch(strtoull(e, nullptr, 10));  // reproducible builds
    std::string err;
    if (!add_inputs(b, inputs, format, verbose, err)) { fprintf(stderr, "Error: %s\n", err.c_str()); return 1; }
    std::vector<uint8_t> blob;
    if (!b.build(blob)) { fprintf(stderr, "Error: %s\n", b.error().c_str()); return 1; }
    (void)chmod(out.c_str(), 0644);  // a previous build left the file read-only
    FILE* f = fopen(out.c_str(), "wb");
    if (!f || fwrite(blob.data(), 1, blob.size(), f) != blob.size()) { fprintf(stderr, "Error: Failed to write %s\n", out.c_str()); if (f) fclose(f); return 1; }
    fclose(f);
    (void)chmod(out.c_str(), 0444);  // the reference marks databases read-only (build_cmd.rs:15-35, 364-371)
    printf("\xE2\x9C\x93 Database built: %s\n", out.c_str());
    if (verbose) {
        const BuildStats& st = b.stats();
        printf("  IP entries: %zu, literals: %zu, globs: %zu, tree nodes: %u (%d-bit records), size: %zu bytes\n", st.ip_entries,
               st.literal_entries, st.glob_entries, st.node_count, st.record_size, blob.size());
    }
    return 0;
}

// --extractors=ip,domain / -crypto,-hash ... (bin/matchy.rs:124-131): returns the MATCHY_EXTRACT_* mask, 0 = auto
bool parse_extractors(const std::string& spec, uint32_t auto_mask, uint32_t& mask, std::string& err) {
    struct { const char* name; uint32_t bits; } names[] = {
        {"ipv4", MATCHY_EXTRACT_IPV4}, {"ipv6", MATCHY_EXTRACT_IPV6}, {"ip", MATCHY_EXTRACT_IPV4 | MATCHY_EXTRACT_IPV6}, {"ips", MATCHY_EXTRACT_IPV4 | MATCHY_EXTRACT_IPV6},
        {"domain", MATCHY_EXTRACT_DOMAINS}, {"domains", MATCHY_EXTRACT_DOMAINS}, {"email", MATCHY_EXTRACT_EMAILS}, {"emails", MATCHY_EXTRACT_EMAILS},
        {"hash", MATCHY_EXTRACT_HASHES}, {"hashes", MATCHY_EXTRACT_HASHES}, {"bitcoin", MATCHY_EXTRACT_BITCOIN}, {"ethereum", MATCHY_EXTRACT_ETHEREUM},
        {"monero", MATCHY_EXTRACT_MONERO}, {"crypto", MATCHY_EXTRACT_BITCOIN | MATCHY_EXTRACT_ETHEREUM | MATCHY_EXTRACT_MONERO},
    };
    uint32_t enable = 0, disable = 0;
    std::stringstream ss(spec);
    std::string tok;
    while (std::getline(ss, tok, ',')) {
        tok = trim(tok);
        if (tok.empty()) continue;
        bool neg = tok[0] == '-';
        std::string n = neg ? tok.substr(1) : tok;
        for (auto& c : n) c = (char)tolower((unsigned char)c);
        uint32_t bits = 0;
        for (auto& e : names) if (n == e.name) bits = e.bits;
        if (!bits) { err = "Unknown extractor: " + n; return false; }
        (neg ? disable : enable) |= bits;
    }
    mask = (enable ? enable : auto_mask) & ~disable;
    return true;
}

struct Totals {
    unsigned long long lines = 0, lines_with_matches = 0, matches = 0, candidates = 0, bytes = 0;
};

// ---- `matchy match`: reader -> batches -> one worker per device entry -> ordered printer
// The pipeline itself is the library's (matchy_multi_scanner_*: one worker thread and scanner per device entry, batches handed out in
// sequence, results taken back in sequence — processing/parallel.rs:494-505 behind the C ABI). What stays here is what belongs to
// the command line: the reader cuts every input into newline-aligned batches (FileReader::next_batch, processing/mod.rs:206-251;
// regular files are mapped, .gz / stdin are read), the batch hook renders a batch's matches on the worker that scanned it, and the
// printer emits the rendered batches in sequence order, so the output is the same for every device list (line blocks
// are independent, the database is replicated, the host gathers the hit records and sums the counters; no collective).
// bytes without the value-initialisation of std::vector (a 256 MiB batch buffer would be zeroed before every read)
struct RawBuf {
    std::unique_ptr<uint8_t[]> p;
    size_t cap = 0;
    explicit RawBuf(size_t n = 0) : p(n ? new uint8_t[n] : nullptr), cap(n) {}
    uint8_t* data() { return p.get(); }
    void grow(size_t n, size_t keep) {
        std::unique_ptr<uint8_t[]> q(new uint8_t[n]);
        if (keep) memcpy(q.get(), p.get(), keep);
        p = std::move(q); cap = n;
    }
};
// `ptr` points into `own` (inputs that are read: stdin, .gz) or into a file mapping that outlives the batch
struct Batch { size_t input = 0; RawBuf own; const uint8_t* ptr = nullptr; size_t len = 0; bool mapped = false;
               const void* reg = nullptr; };   // reg: page range of a mapped batch the reader pinned ahead of the scan (unpinned by the worker)
// what a batch contributes to the output: `text` / `text_len` is the buffer matchy_scan_result_to_ndjson returned (written to stdout as
// it is and released by the printer: 200 bytes per match are not copied again on the way), `out` what --follow builds from it
struct Done {
    std::string out; char* text = nullptr; size_t text_len = 0; Totals t; bool ok = true; size_t input = 0;
    Done() = default;
    Done(const Done&) = delete;
    Done& operator=(const Done&) = delete;
    ~Done() { if (text) matchy_free_string(text); }
};
// all of a buffer to stdout, past stdio (a gigabyte of NDJSON per ten gigabytes of log need not pass through its buffer)
static void write_all_stdout(const char* p, size_t n) {
    fflush(stdout);
    while (n) {
        const ssize_t w = write(1, p, n);
        if (w < 0) { if (errno == EINTR) continue; return; }
        p += w; n -= (size_t)w;
    }
}

struct MatchPipeline {
    matchy_multi_scanner_t* ms = nullptr;
    std::mutex mu;
    size_t submitted = 0;
    std::atomic<bool> reader_done{false};
    bool json = true;
    std::vector<std::string> sources;

    // the batch goes to the library's queue; its Batch object travels as the tag and comes back with the result
    size_t submit(Batch&& b) {
        Batch* hb = new Batch(std::move(b));
        size_t seq;
        { std::lock_guard<std::mutex> lk(mu); seq = submitted++; }
        if (matchy_multi_scanner_submit(ms, hb->ptr, hb->len, hb, hb->reg) != MATCHY_SUCCESS) {
            // not queued (cannot happen for batches below 4 GiB): the printer never sees it
            fprintf(stderr, "[ERROR] batch of %zu bytes rejected: %s\n", hb->len, matchy_amd_last_error());
            { std::lock_guard<std::mutex> lk(mu); rejected_inputs.push_back(hb->input); }   // its matches are missing: the input counts as failed
            if (hb->reg) matchy_amd_host_unregister(hb->reg);
            delete hb;
        }
        return seq;
    }
    // Mapped inputs: the mapping of a file is released by the printer as soon as the file's last batch has been printed — the
    // page tables of a file of gigabytes take tens of milliseconds to tear down, and that runs beside the scans of the next
    // files instead of behind the last one.
    // bytes of each input the first pass has read: where --follow starts watching (a size taken AFTER that pass would skip what was appended in between)
    std::vector<long long> consumed;
    std::vector<size_t> rejected_inputs;   // guarded by mu: inputs with a batch the library did not take
    struct Mapping { void* p; size_t len; size_t last_seq; bool released; };
    std::vector<Mapping> mappings;   // guarded by mu
    void add_mapping(void* p, size_t len, size_t last_seq) { std::lock_guard<std::mutex> lk(mu); mappings.push_back({p, len, last_seq, false}); }
    void release_mappings(size_t printed_below) {   // every batch with seq < printed_below is done
        std::vector<Mapping> go;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (Mapping& m : mappings) if (!m.released && m.last_seq < printed_below) { m.released = true; go.push_back(m); }
        }
        for (const Mapping& m : go) munmap(m.p, m.len);
    }
    // what one batch contributes to the output: counters, and (json) its matches rendered — on the worker thread that scanned it
    void render(matchy_scanner_t* sc, const matchy_scan_result_t& r, const uint8_t* data, size_t len, size_t input, Done& d) {
        d.input = input;
        Totals& t = d.t;
        t.lines += r.lines; t.candidates += r.candidates; t.bytes += len; t.matches += r.n_hits;
        // lines with matches: hits come sorted by offset; a new line starts when a '\n' lies between two hit starts
        const std::string& source = sources[input];
        size_t prev = (size_t)-1;
        for (size_t i = 0; i < r.n_hits; ++i) {
            const size_t s = (size_t)r.hits[i].start;
            if (prev == (size_t)-1 || memchr(data + prev, '\n', s - prev)) ++t.lines_with_matches;
            prev = s;
        }
        if (json && r.n_hits) {   // every match of the batch in one call (the scanner caches the rendered data payloads)
            char* text = nullptr;
            size_t n = 0;
            if (matchy_scan_result_to_ndjson(sc, &r, data, source.c_str(), &text, &n) == MATCHY_SUCCESS) { d.text = text; d.text_len = n; }
            else { fprintf(stderr, "[ERROR] rendering failed: %s\n", matchy_amd_last_error()); d.ok = false; }
        }
    }
    static void* batch_hook(void* user, size_t, matchy_scanner_t* sc, const matchy_scan_result_t* r, const uint8_t* data, size_t len, void* tag) {
        MatchPipeline* pl = (MatchPipeline*)user;
        Done* d = new Done();
        pl->render(sc, *r, data, len, ((const Batch*)tag)->input, *d);
        return d;
    }
    // scan one batch on the calling thread and render its matches (--follow: one scanner, batches as they appear)
    void run_batch(matchy_scanner_t* sc, Batch& b, Done& d) {
        d.input = b.input;
        if (!b.len) return;
        matchy_scan_result_t r;
        memset(&r, 0, sizeof(r));
        if (matchy_scanner_scan(sc, b.ptr, b.len, &r) != MATCHY_SUCCESS) {
            fprintf(stderr, "[ERROR] scan failed: %s\n", matchy_amd_last_error());
            d.ok = false;
            return;
        }
        render(sc, r, b.ptr, b.len, b.input, d);
        matchy_scan_result_free(&r);
    }
    // takes the batches back in sequence order and prints them, until the reader is done and nothing is pending
    void printer(Totals& total, std::vector<char>& input_failed) {
        for (;;) {
            matchy_multi_batch_t b;
            int32_t r = matchy_multi_scanner_next(ms, &b);
            if (r == 0) {
                if (!reader_done) { std::this_thread::sleep_for(std::chrono::microseconds(200)); continue; }   // between two submits
                r = matchy_multi_scanner_next(ms, &b);
                if (r == 0) return;
            }
            if (r != 1) { fprintf(stderr, "[ERROR] gathering results failed: %s\n", matchy_amd_last_error()); return; }
            Batch* hb = (Batch*)b.tag;
            Done* d = (Done*)b.payload;
            if (b.status != MATCHY_SUCCESS) {
                fprintf(stderr, "[ERROR] scan failed: %s\n", matchy_amd_last_error());
                input_failed[hb->input] = 1;
            } else if (d) {
                if (!d->ok) input_failed[hb->input] = 1;   // rendering failed in the batch hook: the batch's matches are missing, exit status says so
                if (d->text_len) write_all_stdout(d->text, d->text_len);
                total.lines += d->t.lines; total.lines_with_matches += d->t.lines_with_matches; total.matches += d->t.matches;
                total.candidates += d->t.candidates; total.bytes += d->t.bytes;
            }
            matchy_scan_result_free(&b.result);
            delete d;
            delete hb;   // frees an owned (read) batch's bytes; a mapped batch's pages go with its file's mapping
            release_mappings(b.seq + 1);
        }
    }
};

// Inputs ending in .gz (case-insensitive) are decompressed on the fly like the reference's file reader does
// (crates/matchy/src/file_reader.rs:45-75, by extension); "-" is stdin. Regular files are mapped and their batches are
// views of the mapping (no copy on the host; this thread pre-faults and pins each batch's pages ahead of its scan;
// MATCHY_AMD_NO_MMAP=1 reads them like a stream instead). Returns false when the input could not be read. Mappings are
// released by the printer as their files complete.
bool read_input(MatchPipeline& pl, size_t input, const std::string& path, size_t batch_bytes) {
    const bool gz = ends_with_ci(path, ".gz");
    int fd = path == "-" ? 0 : open(path.c_str(), O_RDONLY);
    if (fd < 0) { fprintf(stderr, "[ERROR] Failed to process %s: %s\n", path.c_str(), strerror(errno)); return false; }
    struct stat sb;
    if (!gz && fd != 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0 && !getenv("MATCHY_AMD_NO_MMAP")) {
        const size_t size = (size_t)sb.st_size;
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            close(fd);
            (void)madvise(m, size, MADV_SEQUENTIAL);
            const uint8_t* base = (const uint8_t*)m;
            size_t last_seq = 0;
            for (size_t pos = 0; pos < size;) {
                size_t end = std::min(size, pos + batch_bytes);
                if (end < size) {   // newline-aligned cut; a line longer than the batch extends it to that line's end
                    const void* nl = memrchr(base + pos, '\n', end - pos);
                    if (nl) end = (const uint8_t*)nl - base + 1;
                    else {
                        const void* fw = memchr(base + end, '\n', size - end);
                        end = fw ? (size_t)((const uint8_t*)fw - base) + 1 : size;
                    }
                }
                Batch b;
                b.input = input; b.ptr = base + pos; b.len = end - pos; b.mapped = true;
                // This (reader) thread faults the batch's pages in and pins them; the workers only copy, scan and post-process. With
                // every worker doing its own page faults and pinning, the address-space lock of the process was the limit
                // (four workers: 28-31 GB/s; with the reader feeding them: 42-45). MATCHY_AMD_NO_FEEDER=1 restores that.
                static const bool feeder = getenv("MATCHY_AMD_NO_FEEDER") == nullptr;
                if (feeder && b.len >= ((size_t)4 << 20)) {
                    const uintptr_t a = ((uintptr_t)b.ptr + 4095) & ~(uintptr_t)4095, z = ((uintptr_t)b.ptr + b.len) & ~(uintptr_t)4095;
#ifdef MADV_POPULATE_READ
                    (void)madvise((void*)((uintptr_t)b.ptr & ~(uintptr_t)4095), (uintptr_t)b.ptr + b.len - ((uintptr_t)b.ptr & ~(uintptr_t)4095), MADV_POPULATE_READ);
#endif
                    if (z > a && matchy_amd_host_register((const void*)a, z - a) == MATCHY_SUCCESS) b.reg = (const void*)a;
                }
                last_seq = pl.submit(std::move(b));
                pos = end;
            }
            pl.add_mapping(m, size, last_seq);
            if (input < pl.consumed.size()) pl.consumed[input] = (long long)size;
            return true;
        }
    }
    gzFile zf = nullptr;
    if (gz) {
        zf = gzdopen(fd, "rb");
        if (!zf) { fprintf(stderr, "[ERROR] Failed to process %s: cannot start gzip decoder\n", path.c_str()); close(fd); return false; }
        gzbuffer(zf, 1u << 20);
    }
    RawBuf buf(batch_bytes + 16);
    size_t have = 0, total_read = 0;
    bool ok = true;
    auto send = [&](RawBuf&& data, size_t len) {
        Batch b;
        b.input = input; b.own = std::move(data); b.ptr = b.own.data(); b.len = len;
        pl.submit(std::move(b));
    };
    for (;;) {
        if (have == buf.cap - 16) buf.grow(buf.cap * 2, have);  // a single line longer than the batch: grow
        const size_t room = buf.cap - 16 - have;
        ssize_t n;
        if (gz) {
            n = gzread(zf, buf.data() + have, (unsigned)std::min<size_t>(room, 1u << 30));
            if (n < 0) { int e; fprintf(stderr, "[ERROR] Failed to process %s: %s\n", path.c_str(), gzerror(zf, &e)); ok = false; break; }
        } else {
            n = read(fd, buf.data() + have, room);
            if (n < 0) { fprintf(stderr, "[ERROR] Failed to process %s: %s\n", path.c_str(), strerror(errno)); ok = false; break; }
        }
        have += (size_t)n;
        total_read += (size_t)n;
        if (n == 0) { send(std::move(buf), have); break; }
        if (have < batch_bytes) continue;
        // newline-aligned cut: scan up to the last '\n', carry the rest into the next batch's buffer
        const void* nl = memrchr(buf.data(), '\n', have);
        if (!nl) continue;
        const size_t cut = (const uint8_t*)nl - buf.data() + 1;
        RawBuf nxt(std::max(batch_bytes, have - cut) + 16);
        memcpy(nxt.data(), buf.data() + cut, have - cut);
        send(std::move(buf), cut);
        buf = std::move(nxt);
        have -= cut;
    }
    if (gz) gzclose(zf);   // closes fd as well
    else if (fd) close(fd);
    if (ok && !gz && fd != 0 && input < pl.consumed.size()) pl.consumed[input] = (long long)total_read;
    return ok;
}

// -f / --follow (match_processor/follow.rs:20-264): after the existing content, keep watching the files and scan what is
// appended — from the last known size to the new one, whatever it ends with, as the reference's `reader.lines()` does; a
// file that shrank is read again from the start; a file that disappeared is reported once. Records carry the wall-clock
// time of their batch as `timestamp` (the parallel path's constant "0.000" everywhere else). Ends on SIGINT / SIGTERM.
volatile sig_atomic_t g_stop = 0;
void on_stop(int) { g_stop = 1; }

void follow_inputs(MatchPipeline& pl, matchy_scanner_t* sc, const std::vector<std::string>& paths, bool stats, Totals& total) {
    struct Tail { std::string path; size_t input; off_t pos; bool gone; };
    std::vector<Tail> tails;
    for (size_t i = 0; i < paths.size(); ++i) {
        struct stat sb;
        // from where the first pass stopped reading; inputs it could not count (.gz, unreadable): from their size now
        const bool counted = i < pl.consumed.size() && pl.consumed[i] >= 0;
        tails.push_back({paths[i], i, counted ? (off_t)pl.consumed[i] : (stat(paths[i].c_str(), &sb) == 0 ? sb.st_size : 0), false});
    }
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_stop;
    sigaction(SIGINT, &sa, nullptr);
    sigaction(SIGTERM, &sa, nullptr);
    if (stats) fprintf(stderr, "[INFO] Watching for new content (Ctrl+C to stop)...\n");
    while (!g_stop) {
        bool any = false;
        for (Tail& tl : tails) {
            struct stat sb;
            if (stat(tl.path.c_str(), &sb) != 0) {
                if (!tl.gone && stats) fprintf(stderr, "[WARN] File deleted/rotated: %s\n", tl.path.c_str());
                tl.gone = true;
                continue;
            }
            tl.gone = false;
            if (sb.st_size < tl.pos) tl.pos = 0;          // truncated: start over
            if (sb.st_size == tl.pos) continue;
            const int fd = open(tl.path.c_str(), O_RDONLY);
            if (fd < 0) continue;
            Batch b;
            b.input = tl.input;
            b.own = RawBuf((size_t)(sb.st_size - tl.pos) + 16);
            size_t have = 0;
            while (have < (size_t)(sb.st_size - tl.pos)) {
                const ssize_t n = pread(fd, b.own.data() + have, (size_t)(sb.st_size - tl.pos) - have, tl.pos + (off_t)have);
                if (n <= 0) break;
                have += (size_t)n;
            }
            close(fd);
            tl.pos += (off_t)have;
            if (!have) continue;
            any = true;
            b.ptr = b.own.data(); b.len = have;
            Done d;
            pl.run_batch(sc, b, d);
            if (d.text_len) d.out.assign(d.text, d.text_len);
            if (!d.out.empty()) {
                // this batch's records carry the current time
                char ts[48];
                struct timespec now;
                clock_gettime(CLOCK_REALTIME, &now);
                snprintf(ts, sizeof(ts), "\"timestamp\":\"%.3f\"}", (double)now.tv_sec + (double)now.tv_nsec * 1e-9);
                static const std::string fixed = "\"timestamp\":\"0.000\"}";
                std::string out;
                size_t from = 0;
                for (;;) {
                    const size_t k = d.out.find(fixed + "\n", from);
                    if (k == std::string::npos) { out.append(d.out, from, std::string::npos); break; }
                    out.append(d.out, from, k - from);
                    out += ts;
                    from = k + fixed.size();
                }
                fwrite(out.data(), 1, out.size(), stdout);
                fflush(stdout);
            }
            total.lines += d.t.lines; total.lines_with_matches += d.t.lines_with_matches; total.matches += d.t.matches;
            total.candidates += d.t.candidates; total.bytes += d.t.bytes;
        }
        if (!any) usleep(100 * 1000);
    }
    if (stats) fprintf(stderr, "[INFO] Follow mode stopped\n");
}

int cmd_match(int argc, char** argv) {
    std::vector<std::string> pos;
    std::string format = "json", extractors, devices;
    bool stats = false, follow = false;
    size_t batch_bytes = (size_t)256 << 20;  // GPU batches: large, so that one launch amortises PCIe latency
    int device = 0;
    int jobs = 0;   // -j N|auto (matchy.rs:98-103: worker threads): scanners here; auto = 2 (summary) / 4 (json) per device listed once
    if (const char* d = getenv("MATCHY_AMD_DEVICE")) device = atoi(d);
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&](const char* name) -> const char* { if (i + 1 >= argc) { fprintf(stderr, "error: %s needs a value\n", name); exit(2); } return argv[++i]; };
        auto eqval = [&](const char* name, std::string& out) {  // --name=value or --name value
            size_t n = strlen(name);
            if (a.compare(0, n, name) != 0) return false;
            if (a.size() == n) { out = next(name); return true; }
            if (a[n] == '=') { out = a.substr(n + 1); return true; }
            return false;
        };
        std::string v;
        if (eqval("--format", v)) format = v;
        else if (a == "-s" || a == "--stats") stats = true;
        else if (eqval("--batch-bytes", v)) { size_t b = strtoull(v.c_str(), nullptr, 10); if (b >= 4096) batch_bytes = b; }
        else if (eqval("--extractors", v)) extractors = v;
        else if (eqval("--devices", v)) devices = v;
        else if (eqval("--device", v)) device = atoi(v.c_str());
        else if (eqval("--threads", v) || eqval("--readers", v) || eqval("--cache-size", v)) {}
        else if (a == "-j") { const std::string j = next("-j"); jobs = j == "auto" ? 0 : std::max(0, atoi(j.c_str())); }
        else if (a == "-p" || a == "--progress" || a == "--debug-routing") {}
        else if (a == "-f" || a == "--follow") follow = true;
        else if (a.size() > 1 && a[0] == '-') { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); return 2; }
        else pos.push_back(a);
    }
    if (pos.size() < 2) return usage();
    if (format != "json" && format != "summary") { fprintf(stderr, "Error: Unknown format: %s. Use 'json' or 'summary'\n", format.c_str()); return 1; }
    if (follow && stats) fprintf(stderr, "[INFO] Processing existing file content...\n");
    if (follow) for (size_t i = 1; i < pos.size(); ++i) if (pos[i] == "-") { fprintf(stderr, "Error: --follow mode not supported with stdin\n"); return 1; }
    // device list: one worker (scanner) per entry; an entry may repeat (two scanners on one GPU overlap one batch's
    // transfers with the other's kernels)
    std::vector<int> devs;
    if (devices.empty()) devs.push_back(device);
    else if (devices == "all") {
        const int n = matchy_amd_device_count();
        if (n < 1) { fprintf(stderr, "Error: no HIP device available\n"); return 1; }
        for (int d = 0; d < n; ++d) devs.push_back(d);
    } else {
        std::stringstream ss(devices);
        std::string part;
        while (std::getline(ss, part, ',')) {
            part = trim(part);
            if (part.empty() || part.find_first_not_of("0123456789") != std::string::npos) { fprintf(stderr, "Error: bad --devices entry '%s'\n", part.c_str()); return 1; }
            devs.push_back(atoi(part.c_str()));
        }
        if (devs.empty()) { fprintf(stderr, "Error: --devices needs at least one device\n"); return 1; }
    }
    {
        // Workers. A device that is listed once gets several scanners (-j N: N in all, spread over the devices): one scanner's hit
        // post-processing runs under another's transfer (one scanner per GPU: 24 GB/s in the scan phase, two: 44). A device list
        // with repeats is taken as written.
        std::vector<int> uniq = devs;
        std::sort(uniq.begin(), uniq.end());
        const bool repeats = std::adjacent_find(uniq.begin(), uniq.end()) != uniq.end();
        if (!repeats) {
            // auto: two scanners per device keep the bus busy when only counters come back (one copies while the other post-processes);
            // rendering NDJSON is per-hit host work, which four workers share better
            const size_t want = jobs > 0 ? std::max<size_t>((size_t)jobs, devs.size()) : devs.size() * (format == "json" ? 4 : 2);
            const std::vector<int> base = devs;
            for (size_t k = base.size(); k < want; ++k) devs.push_back(base[k % base.size()]);
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    // database: .mxy file, or a .csv / .json source that is built in memory first (match_cmd.rs:20-31, 222-239)
    matchy_t* db = nullptr;
    const std::string dbpath = pos[0];
    if (ends_with_ci(dbpath, ".csv") || ends_with_ci(dbpath, ".json")) {
        DatabaseBuilder b;
        std::string err;
        std::vector<uint8_t> blob;
        if (!add_inputs(b, {dbpath}, ends_with_ci(dbpath, ".csv") ? "csv" : "json", false, err) || !b.build(blob)) {
            fprintf(stderr, "Error: %s\n", err.empty() ? b.error().c_str() : err.c_str());
            return 1;
        }
        db = matchy_open_buffer(blob.data(), blob.size());
    } else {
        db = matchy_open(dbpath.c_str());
    }
    if (!db) { fprintf(stderr, "Error: Failed to open database %s: %s\n", dbpath.c_str(), matchy_amd_last_error()); return 1; }
    const bool trace = getenv("MATCHY_AMD_TRACE") != nullptr;
    auto since0 = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    if (trace) fprintf(stderr, "[matchy] database open after %.1f ms\n", since0());
    uint32_t mask = 0;
    if (!extractors.empty()) {
        uint32_t auto_mask = 0;
        if (matchy_has_ip_data(db)) auto_mask |= MATCHY_EXTRACT_IPV4 | MATCHY_EXTRACT_IPV6;
        if (matchy_has_literal_data(db) || matchy_has_glob_data(db))
            auto_mask |= MATCHY_EXTRACT_DOMAINS | MATCHY_EXTRACT_EMAILS | MATCHY_EXTRACT_HASHES | MATCHY_EXTRACT_BITCOIN | MATCHY_EXTRACT_ETHEREUM | MATCHY_EXTRACT_MONERO;
        std::string err;
        if (!parse_extractors(extractors, auto_mask, mask, err)) { fprintf(stderr, "Error: %s\n", err.c_str()); matchy_close(db); return 1; }
    }
    // The pipeline: one worker and scanner per device entry inside the library (the first scanner is created here and now, so a device
    // that cannot be used fails the command; the others when the first batch reaches their workers — a small input never pays for
    // scanners it does not use)
    std::vector<int32_t> dev32(devs.begin(), devs.end());
    matchy_multi_scanner_t* ms = matchy_multi_scanner_create(db, mask, dev32.data(), dev32.size());
    if (!ms) {
        fprintf(stderr, "Error: Failed to create the GPU scanner on device %d: %s\n", devs[0], matchy_amd_last_error());
        matchy_close(db);
        return 1;
    }
    const size_t n_scanners = devs.size();
    if (trace) fprintf(stderr, "[matchy] first of %zu scanner(s) created after %.1f ms\n", n_scanners, since0());
    MatchPipeline pl;
    pl.ms = ms;
    pl.json = format == "json";
    matchy_multi_scanner_set_batch_hook(ms, &MatchPipeline::batch_hook, &pl);
    std::vector<std::string> paths;
    bool stdin_seen = false;
    for (size_t i = 1; i < pos.size(); ++i) {
        if (pos[i] == "-") {
            if (stdin_seen) { if (stats) fprintf(stderr, "[WARN] Skipping duplicate stdin argument\n"); continue; }
            stdin_seen = true;
        }
        paths.push_back(pos[i]);
        pl.sources.push_back(pos[i] == "-" ? "stdin" : pos[i]);
    }
    Totals t;
    std::vector<char> input_failed(paths.size(), 0);
    std::vector<char> read_failed(paths.size(), 0);   // written by this thread only (input_failed belongs to the printer until it is joined)
    std::thread printer([&] { pl.printer(t, input_failed); });
    pl.consumed.assign(paths.size(), -1);
    for (size_t i = 0; i < paths.size(); ++i)
        if (!read_input(pl, i, paths[i], batch_bytes)) read_failed[i] = 1;
    pl.reader_done = true;
    printer.join();
    for (size_t i = 0; i < paths.size(); ++i) if (read_failed[i]) input_failed[i] = 1;
    for (size_t i : pl.rejected_inputs) if (i < input_failed.size()) input_failed[i] = 1;
    fflush(stdout);
    if (trace) fprintf(stderr, "[matchy] all batches done after %.1f ms\n", since0());
    pl.release_mappings((size_t)-1);
    if (follow) {
        matchy_scanner_t* fsc = matchy_scanner_create(db, mask, devs[0]);
        if (!fsc) fprintf(stderr, "Error: Failed to create the GPU scanner on device %d: %s\n", devs[0], matchy_amd_last_error());
        else { follow_inputs(pl, fsc, paths, stats, t); matchy_scanner_free(fsc); }
    }
    size_t failed = 0;
    for (char f : input_failed) failed += f != 0;
    const size_t processed = paths.size() - failed;
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) {
        fprintf(stderr, "\n[INFO] === Processing Complete ===\n");
        if (pos.size() > 2) {
            fprintf(stderr, "[INFO] Files processed: %zu\n", processed);
            if (failed) fprintf(stderr, "[INFO] Files failed: %zu\n", failed);
        }
        fprintf(stderr, "[INFO] Lines processed: %s\n", fmt_num(t.lines).c_str());
        fprintf(stderr, "[INFO] Lines with matches: %s (%.1f%%)\n", fmt_num(t.lines_with_matches).c_str(),
                t.lines ? 100.0 * (double)t.lines_with_matches / (double)t.lines : 0.0);
        fprintf(stderr, "[INFO] Total matches: %s\n", fmt_num(t.matches).c_str());
        fprintf(stderr, "[INFO] Candidates tested: %s\n", fmt_num(t.candidates).c_str());
        fprintf(stderr, "[INFO] Throughput: %.2f MB/s\n", secs > 0 ? (double)t.bytes / 1e6 / secs : 0.0);
        fprintf(stderr, "[INFO] Total time: %.2fs\n", secs);
        fprintf(stderr, "[INFO] Query rate: %.0f queries/s\n", secs > 0 ? (double)t.candidates / secs : 0.0);
        std::string dl;
        for (int d : devs) { if (!dl.empty()) dl += ","; dl += std::to_string(d); }
        fprintf(stderr, "\n[INFO] === Devices ===\n[INFO] HIP devices: %s (%zu scanner%s, batches of %zu MiB)\n", dl.c_str(), n_scanners,
                n_scanners == 1 ? "" : "s", batch_bytes >> 20);
    }
    matchy_multi_scanner_free(ms);
    matchy_close(db);
    if (trace) fprintf(stderr, "[matchy] cleaned up after %.1f ms\n", since0());
    if (failed) { fprintf(stderr, "Error: %zu file(s) failed to process\n", failed); return 1; }
    return 0;
}

// serde_json::to_string_pretty of a compact JSON text: two spaces per level, "key": value, empty containers stay "[]" / "{}"
std::string json_pretty(const std::string& c) {
    std::string o;
    int depth = 0;
    bool in_str = false;
    auto nl = [&]() { o.push_back('\n'); o.append((size_t)depth * 2, ' '); };
    for (size_t i = 0; i < c.size(); ++i) {
        const char ch = c[i];
        if (in_str) {
            o.push_back(ch);
            if (ch == '\\' && i + 1 < c.size()) o.push_back(c[++i]);
            else if (ch == '"') in_str = false;
            continue;
        }
        switch (ch) {
            case '"': in_str = true; o.push_back(ch); break;
            case '[': case '{':
                o.push_back(ch);
                if (i + 1 < c.size() && (c[i + 1] == ']' || c[i + 1] == '}')) { o.push_back(c[++i]); break; }
                ++depth; nl();
                break;
            case ']': case '}': --depth; nl(); o.push_back(ch); break;
            case ',': o.push_back(ch); nl(); break;
            case ':': o += ": "; break;
            default: o.push_back(ch);
        }
    }
    return o;
}

// matchy query (bin/commands/query_cmd.rs:8-69): pretty JSON array on stdout, exit status 0 = found, 1 = not found
int cmd_query(int argc, char** argv) {
    std::vector<std::string> pos;
    bool quiet = false;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-q" || a == "--quiet") quiet = true;
        else if (a == "--") { for (++i; i < argc; ++i) pos.push_back(argv[i]); }
        else if (a.size() > 1 && a[0] == '-' && pos.size() < 1) { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); return 2; }
        else pos.push_back(a);
    }
    if (pos.size() != 2) return usage();
    matchy_t* db = matchy_open(pos[0].c_str());
    if (!db) { fprintf(stderr, "Error: Failed to load database: %s\n", pos[0].c_str()); return 1; }
    int32_t found = 0;
    char* js = matchy_amd_query_json(db, pos[1].c_str(), &found);
    if (!js) { fprintf(stderr, "Error: Query failed for: %s: %s\n", pos[1].c_str(), matchy_amd_last_error()); matchy_close(db); return 1; }
    if (!quiet) printf("%s\n", json_pretty(js).c_str());
    matchy_free_string(js);
    matchy_close(db);
    return found ? 0 : 1;
}

// matchy extract (bin/commands/extract_cmd.rs:40-289). The reference extracts line by line (extract_from_line,
// matchy-extractor/src/lib.rs:1471-1521: domains, IPv4, e-mails, IPv6, hashes, Bitcoin, Ethereum, Monero per line); its line
// functions are the chunk functions applied to one line, no token crosses a newline, so every batch goes through
// the GPU extractor once and the items are put back into line order here. LineScanner (bin/cli_utils.rs:9-110) trims ASCII
// whitespace from both ends of a line and skips empty lines: form feeds in those margins — whitespace for the trim, not
// a word boundary for the extractor — are turned into spaces first.
struct ExtractStats { unsigned long long lines = 0, found = 0, v4 = 0, v6 = 0, dom = 0, mail = 0, bytes = 0; };
inline bool is_ws(uint8_t c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0C; }
inline int line_rank(uint8_t t) {
    switch (t) {
        case MATCHY_ITEM_TYPE_DOMAIN: return 0;
        case MATCHY_ITEM_TYPE_IPV4: return 1;
        case MATCHY_ITEM_TYPE_EMAIL: return 2;
        case MATCHY_ITEM_TYPE_IPV6: return 3;
        case MATCHY_ITEM_TYPE_BITCOIN: return 5;
        case MATCHY_ITEM_TYPE_ETHEREUM: return 6;
        case MATCHY_ITEM_TYPE_MONERO: return 7;
        default: return 4;   // the hash types
    }
}
bool extract_batch(matchy_extractor_t* ex, uint8_t* data, size_t len, int fmt, bool show_cand, std::vector<std::string>* seen_sorted,
                   std::vector<std::string>& seen_new, ExtractStats& st) {
    if (!len) return true;
    // line table: [start, end) of every trimmed, non-empty line
    std::vector<std::pair<size_t, size_t>> lines;
    for (size_t p = 0; p < len;) {
        const uint8_t* nlp = (const uint8_t*)memchr(data + p, '\n', len - p);
        size_t e = nlp ? (size_t)(nlp - data) : len, a = p, b = e;
        while (a < b && is_ws(data[a])) { if (data[a] == 0x0C) data[a] = ' '; ++a; }
        while (b > a && is_ws(data[b - 1])) { if (data[b - 1] == 0x0C) data[b - 1] = ' '; --b; }
        if (b > a) { lines.push_back({a, b}); st.lines++; st.bytes += b - a; }
        p = e + 1;
    }
    matchy_matches_t m;
    memset(&m, 0, sizeof(m));
    if (matchy_extractor_extract_chunk(ex, data, len, &m) != MATCHY_SUCCESS) {
        fprintf(stderr, "[ERROR] extraction failed: %s\n", matchy_amd_last_error());
        return false;
    }
    struct It { size_t line; int rank; size_t idx; };
    std::vector<It> its(m.count);
    for (size_t i = 0; i < m.count; ++i) {
        const size_t s = m.items[i].start;
        // the trimmed line that holds position s
        size_t lo = 0, hi = lines.size();
        while (lo < hi) { size_t mid = (lo + hi) / 2; if (lines[mid].second <= s) lo = mid + 1; else hi = mid; }
        its[i] = It{lo, line_rank(m.items[i].item_type), i};
    }
    std::sort(its.begin(), its.end(), [&](const It& a, const It& b) {
        if (a.line != b.line) return a.line < b.line;
        if (a.rank != b.rank) return a.rank < b.rank;
        return m.items[a.idx].start < m.items[b.idx].start;
    });
    std::string out;
    for (const It& it : its) {
        const matchy_match_t& mt = m.items[it.idx];
        const std::string text((const char*)data + mt.start, mt.end - mt.start);
        std::string tname = matchy_item_type_name(mt.item_type);
        if (show_cand) {
            const size_t ls = it.line < lines.size() ? lines[it.line].first : 0;
            fprintf(stderr, "[CANDIDATE] %s at %zu-%zu: %s\n", tname.c_str(), (size_t)mt.start - ls, (size_t)mt.end - ls, text.c_str());
        }
        if (seen_sorted) {
            if (std::binary_search(seen_sorted->begin(), seen_sorted->end(), text)) continue;
            if (std::find(seen_new.begin(), seen_new.end(), text) != seen_new.end()) continue;
            seen_new.push_back(text);
            if (seen_new.size() >= 4096) {   // fold the recent ones into the sorted set
                seen_sorted->insert(seen_sorted->end(), seen_new.begin(), seen_new.end());
                std::sort(seen_sorted->begin(), seen_sorted->end());
                seen_new.clear();
            }
        }
        for (char& ch : tname) ch = (char)tolower((unsigned char)ch);
        out.clear();
        if (fmt == 0) {
            out += "{\"type\":\"" + tname + "\",\"value\":\"";
            for (char ch : text) { if (ch == '\\') out += "\\\\"; else if (ch == '"') out += "\\\""; else out.push_back(ch); }
            out += "\"}\n";
        } else if (fmt == 1) {
            out += tname + ",\"";
            for (char ch : text) { if (ch == '"') out += "\"\""; else out.push_back(ch); }
            out += "\"\n";
        } else {
            out += text; out.push_back('\n');
        }
        fwrite(out.data(), 1, out.size(), stdout);
        st.found++;
        if (mt.item_type == MATCHY_ITEM_TYPE_IPV4) st.v4++;
        else if (mt.item_type == MATCHY_ITEM_TYPE_IPV6) st.v6++;
        else if (mt.item_type == MATCHY_ITEM_TYPE_DOMAIN) st.dom++;
        else if (mt.item_type == MATCHY_ITEM_TYPE_EMAIL) st.mail++;
    }
    matchy_matches_free(&m);
    return true;
}

int cmd_extract(int argc, char** argv) {
    std::vector<std::string> inputs;
    std::string format = "json", types;
    bool have_types = false, unique = false, stats = false, show_cand = false;
    uint32_t min_labels = 2;
    size_t batch_bytes = (size_t)256 << 20;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&](const char* name) -> const char* { if (i + 1 >= argc) { fprintf(stderr, "error: %s needs a value\n", name); exit(2); } return argv[++i]; };
        auto eqval = [&](const char* name, std::string& out) {
            size_t n = strlen(name);
            if (a.compare(0, n, name) != 0) return false;
            if (a.size() == n) { out = next(name); return true; }
            if (a[n] == '=') { out = a.substr(n + 1); return true; }
            return false;
        };
        std::string v;
        if (eqval("--format", v)) format = v;
        else if (eqval("--types", v)) { types = v; have_types = true; }
        else if (eqval("--min-labels", v)) min_labels = (uint32_t)strtoul(v.c_str(), nullptr, 10);
        else if (eqval("--batch-bytes", v)) { size_t b = strtoull(v.c_str(), nullptr, 10); if (b >= 4096) batch_bytes = b; }
        else if (a == "--no-boundaries") { fprintf(stderr, "Error: --no-boundaries is not supported by this build\n"); return 1; }
        else if (a == "-u" || a == "--unique") unique = true;
        else if (a == "-s" || a == "--stats") stats = true;
        else if (a == "--show-candidates") show_cand = true;
        else if (a.size() > 1 && a[0] == '-') { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); return 2; }
        else inputs.push_back(a);
    }
    if (inputs.empty()) return usage();
    for (char& ch : format) ch = (char)tolower((unsigned char)ch);
    const int fmt = format == "json" ? 0 : format == "csv" ? 1 : format == "text" ? 2 : -1;
    if (fmt < 0) { fprintf(stderr, "Error: Invalid format '%s', expected: json, csv, or text\n", format.c_str()); return 1; }
    bool v4 = true, v6 = true, dom = true, mail = true;
    if (have_types) {
        v4 = v6 = dom = mail = false;
        std::string low = types;
        for (char& ch : low) ch = (char)tolower((unsigned char)ch);
        std::stringstream ss(low);
        std::string part;
        while (std::getline(ss, part, ',')) {
            part = trim(part);
            if (part == "ipv4" || part == "ip4") v4 = true;
            else if (part == "ipv6" || part == "ip6") v6 = true;
            else if (part == "domain" || part == "domains") dom = true;
            else if (part == "email" || part == "emails") mail = true;
            else if (part == "ip") v4 = v6 = true;
            else if (part == "all") v4 = v6 = dom = mail = true;
            else { fprintf(stderr, "Error: Unknown extraction type '%s', expected: ipv4, ipv6, ip, domain, email, all\n", part.c_str()); return 1; }
        }
        if (!v4 && !v6 && !dom && !mail) { fprintf(stderr, "Error: At least one extraction type must be enabled\n"); return 1; }
    }
    // the command only switches these four; hashes and the crypto-address extractors keep the builder's default (on)
    uint32_t flags = MATCHY_EXTRACT_HASHES | MATCHY_EXTRACT_BITCOIN | MATCHY_EXTRACT_ETHEREUM | MATCHY_EXTRACT_MONERO;
    if (v4) flags |= MATCHY_EXTRACT_IPV4;
    if (v6) flags |= MATCHY_EXTRACT_IPV6;
    if (dom) flags |= MATCHY_EXTRACT_DOMAINS;
    if (mail) flags |= MATCHY_EXTRACT_EMAILS;
    matchy_extractor_t* ex = matchy_amd_extractor_create(flags, min_labels);
    if (!ex) { fprintf(stderr, "Error: Failed to create pattern extractor: %s\n", matchy_amd_last_error()); return 1; }
    if (stats) {
        std::string en;
        for (auto& pr : {std::make_pair(v4, "IPv4"), std::make_pair(v6, "IPv6"), std::make_pair(dom, "domains"), std::make_pair(mail, "emails")})
            if (pr.first) { if (!en.empty()) en += ", "; en += pr.second; }
        fprintf(stderr, "[INFO] Extracting: %s\n", en.c_str());
        if (dom) fprintf(stderr, "[INFO] Min domain labels: %u\n", min_labels);
        fprintf(stderr, "[INFO] Word boundaries: true\n[INFO] Unique mode: %s\n", unique ? "true" : "false");
    }
    const auto t0 = std::chrono::steady_clock::now();
    ExtractStats st;
    std::vector<std::string> seen_sorted, seen_new;
    if (fmt == 1) fputs("type,value\n", stdout);
    bool ok = true;
    for (const std::string& path : inputs) {
        int fd = path == "-" ? 0 : open(path.c_str(), O_RDONLY);
        if (fd < 0) { fprintf(stderr, "Error: Failed to open file: %s\n", path.c_str()); ok = false; break; }
        std::vector<uint8_t> buf(batch_bytes + 16);
        size_t have = 0;
        for (;;) {
            if (have == buf.size() - 16) buf.resize(buf.size() * 2);
            const ssize_t n = read(fd, buf.data() + have, buf.size() - 16 - have);
            if (n < 0) { fprintf(stderr, "Error: read failed on %s: %s\n", path.c_str(), strerror(errno)); ok = false; break; }
            have += (size_t)n;
            if (n == 0) { ok = extract_batch(ex, buf.data(), have, fmt, show_cand, unique ? &seen_sorted : nullptr, seen_new, st) && ok; break; }
            if (have < batch_bytes) continue;
            const void* nlp = memrchr(buf.data(), '\n', have);
            if (!nlp) continue;
            const size_t cut = (const uint8_t*)nlp - buf.data() + 1;
            if (!extract_batch(ex, buf.data(), cut, fmt, show_cand, unique ? &seen_sorted : nullptr, seen_new, st)) { ok = false; break; }
            memmove(buf.data(), buf.data() + cut, have - cut);
            have -= cut;
        }
        if (fd) close(fd);
        if (!ok) break;
    }
    fflush(stdout);
    matchy_extractor_free(ex);
    if (!ok) return 1;
    if (stats) {
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "\n[INFO] === Extraction Complete ===\n[INFO] Lines processed: %s\n[INFO] Patterns found: %s\n", fmt_num(st.lines).c_str(), fmt_num(st.found).c_str());
        if (v4 && st.v4) fprintf(stderr, "[INFO]   IPv4: %s\n", fmt_num(st.v4).c_str());
        if (v6 && st.v6) fprintf(stderr, "[INFO]   IPv6: %s\n", fmt_num(st.v6).c_str());
        if (dom && st.dom) fprintf(stderr, "[INFO]   Domains: %s\n", fmt_num(st.dom).c_str());
        if (mail && st.mail) fprintf(stderr, "[INFO]   Emails: %s\n", fmt_num(st.mail).c_str());
        fprintf(stderr, "[INFO] Throughput: %.2f MB/s\n[INFO] Total time: %.2fs\n", secs > 0 ? (double)st.bytes / 1e6 / secs : 0.0, secs);
    }
    return 0;
}

// matchy inspect (bin/commands/inspect_cmd.rs:10-119): what the file holds, from its metadata; no GPU involved
bool meta_uint_of(const DataValue& m, const char* key, unsigned long long& out) {
    if (m.type != DataValue::MAP) return false;
    auto it = m.map.find(key);
    if (it == m.map.end()) return false;
    const DataValue& v = it->second;
    if (v.type == DataValue::UINT16 || v.type == DataValue::UINT32 || v.type == DataValue::UINT64) { out = v.u; return true; }
    return false;
}
std::string fmt_unix_time(unsigned long long ts) {   // bin/cli_utils.rs:255-283
    time_t t = (time_t)ts;
    struct tm g;
    gmtime_r(&t, &g);
    char b[64];
    snprintf(b, sizeof(b), "%04d-%02d-%02d %02d:%02d:%02d UTC", g.tm_year + 1900, g.tm_mon + 1, g.tm_mday, g.tm_hour, g.tm_min, g.tm_sec);
    return b;
}
std::string fmt_data_value(const DataValue& v, const std::string& indent) {   // bin/cli_utils.rs:322-364
    std::string o;
    switch (v.type) {
        case DataValue::STRING: return "\"" + v.str + "\"";
        case DataValue::MAP:
            if (v.map.empty()) return "{}";
            o = "{\n";
            for (auto& kv : v.map) o += indent + "  " + kv.first + ": " + fmt_data_value(kv.second, indent + "  ") + ",\n";
            return o + indent + "}";
        case DataValue::ARRAY:
            if (v.arr.empty()) return "[]";
            o = "[";
            for (size_t i = 0; i < v.arr.size(); ++i) { if (i) o += ", "; o += fmt_data_value(v.arr[i], indent); }
            return o + "]";
        default: { std::string j; to_json(v, j); return j; }
    }
}
int cmd_inspect(int argc, char** argv) {
    std::vector<std::string> pos;
    bool json = false, verbose = false;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-j" || a == "--json") json = true;
        else if (a == "-v" || a == "--verbose") verbose = true;
        else if (a.size() > 1 && a[0] == '-') { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); return 2; }
        else pos.push_back(a);
    }
    if (pos.size() != 1) return usage();
    std::vector<uint8_t> bytes;
    {
        std::ifstream f(pos[0], std::ios::binary);
        if (!f) { fprintf(stderr, "Error: Failed to load database: %s\n", pos[0].c_str()); return 1; }
        bytes.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    }
    DbImage img;
    std::string err;
    if (!img.open(std::move(bytes), err)) { fprintf(stderr, "Error: Failed to load database: %s: %s\n", pos[0].c_str(), err.c_str()); return 1; }
    // counts as Database::ip_count / literal_count / glob_count report them (database.rs:1147-1210): metadata first
    unsigned long long ipc = 0, litc = 0, globc = 0;
    if (!meta_uint_of(img.metadata, "ip_entry_count", ipc)) (void)meta_uint_of(img.metadata, "node_count", ipc);
    (void)meta_uint_of(img.metadata, "literal_entry_count", litc);
    if (!meta_uint_of(img.metadata, "glob_entry_count", globc)) globc = img.pattern_count;
    const bool has_string = img.has_literal || img.has_glob;
    if (json) {
        std::string o = "{\"file\":";
        auto esc = [&](const std::string& x) { std::string q; to_json(DataValue::String(x), q); return q; };
        o += esc(pos[0]) + ",\"format\":" + esc(img.format_name());
        o += std::string(",\"glob_count\":") + std::to_string(globc) + ",\"has_glob_data\":" + (img.has_glob ? "true" : "false");
        o += std::string(",\"has_ip_data\":") + (img.has_ip ? "true" : "false") + ",\"has_literal_data\":" + (img.has_literal ? "true" : "false");
        o += std::string(",\"has_string_data\":") + (has_string ? "true" : "false") + ",\"ip_count\":" + std::to_string(ipc) + ",\"literal_count\":" + std::to_string(litc);
        if (img.metadata.type == DataValue::MAP) { o += ",\"metadata\":"; to_json(img.metadata, o); }
        o += "}";
        printf("%s\n", json_pretty(o).c_str());
        return 0;
    }
    printf("Database: %s\n", pos[0].c_str());
    printf("Format:   %s\n\n", ipc > 0 && (litc > 0 || globc > 0) ? "Combined IP+String database" : ipc > 0 ? "IP database" : (litc > 0 || globc > 0) ? "String database" : "Empty database");
    printf("Capabilities:\n");
    if (ipc > 0) printf("  IP lookups:      \xE2\x9C\x93\n    Entries:       %llu\n", ipc);
    else printf("  IP lookups:      \xE2\x9C\x97\n");
    printf("  String lookups:  %s\n", has_string ? "\xE2\x9C\x93" : "\xE2\x9C\x97");
    if (img.has_literal) printf("    Literals:      \xE2\x9C\x93 (%llu strings)\n", litc);
    if (img.has_glob) printf("    Globs:         \xE2\x9C\x93 (%llu patterns)\n", globc);
    if (img.metadata.type == DataValue::MAP) {
        printf("\nMetadata:\n");
        auto it = img.metadata.map.find("database_type");
        if (it != img.metadata.map.end() && it->second.type == DataValue::STRING) printf("  Database type:   %s\n", it->second.str.c_str());
        it = img.metadata.map.find("description");
        if (it != img.metadata.map.end() && it->second.type == DataValue::MAP) {
            printf("  Description:\n");
            for (auto& kv : it->second.map) if (kv.second.type == DataValue::STRING) printf("    %s: %s\n", kv.first.c_str(), kv.second.str.c_str());
        }
        unsigned long long v;
        if (meta_uint_of(img.metadata, "build_epoch", v)) printf("  Build time:      %s (%llu)\n", fmt_unix_time(v).c_str(), v);
        if (meta_uint_of(img.metadata, "ip_version", v)) printf("  IP version:      IPv%llu\n", v);
        if (verbose) printf("\nFull metadata:\n%s\n", fmt_data_value(img.metadata, "  ").c_str());
    }
    return 0;
}

// matchy validate (bin/commands/validate_cmd.rs): the structural checks of matchy_validate = DbImage::check_structure, the same
// checks every matchy_open runs before a file is uploaded (headers, metadata, section bounds, every pointer the kernels follow:
// IP data records, literal-hash strings, wildcard / pattern / glob-segment arrays, the reachable Aho-Corasick nodes). Both
// levels run the same checks here; exit status 0 = valid. The reference's detailed statistics block is not reproduced.
int cmd_validate(int argc, char** argv) {
    std::vector<std::string> pos;
    std::string level = "strict";
    bool json = false;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-j" || a == "--json") json = true;
        else if (a == "-v" || a == "--verbose") {}
        else if (a == "-l" || a == "--level") { if (i + 1 >= argc) return usage(); level = argv[++i]; }
        else if (a.compare(0, 8, "--level=") == 0) level = a.substr(8);
        else if (a.size() > 1 && a[0] == '-') { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); return 2; }
        else pos.push_back(a);
    }
    if (pos.size() != 1) return usage();
    for (char& ch : level) ch = (char)tolower((unsigned char)ch);
    if (level != "standard" && level != "strict") { fprintf(stderr, "Error: Invalid validation level: '%s'. Must be: standard or strict\n", level.c_str()); return 1; }
    char* msg = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const int32_t rc = matchy_validate(pos[0].c_str(), level == "strict" ? MATCHY_VALIDATION_STRICT : MATCHY_VALIDATION_STANDARD, &msg);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const bool valid = rc == MATCHY_SUCCESS;
    if (json) {
        std::string o = "{\"database\":";
        std::string q;
        to_json(DataValue::String(pos[0]), q);
        o += q + ",\"duration_ms\":" + std::to_string((long long)ms) + ",\"errors\":[";
        if (!valid) { q.clear(); to_json(DataValue::String(msg ? msg : "validation failed"), q); o += q; }
        o += std::string("],\"is_valid\":") + (valid ? "true" : "false") + ",\"validation_level\":\"" + level + "\"}";
        printf("%s\n", json_pretty(o).c_str());
    } else {
        printf("Validating: %s\nLevel:      %s\n\n", pos[0].c_str(), level.c_str());
        if (!valid) printf("\xE2\x9D\x8C ERRORS (1):\n  \xE2\x80\xA2 %s\n\n", msg ? msg : "validation failed");
        printf("%s\n", valid ? "\xE2\x9C\x85 VALIDATION PASSED (structural checks: every offset the readers follow lies inside its section)" : "\xE2\x9D\x8C VALIDATION FAILED");
    }
    if (msg) matchy_free_string(msg);
    return valid ? 0 : 1;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) return usage();
    std::string cmd = argv[1];
    if (cmd == "build") return cmd_build(argc - 2, argv + 2);
    if (cmd == "match") return cmd_match(argc - 2, argv + 2);
    if (cmd == "query") return cmd_query(argc - 2, argv + 2);
    if (cmd == "extract") return cmd_extract(argc - 2, argv + 2);
    if (cmd == "inspect") return cmd_inspect(argc - 2, argv + 2);
    if (cmd == "validate") return cmd_validate(argc - 2, argv + 2);
    if (cmd == "--version" || cmd == "-V" || cmd == "version") { printf("matchy %s (MI355X build)\n", matchy_version()); return 0; }
    return usage();
}
