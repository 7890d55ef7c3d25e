/*
 * matchy_amd.h — C ABI of the MI355X-native `matchy match` engine (libmatchy_amd.so).
 *
 * Drop-in boundary: every `matchy_*` function declared in PART 1 keeps the name, argument meaning, ownership and
 * error convention of the reference's C API (reference = matchylabs/matchy @ 2025-12-12,
 * crates/matchy/include/matchy/matchy.h, implemented in crates/matchy/src/c_api/matchy.rs). The citation after
 * each declaration names the reference interface it replaces. PART 2 is additive (`matchy_scanner_*`): the bulk
 * entry point behind which the HIP pipeline sits; it is called exactly where the reference calls
 * Worker::process_batch (crates/matchy/src/processing/parallel.rs:411-439 -> processing/mod.rs:353-448).
 *
 * All lookups and extractions run on the GPU. There is no CPU fallback: without a usable HIP device
 * matchy_open()/matchy_extractor_create() return NULL and scanner calls return MATCHY_ERROR_IO.
 */
#ifndef MATCHY_AMD_H
#define MATCHY_AMD_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (matchy.h:46-86) */
#define MATCHY_SUCCESS 0
#define MATCHY_ERROR_FILE_NOT_FOUND -1
#define MATCHY_ERROR_INVALID_FORMAT -2
#define MATCHY_ERROR_CORRUPT_DATA -3
#define MATCHY_ERROR_OUT_OF_MEMORY -4
#define MATCHY_ERROR_INVALID_PARAM -5
#define MATCHY_ERROR_IO -6

/* ---- extractor flags and item types (matchy.h:188-288) */
#define MATCHY_EXTRACT_DOMAINS (1 << 0)
#define MATCHY_EXTRACT_EMAILS (1 << 1)
#define MATCHY_EXTRACT_IPV4 (1 << 2)
#define MATCHY_EXTRACT_IPV6 (1 << 3)
#define MATCHY_EXTRACT_HASHES (1 << 4)
#define MATCHY_EXTRACT_BITCOIN (1 << 5)
#define MATCHY_EXTRACT_ETHEREUM (1 << 6)
#define MATCHY_EXTRACT_MONERO (1 << 7)
#define MATCHY_EXTRACT_ALL 255

#define MATCHY_ITEM_TYPE_DOMAIN 0
#define MATCHY_ITEM_TYPE_EMAIL 1
#define MATCHY_ITEM_TYPE_IPV4 2
#define MATCHY_ITEM_TYPE_IPV6 3
#define MATCHY_ITEM_TYPE_MD5 4
#define MATCHY_ITEM_TYPE_SHA1 5
#define MATCHY_ITEM_TYPE_SHA256 6
#define MATCHY_ITEM_TYPE_SHA384 7
#define MATCHY_ITEM_TYPE_SHA512 8
#define MATCHY_ITEM_TYPE_BITCOIN 9
#define MATCHY_ITEM_TYPE_ETHEREUM 10
#define MATCHY_ITEM_TYPE_MONERO 11

/* ================================================================================================
 * PART 1 — reference-compatible surface
 * ================================================================================================ */

typedef struct matchy_builder_t matchy_builder_t; /* opaque (matchy.h:293-295) */
typedef struct matchy_t matchy_t;                 /* opaque (matchy.h:396-398) */
typedef struct matchy_extractor_t matchy_extractor_t; /* opaque (matchy.h:513-515) */

/* matchy.h:361-391. auto_reload / reload_callback are accepted and ignored (watching is out of scope);
 * cache_capacity: entries of the query cache of matchy_query (LRU keyed by the query string, "not found" included; 0 disables it:
 * c_api/matchy.rs:805-808). The reference keeps one such cache per thread, this build one per handle. */
typedef struct matchy_open_options_t {
  uint32_t cache_capacity;
  bool auto_reload;
  void (*reload_callback)(const void *event, void *user_data);
  void *reload_callback_user_data;
} matchy_open_options_t;

/* matchy.h:437-454 */
typedef struct matchy_result_t {
  bool found;
  uint8_t prefix_len;
  void *_data_cache;
  const matchy_t *_db_ref;
} matchy_result_t;

/* matchy.h:520-556 */
typedef struct matchy_match_t {
  uint8_t item_type;
  const char *value;
  uintptr_t start;
  uintptr_t end;
} matchy_match_t;
typedef struct matchy_matches_t {
  const matchy_match_t *items;
  uintptr_t count;
  void *_internal;
} matchy_matches_t;

/* ---- builder (replaces c_api/matchy.rs:300-600; matchy.h:578-753) */
matchy_builder_t *matchy_builder_new(void);                                            /* matchy.h:578 */
int32_t matchy_builder_set_case_insensitive(matchy_builder_t *b, bool case_insensitive); /* matchy.h:605 */
int32_t matchy_builder_add(matchy_builder_t *b, const char *key, const char *json_data); /* matchy.h:670, c_api/matchy.rs:401-455 */
int32_t matchy_builder_set_description(matchy_builder_t *b, const char *description);   /* matchy.h:687 */
int32_t matchy_builder_save(matchy_builder_t *b, const char *filename);                 /* matchy.h:711 */
int32_t matchy_builder_build(matchy_builder_t *b, uint8_t **buffer, uintptr_t *size);   /* matchy.h:740 (buffer is malloc'ed; caller frees with free()) */
void matchy_builder_free(matchy_builder_t *b);                                          /* matchy.h:753 */

/* ---- open / close (c_api/matchy.rs:781-939, 1055-1059; matchy.h:777-938) */
void matchy_init_open_options(matchy_open_options_t *options);                                   /* matchy.h:777 */
matchy_t *matchy_open_with_options(const char *filename, const matchy_open_options_t *options);  /* matchy.h:815 (NULL options -> NULL) */
matchy_t *matchy_open(const char *filename);                                                     /* matchy.h:844 */
matchy_t *matchy_open_buffer(const uint8_t *buffer, uintptr_t size);                             /* matchy.h:863 (copies the buffer) */
void matchy_close(matchy_t *db);                                                                 /* matchy.h:938 (NULL-safe) */

/* ---- query (c_api/matchy.rs:1100-1239; matchy.h:980-1034). A single query is answered on the HOST (csrc/host_lookup.cpp: trie walk,
 * literal probe, Paraglob::find_all over the mapped sections; SURVEY §8b "single queries stay on the CPU path"): a kernel launch would cost
 * tens of microseconds where the reference answers in 0.2. Bulk work — matchy_scanner_*, matchy_extractor_extract_chunk — runs the HIP kernels. */
matchy_result_t matchy_query(const matchy_t *db, const char *query);                    /* matchy.h:980 */
void matchy_query_into(const matchy_t *db, const char *query, matchy_result_t *result); /* matchy.h:1008 */
void matchy_free_result(matchy_result_t *result);                                       /* matchy.h:1022 */
void matchy_free_string(char *string);                                                  /* matchy.h:1034 */
char *matchy_result_to_json(const matchy_result_t *result);                             /* matchy.h:1323 */

/* ---- introspection (matchy.h:1043-1190) */
const char *matchy_version(void);                                  /* matchy.h:1043 */
const char *matchy_format(const matchy_t *db);                     /* matchy.h:1059 */
bool matchy_has_ip_data(const matchy_t *db);                       /* matchy.h:1074 */
bool matchy_has_string_data(const matchy_t *db);                   /* matchy.h:1089 */
bool matchy_has_literal_data(const matchy_t *db);                  /* matchy.h:1104 */
bool matchy_has_glob_data(const matchy_t *db);                     /* matchy.h:1119 */
char *matchy_metadata(const matchy_t *db);                         /* matchy.h:1154 (JSON; free with matchy_free_string) */
char *matchy_get_pattern_string(const matchy_t *db, uint32_t id);  /* matchy.h:1173 */
uintptr_t matchy_pattern_count(const matchy_t *db);                /* matchy.h:1190 */

/* ---- extractor (c_api/matchy.rs:2270-2395; matchy.h:1380-1460). Items come back in the reference's chunk-path
 * order (IPv6, IPv4, e-mail, domain, hashes, Bitcoin, Ethereum, Monero; by position inside each class). */
matchy_extractor_t *matchy_extractor_create(uint32_t flags);                                  /* matchy.h:1380 */
int32_t matchy_extractor_extract_chunk(const matchy_extractor_t *extractor, const uint8_t *data, uintptr_t len,
                                       matchy_matches_t *matches);                             /* matchy.h:1423 */
void matchy_matches_free(matchy_matches_t *matches);                                          /* matchy.h:1434 */
void matchy_extractor_free(matchy_extractor_t *extractor);                                    /* matchy.h:1445 */
const char *matchy_item_type_name(uint8_t item_type);                                         /* matchy.h:1460 */

/* ---- statistics (matchy.h:403-432, 894, 917). cache_hits / cache_misses count the query cache of matchy_query (bulk scans do not
 * use it: their lookups run on the device). Query-type accounting follows Database::lookup (database.rs:725-804), including its
 * quirk that a miss is always counted as a string query. */
typedef struct matchy_stats_t {
  uint64_t total_queries, queries_with_match, queries_without_match, cache_hits, cache_misses, ip_queries, string_queries;
} matchy_stats_t;
void matchy_get_stats(const matchy_t *db, matchy_stats_t *stats);      /* matchy.h:894 */
void matchy_clear_cache(const matchy_t *db);                            /* matchy.h:917 */
bool matchy_has_pattern_data(const matchy_t *db);                       /* matchy.h:1137 (deprecated alias of has_string_data) */

/* ---- structured access to the data of a query result (MaxMind-style walkers, matchy.h:452-508, 1220-1290) */
#define MATCHY_ERROR_LOOKUP_PATH_INVALID (-7) /* matchy.h:163 */
#define MATCHY_ERROR_NO_DATA (-8)             /* matchy.h:168 */
#define MATCHY_ERROR_DATA_PARSE (-9)          /* matchy.h:173 */
#define MATCHY_DATA_TYPE_POINTER 1            /* matchy.h:92-157 */
#define MATCHY_DATA_TYPE_UTF8_STRING 2
#define MATCHY_DATA_TYPE_DOUBLE 3
#define MATCHY_DATA_TYPE_BYTES 4
#define MATCHY_DATA_TYPE_UINT16 5
#define MATCHY_DATA_TYPE_UINT32 6
#define MATCHY_DATA_TYPE_MAP 7
#define MATCHY_DATA_TYPE_INT32 8
#define MATCHY_DATA_TYPE_UINT64 9
#define MATCHY_DATA_TYPE_UINT128 10
#define MATCHY_DATA_TYPE_ARRAY 11
#define MATCHY_DATA_TYPE_BOOLEAN 14
#define MATCHY_DATA_TYPE_FLOAT 15
typedef union matchy_entry_data_value_u { /* matchy.h:24-36 */
  uint32_t pointer;
  const char *utf8_string;
  double double_value;
  const uint8_t *bytes;
  uint16_t uint16;
  uint32_t uint32;
  int32_t int32;
  uint64_t uint64;
  uint8_t uint128[16];
  bool boolean;
  float float_value;
} matchy_entry_data_value_u;
typedef struct matchy_entry_s { /* matchy.h:457-466 */
  const matchy_t *db;
  const void *data_ptr;
} matchy_entry_s;
typedef struct matchy_entry_data_t { /* matchy.h:471-494 */
  bool has_data;
  uint32_t type_;
  matchy_entry_data_value_u value;
  uint32_t data_size; /* string / bytes: length; map / array: element count; scalars: width */
  uint32_t offset;
} matchy_entry_data_t;
typedef struct matchy_entry_data_list_t { /* matchy.h:499-508 */
  matchy_entry_data_t entry_data;
  struct matchy_entry_data_list_t *next;
} matchy_entry_data_list_t;
int32_t matchy_result_get_entry(const matchy_result_t *result, matchy_entry_s *entry);                               /* matchy.h:1220 */
/* path = NULL-terminated array of map keys / decimal array indexes. String and byte pointers stay valid until
 * matchy_free_result (the reference leaks them instead). */
int32_t matchy_aget_value(const matchy_entry_s *entry, matchy_entry_data_t *entry_data, const char *const *path);   /* matchy.h:1245 */
/* Depth-first list: a node, then its children (map values in key order; the reference's HashMap order is unspecified). */
int32_t matchy_get_entry_data_list(const matchy_entry_s *entry, matchy_entry_data_list_t **entry_data_list);         /* matchy.h:1277 */
void matchy_free_entry_data_list(matchy_entry_data_list_t *list);                                                     /* matchy.h:1290 */

/* ---- validation (matchy.h:1357): both levels run the bounds-checked parse every open performs (header, tree size,
 * section offsets, literal / paraglob tables). *error_message (if any) is freed with matchy_free_string. */
#define MATCHY_VALIDATION_STANDARD 0 /* matchy.h:178 */
#define MATCHY_VALIDATION_STRICT 1   /* matchy.h:183 */
int32_t matchy_validate(const char *filename, int32_t level, char **error_message);
/* Schema validation of entry data is outside this build: every schema name is reported as unknown (matchy.h:86, 625). */
#define MATCHY_ERROR_UNKNOWN_SCHEMA (-8)
int32_t matchy_builder_set_schema(matchy_builder_t *builder, const char *schema_name);

/* ================================================================================================
 * PART 2 — additive bulk-scan surface (not in the reference)
 * ================================================================================================ */

typedef struct matchy_scanner_t matchy_scanner_t;

/* One match of Worker::process_bytes (processing/mod.rs:423-443): candidate span + lookup result, 16 bytes (the
 * GPU writes these records straight into pinned host memory; compact records halve the PCIe time of a scan). */
typedef struct matchy_scan_hit_t {
  uint32_t start;        /* byte offset of the matched text in the scanned buffer */
  uint32_t len_type;     /* length of the matched text in bits 0..23, MATCHY_ITEM_TYPE_* in bits 24..31 */
  uint32_t value;        /* IP results: offset of the entry data in the MMDB data section;
                            pattern results: index of the first id in pattern_ids / data_offsets */
  uint8_t kind;          /* 2 = IP result, 3 = pattern result */
  uint8_t prefix_len;    /* IP results */
  uint16_t n_ids;        /* pattern results: number of pattern ids (literal id first, then glob ids ascending) */
} matchy_scan_hit_t;
/* Limits of the record, not of the matching: a candidate text of 16 MiB or more (24-bit length) and a candidate that matches
 * more than 65535 patterns at once (16-bit id count) make the scan fail with an error instead of truncating. Glob matching
 * itself is unbounded like Paraglob::find_all (any number of results, any nesting of '*': candidates beyond what a lane of
 * the streaming pass holds are answered by a spill pass), and literal queries of case-insensitive databases are lower-cased
 * as a stream (no length limit). */
#define MATCHY_SCAN_HIT_LEN(h) ((h).len_type & 0xFFFFFFu)
#define MATCHY_SCAN_HIT_END(h) ((uint64_t)(h).start + MATCHY_SCAN_HIT_LEN(h))
#define MATCHY_SCAN_HIT_TYPE(h) ((uint8_t)((h).len_type >> 24))

/* Compact form of an IPv4 result (fetch_mode MATCHY_SCAN_FETCH_HITS | MATCHY_SCAN_FETCH_COMPACT), 8 bytes: where nearly every line
 * of a log hits an IP entry (a CIDR-heavy database) the hit records are what crosses PCIe, and half the bytes is half the time.
 * packed: offset of the entry data in the MMDB data section in bits 0..21, length of the matched text - 7 in bits 22..25 (a dotted
 * quad has 7..15 bytes), prefix length (0..32) in bits 26..31. matchy_scan_ip4_hit_expand() gives the 16-byte record back. */
typedef struct matchy_scan_ip4_hit_t {
  uint32_t start;
  uint32_t packed;
} matchy_scan_ip4_hit_t;
#define MATCHY_SCAN_IP4_DATA_BITS 22
static inline matchy_scan_hit_t matchy_scan_ip4_hit_expand(matchy_scan_ip4_hit_t c) {
  matchy_scan_hit_t h;
  h.start = c.start;
  h.len_type = (((c.packed >> MATCHY_SCAN_IP4_DATA_BITS) & 15u) + 7u) | ((uint32_t)MATCHY_ITEM_TYPE_IPV4 << 24);
  h.value = c.packed & ((1u << MATCHY_SCAN_IP4_DATA_BITS) - 1u);
  h.kind = 2;
  h.prefix_len = (uint8_t)(c.packed >> (MATCHY_SCAN_IP4_DATA_BITS + 4));
  h.n_ids = 0;
  return h;
}

typedef struct matchy_scan_result_t {
  const matchy_scan_hit_t *hits;  /* matchy_scanner_scan / fetch_mode 3: canonical order (by start, then chunk-path
                                     class order); fetch_mode 1: device order */
  size_t n_hits;
  const uint32_t *pattern_ids;
  const int64_t *data_offsets;    /* per pattern id: data-section offset or -1 */
  size_t n_ids;
  uint64_t lines;                 /* number of '\n' bytes (WorkerStats::lines_processed) */
  uint64_t candidates;            /* WorkerStats::candidates_tested */
  uint64_t bytes;
  const matchy_scan_ip4_hit_t *ip4_hits; /* MATCHY_SCAN_FETCH_COMPACT: the IPv4 results (they are NOT in `hits` then); else NULL */
  size_t n_ip4_hits;                     /* matches of the scan = n_hits + n_ip4_hits */
  void *_internal;
} matchy_scan_result_t;

/* extract_flags 0 = derive from the database like `matchy match` does (match_cmd.rs:276-303).
 * device = HIP device ordinal. Returns NULL on failure. One scanner per thread. */
matchy_scanner_t *matchy_scanner_create(const matchy_t *db, uint32_t extract_flags, int32_t device);
void matchy_scanner_free(matchy_scanner_t *scanner);
/* Scan a host buffer (copied to the device in newline-aligned pieces of < 1 GiB). len must be below 4 GiB per call
 * (hit offsets are 32 bit); feed larger inputs in batches cut at newlines, like `matchy match` does. */
int32_t matchy_scanner_scan(matchy_scanner_t *scanner, const uint8_t *data, size_t len, matchy_scan_result_t *out);
/* Scan bytes that are already resident in device memory (16-byte aligned, len < 2^31) on `hip_stream`
 * (a hipStream_t, NULL = default stream). fetch_mode: 0 = only counters are read back (n_hits is set, hits is NULL),
 * 1 = hit records in device order (like the reference, whose result order is unspecified); the arrays are BORROWED
 * from the scanner and stay valid until its next scan or matchy_scanner_free; 3 = owned copy in canonical order. */
#define MATCHY_SCAN_FETCH_COUNTS 0u
#define MATCHY_SCAN_FETCH_HITS 1u
/* 4 = the records stay in device memory: hits / pattern_ids / data_offsets are DEVICE pointers borrowed from the scanner
 * (device order, valid until its next scan), only the counters cross the bus. For consumers that aggregate or filter on the
 * GPU, and for inputs where nearly every line hits: 16 bytes per hit over PCIe otherwise bound the scan. */
#define MATCHY_SCAN_FETCH_DEVICE 4u
#define MATCHY_SCAN_FETCH_SORTED 3u
/* 1 | 8 = like 1, but the IPv4 results come as 8-byte records in ip4_hits / n_ip4_hits (same borrowing rules; device order) and
 * everything else in hits as before. Takes effect when the entry data of every IP entry lies in the first 4 MiB of the data section (22-bit offsets) —
 * otherwise, and with any other fetch mode, the flag is ignored and n_ip4_hits is 0: read both arrays. */
#define MATCHY_SCAN_FETCH_COMPACT 8u
int32_t matchy_scanner_scan_device(matchy_scanner_t *scanner, const void *device_ptr, size_t len, void *hip_stream,
                                   uint32_t fetch_mode, matchy_scan_result_t *out);
/* The two halves of matchy_scanner_scan_device: submit launches one batch on `hip_stream` and returns, wait blocks until
 * it is done and hands out its result (one batch in flight per scanner). Two scanners on two streams let one batch's
 * result transfer overlap the next batch's streaming kernel. */
int32_t matchy_scanner_submit_device(matchy_scanner_t *scanner, const void *device_ptr, size_t len, void *hip_stream,
                                     uint32_t fetch_mode);
int32_t matchy_scanner_wait(matchy_scanner_t *scanner, matchy_scan_result_t *out);
void matchy_scan_result_free(matchy_scan_result_t *result);
/* true for a result of fetch_mode MATCHY_SCAN_FETCH_DEVICE: hits / pattern_ids / data_offsets are DEVICE pointers and must not be
 * dereferenced on the host (matchy_scan_hit_to_json returns NULL for such a result). */
bool matchy_scan_result_on_device(const matchy_scan_result_t *result);
/* The NDJSON record `matchy match` prints for hit i (match_processor/parallel.rs:297-369). `text` points at the
 * scanned bytes on the host. Returned string: matchy_free_string(). */
char *matchy_scan_hit_to_json(const matchy_scanner_t *scanner, const matchy_scan_result_t *result, size_t i,
                              const uint8_t *text, const char *source);
/* Every match of a result as NDJSON — the line matchy_scan_hit_to_json returns for each hit, in the order of the arrays (hits, then
 * ip4_hits), '\n' behind every line — in one call: *out is a malloc'ed buffer of *out_len bytes (NUL behind them), released with
 * matchy_free_string. The scanner caches the rendered data payloads by data offset (not thread-safe per scanner, like its scans). */
int32_t matchy_scan_result_to_ndjson(matchy_scanner_t *scanner, const matchy_scan_result_t *result, const uint8_t *text,
                                     const char *source, char **out, size_t *out_len);
/* Per-kernel HIP-event timing of the last scan (recorded on the scan's stream):
 * out[0..4] = k_anchor, k_validate_dom + k_validate, k_rare, k_lookup (incl. writing the hit records), total (milliseconds).
 * matchy_scanner_scan_device runs the kernels behind k_anchor on three streams: then out[1] is that whole tail and out[2] = out[3] = 0. */
void matchy_scanner_set_profile(matchy_scanner_t *scanner, bool enabled);
/* matchy_scanner_scan_device cuts a large batch into slices and runs the kernels behind the streaming pass of one slice beside
 * the streaming pass of the next (results do not depend on it: N4 of the design notes, positions stay absolute). 0 = default for
 * the batch size, 1 = never cut, n = n equal slices (at most 8). matchy_scanner_last_slices: what the last scan used. */
void matchy_scanner_set_slices(matchy_scanner_t *scanner, int32_t slices);
int32_t matchy_scanner_last_slices(const matchy_scanner_t *scanner);
void matchy_scanner_get_timing(const matchy_scanner_t *scanner, float out_ms[5]);
/* Last error message of the calling thread ("" if none). */
/* `matchy query DB QUERY` (bin/commands/query_cmd.rs:8-69) as one call: compact JSON array — one object per matching
 * pattern that carries data (literal first, then globs by id), or the IP entry's data plus "cidr" and "prefix_len", or
 * [] — and *found = the command's exit-status rule. Free with matchy_free_string. NULL on error. */
char *matchy_amd_query_json(const matchy_t *db, const char *query, int32_t *found);
/* matchy_extractor_create with ExtractorBuilder::min_domain_labels (matchy-extractor/src/lib.rs:101-104;
 * `matchy extract --min-labels`); 0 = the default of 2. */
matchy_extractor_t *matchy_amd_extractor_create(uint32_t flags, uint32_t min_domain_labels);
const char *matchy_amd_last_error(void);
/* Diagnostics: states of the flattened Aho-Corasick automaton on the handle's default device; 0 = the database has no glob
 * section or its automaton is walked node by node (flattened table above MATCHY_AMD_DFA_MAX_MB, default 8192), -1 = error. */
int32_t matchy_amd_ac_dfa_states(const matchy_t *db);
/* Diagnostics: 1 when the database's globs are all of the form *LITERAL with one common first byte ("*.evil.com" lists) and the
 * scan decides glob candidates from hashed suffixes of a name instead of walking the automaton; 0 otherwise, -1 = error. */
int32_t matchy_amd_suffix_filter(const matchy_t *db);
/* Page-locked host memory (hipHostMalloc): buffers a host fills with log data and hands to matchy_scanner_scan reach the device
 * by DMA at the bus rate without being pinned per call. `matchy match` reads its input files into such buffers. */
void *matchy_amd_pinned_alloc(size_t bytes);
void matchy_amd_pinned_free(void *ptr);
/* Pin / unpin host memory the caller owns (hipHostRegister; page-aligned ranges): what matchy_scanner_scan does per call for
 * pageable buffers, for hosts that want to do it ahead of the scan (e.g. on a reader thread). */
int32_t matchy_amd_host_register(const void *ptr, size_t bytes);
void matchy_amd_host_unregister(const void *ptr);
/* HIP devices visible to the process (0 when there is none); `matchy match --devices all` */
int32_t matchy_amd_device_count(void);
/* Host topology for multi-GPU scatter / gather (every GPU is fed from host memory over its own PCIe link, and on a
 * two-socket node the H2D rate depends on which socket the feeding thread runs on). NUMA node of a device's PCI function
 * (hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<id>/numa_node; -1 = the platform does not say), and binding of the CALLING thread to
 * the CPUs of that node (sched_setaffinity within the PROCESS's affinity as it was when the library was loaded — not the thread's current
 * mask, which a creating thread may already have narrowed to another node; returns the CPUs it may run on afterwards, 0 = unchanged).
 * matchy_amd_numa_cpus is the mapping itself over any sysfs root (tests): CPUs near `pci_bus_id`, count returned, first `cap` stored. */
int32_t matchy_amd_device_numa_node(int32_t device);
int32_t matchy_amd_bind_thread_to_device(int32_t device);
int32_t matchy_amd_unbind_thread(void); /* back to the process's affinity at load time; CPUs afterwards, 0 = unchanged */
int32_t matchy_amd_numa_cpus(const char *sysfs_root, const char *pci_bus_id, int32_t *out_cpus, size_t cap);

/* ---- Multi-device scanner (additive). The reader -> workers -> ordered gather of the reference's process_files_parallel
 * (crates/matchy/src/processing/parallel.rs:494-505; workers :594-704) behind the C ABI: the host submits newline-aligned batches,
 * one worker thread per entry of `devices` (an entry may repeat: several batches of one GPU in flight) scans them with a scanner of
 * its own, bound to the NUMA node of its GPU, and the host takes the results back in SUBMISSION order. Line blocks are independent
 * and the database is replicated per device: no collective; counters are summed by the caller. */
typedef struct matchy_multi_scanner_t matchy_multi_scanner_t;
typedef struct matchy_multi_batch_t {
  size_t seq;                  /* submission index */
  int32_t status;              /* MATCHY_SUCCESS or the error of this batch (text: matchy_amd_last_error) */
  matchy_scan_result_t result; /* canonical order, offsets relative to `data`; owned by the taker: matchy_scan_result_free */
  const uint8_t *data;
  size_t len;
  void *tag;                   /* as submitted (matchy_multi_scanner_scan_file: the batch's offset in the file) */
  void *payload;               /* what the batch hook returned for this batch, or NULL */
  size_t worker;               /* index into the device list of the worker that scanned it */
} matchy_multi_batch_t;
typedef struct matchy_multi_totals_t { uint64_t batches, bytes, lines, candidates, matches; } matchy_multi_totals_t;
/* Called on the WORKER thread right after a batch's scan (per-hit work such as rendering runs in parallel there); the
 * returned pointer travels with the batch as matchy_multi_batch_t.payload. */
typedef void *(*matchy_multi_batch_fn)(void *user, size_t worker, matchy_scanner_t *scanner, const matchy_scan_result_t *result,
                                       const uint8_t *data, size_t len, void *tag);
/* Called on the gathering thread, batches in order; non-zero stops the scan and is returned. */
typedef int32_t (*matchy_multi_ordered_fn)(void *user, const matchy_multi_batch_t *batch);
/* devices NULL / n_devices 0 = the handle's default device once. extract_flags as matchy_scanner_create. NULL on failure. */
matchy_multi_scanner_t *matchy_multi_scanner_create(const matchy_t *db, uint32_t extract_flags, const int32_t *devices, size_t n_devices);
void matchy_multi_scanner_free(matchy_multi_scanner_t *ms);
size_t matchy_multi_scanner_workers(const matchy_multi_scanner_t *ms);
/* The scanner of a worker (NULL until that worker has seen a batch; worker 0's exists from the start): for matchy_scan_hit_to_json /
 * matchy_scan_result_to_ndjson of a batch that worker scanned (use it from one thread at a time). */
matchy_scanner_t *matchy_multi_scanner_worker_scanner(const matchy_multi_scanner_t *ms, size_t worker);
void matchy_multi_scanner_set_batch_hook(matchy_multi_scanner_t *ms, matchy_multi_batch_fn fn, void *user);
/* Queue one batch (it should end at a line end; len < 4 GiB). The bytes stay the caller's and must remain valid until the batch has
 * been taken with _next. BACK-PRESSURE (the reference's bounded channels, processing/parallel.rs:563-577): blocks while one batch
 * per worker is already waiting, and while matchy_multi_scanner_max_pending() batches (2 x workers + 2) are out — submitted and not
 * yet taken with _next, finished or not. Submit from one thread and gather from another, or interleave on one thread: call _next
 * whenever matchy_multi_scanner_pending() has reached the maximum (a lone thread that only submits would block for good).
 * pinned_range: NULL, or the page range the caller registered for this batch with matchy_amd_host_register — the worker unregisters it
 * after the scan (if submit FAILS the range is still the caller's to unregister). */
int32_t matchy_multi_scanner_submit(matchy_multi_scanner_t *ms, const uint8_t *data, size_t len, void *tag, const void *pinned_range);
/* The same with the NUMA node the batch's bytes live on (-1 = anywhere = matchy_multi_scanner_submit): a worker whose GPU hangs off
 * that node takes it first, a worker of another node only when it has nothing of its own. */
int32_t matchy_multi_scanner_submit_near(matchy_multi_scanner_t *ms, const uint8_t *data, size_t len, void *tag, const void *pinned_range,
                                         int32_t numa_node);
/* NUMA node of a worker's GPU (-1 = unknown) and the number of CPUs its thread bound itself to (0 = not bound); valid once the
 * worker thread has started (after the first _next at the latest). */
int32_t matchy_multi_scanner_worker_numa(const matchy_multi_scanner_t *ms, size_t worker, int32_t *node, int32_t *cpus_bound);
size_t matchy_multi_scanner_pending(const matchy_multi_scanner_t *ms);     /* submitted and not yet taken */
size_t matchy_multi_scanner_max_pending(const matchy_multi_scanner_t *ms); /* the bound submit blocks on */
/* The next batch in submission order (blocks until it is done): 1 = *out filled, 0 = nothing pending, < 0 = error. */
int32_t matchy_multi_scanner_next(matchy_multi_scanner_t *ms, matchy_multi_batch_t *out);
/* One buffer through all workers, cut at newlines into pieces of batch_bytes (0 = chosen from len and the worker count), merged into
 * ONE result with offsets into `data`: the same records matchy_scanner_scan returns for these bytes, whatever the device list. */
int32_t matchy_multi_scanner_scan(matchy_multi_scanner_t *ms, const uint8_t *data, size_t len, size_t batch_bytes, matchy_scan_result_t *out);
/* One input file: a regular file is mapped (its batches are cut, pre-faulted and pinned by a reader thread), "-" / a pipe is read.
 * `fn` (may be NULL) sees every batch in file order on the calling thread, tag = offset of the batch in the input; totals may be NULL. */
int32_t matchy_multi_scanner_scan_file(matchy_multi_scanner_t *ms, const char *path, size_t batch_bytes, matchy_multi_ordered_fn fn,
                                       void *user, matchy_multi_totals_t *totals);
/* Deterministic builds for tests: fixes the build_epoch metadata value. */
int32_t matchy_builder_set_build_epoch(matchy_builder_t *b, uint64_t epoch);

#ifdef __cplusplus
}
#endif
#endif /* MATCHY_AMD_H */
