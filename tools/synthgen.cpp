// tools/synthgen.cpp — deterministic synthetic data for tests and bench.py (NOT part of the product library).
//
// Implements the workload of SURVEY.md §8(d) / BASELINE.json: nginx "combined" access-log lines and the
// indicator sets of configs C1..C5. Every line and every indicator is a pure function of (seed, index)
// (splitmix64), so the log generator can plant database indicators without any shared state and any line
// range can be produced independently. Shape follows the reference's own generator idea
// (crates/matchy/examples/generate_logs.rs:21-71: fixed pools, rare threat lines) in nginx format.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace {

inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

struct Cfg {
    uint64_t seed;
    uint32_t n_ip, n_cidr, n_dom, n_hash, n_glob;
    uint32_t hit_permille;   // share of lines (per mille) whose client IP / referer is planted from the DB
    uint32_t cidr_mode;      // 0: prefixes /16../30 uniform; 1: BASELINE configs[4] mix (70 % /24, 20 % /16../23, 10 % /32)
};

const char* TLDS[20] = {"com", "net", "org", "info", "biz", "io", "co", "ru", "cn", "de", "uk", "fr", "nl", "br", "in", "xyz", "top", "site", "online", "app"};
const char* WORDS[32] = {"alpha", "bravo", "cargo", "delta", "ember", "fable", "gamma", "haven", "ionic", "jolly", "karma", "lumen", "mango", "noble", "ocean", "pixel",
                         "quark", "raven", "sigma", "tango", "umbra", "vivid", "waltz", "xenon", "yield", "zebra", "amber", "blaze", "coral", "drift", "eagle", "frost"};
const char* UAS[20] = {
    "Mozilla/5.0 (Windows NT 10.0; Win64; x64) AppleWebKit/537.36 (KHTML, like Gecko) Chrome/120.0.0.0 Safari/537.36",
    "Mozilla/5.0 (Macintosh; Intel Mac OS X 10_15_7) AppleWebKit/605.1.15 (KHTML, like Gecko) Version/17.1 Safari/605.1.15",
    "Mozilla/5.0 (X11; Linux x86_64; rv:121.0) Gecko/20100101 Firefox/121.0",
    "Mozilla/5.0 (iPhone; CPU iPhone OS 17_1 like Mac OS X) AppleWebKit/605.1.15 (KHTML, like Gecko) Version/17.1 Mobile/15E148 Safari/604.1",
    "Mozilla/5.0 (Linux; Android 14; Pixel 8) AppleWebKit/537.36 (KHTML, like Gecko) Chrome/119.0.6045.163 Mobile Safari/537.36",
    "curl/8.4.0",
    "python-requests/2.31.0",
    "Go-http-client/2.0",
    "Mozilla/5.0 (compatible; Googlebot/2.1; +http://www.google.com/bot.html)",
    "Mozilla/5.0 (compatible; bingbot/2.0; +http://www.bing.com/bingbot.htm)",
    "Mozilla/5.0 (Windows NT 10.0; Win64; x64) AppleWebKit/537.36 (KHTML, like Gecko) Chrome/118.0.0.0 Safari/537.36 Edg/118.0.2088.76",
    "Wget/1.21.4",
    "okhttp/4.12.0",
    "Mozilla/5.0 (Windows NT 6.1; WOW64; Trident/7.0; rv:11.0) like Gecko",
    "Apache-HttpClient/4.5.14 (Java/17.0.9)",
    "Mozilla/5.0 (X11; Ubuntu; Linux x86_64; rv:109.0) Gecko/20100101 Firefox/115.0",
    "PostmanRuntime/7.35.0",
    "Mozilla/5.0 (iPad; CPU OS 16_6 like Mac OS X) AppleWebKit/605.1.15 (KHTML, like Gecko) CriOS/119.0.6045.169 Mobile/15E148 Safari/604.1",
    "facebookexternalhit/1.1 (+http://www.facebook.com/externalhit_uatext.php)",
    "Slackbot-LinkExpanding 1.0 (+https://api.slack.com/robots)"};
const char* MONTHS[12] = {"Jan", "Feb", "Mar", "Apr", "May", "Jun", "Jul", "Aug", "Sep", "Oct", "Nov", "Dec"};
const char* EXTS[8] = {"html", "php", "js", "css", "png", "json", "jpg", "svg"};

void ipv4_str(uint32_t a, std::string& out) {
    char b[20];
    snprintf(b, sizeof(b), "%u.%u.%u.%u", a >> 24, (a >> 16) & 255, (a >> 8) & 255, a & 255);
    out += b;
}
uint32_t public_v4(uint64_t h) {  // first octet 1..223, never 10/127 so benign and planted pools stay apart from 10.x test nets
    uint32_t a = (uint32_t)(h >> 16);
    uint32_t o = 1 + (uint32_t)((h >> 48) % 223);
    if (o == 10 || o == 127) o = 11;
    return (o << 24) | (a & 0x00FFFFFF);
}
void hex_of(uint64_t seed, int nchars, std::string& out) {
    static const char* H = "0123456789abcdef";
    uint64_t x = seed;
    for (int i = 0; i < nchars; ++i) {
        if (i % 16 == 0) x = splitmix64(x + i);
        out.push_back(H[(x >> (4 * (i % 16))) & 15]);
    }
}

// ---- indicators: kind 0 ip, 1 cidr, 2 domain, 3 hash, 4 glob
void ioc_key(const Cfg& c, int kind, uint32_t i, std::string& out) {
    uint64_t h = splitmix64(c.seed ^ (0xA5A5ull << 32) ^ ((uint64_t)kind << 40) ^ i);
    switch (kind) {
        case 0: ipv4_str(public_v4(h), out); break;
        case 1: {
            // prefixes /16../30: the set covers ~1-2 % of the IPv4 space, so the line hit rate stays a few per cent
            uint32_t p = 16 + (uint32_t)((h >> 8) % 15);
            if (c.cidr_mode == 1) {  // CIDR-heavy feed: mostly /24 blocks, some larger allocations, some single hosts
                const uint32_t r = (uint32_t)((h >> 8) % 100);
                p = r < 70 ? 24 : r < 90 ? 16 + (uint32_t)((h >> 20) % 8) : 32;
            }
            uint32_t a = public_v4(splitmix64(h));
            a &= p == 32 ? 0xFFFFFFFFu : ~((1u << (32 - p)) - 1);
            ipv4_str(a, out);
            out += "/" + std::to_string(p);
            break;
        }
        case 2: out += std::string(WORDS[h & 31]) + WORDS[(h >> 5) & 31] + std::to_string(i) + "." + TLDS[(h >> 10) % 20]; break;
        case 3: { int n = (i % 3 == 0) ? 32 : (i % 3 == 1) ? 40 : 64; hex_of(h, n, out); break; }
        case 4: out += std::string("*.") + WORDS[h & 31] + "-" + WORDS[(h >> 5) & 31] + std::to_string(i) + "." + TLDS[(h >> 10) % 20]; break;
    }
}
void ioc_data(const Cfg& c, int kind, uint32_t i, std::string& out) {
    static const char* LV[4] = {"low", "medium", "high", "critical"};
    static const char* CAT[8] = {"malware", "phishing", "c2", "botnet", "spam", "scanner", "tor-exit", "ransomware"};
    uint64_t h = splitmix64(c.seed ^ (0x5A5Aull << 32) ^ ((uint64_t)kind << 40) ^ i);
    out += std::string("{\"threat_level\":\"") + LV[h & 3] + "\",\"category\":\"" + CAT[(h >> 2) & 7] + "\",\"source\":\"feed-" + std::to_string((h >> 5) & 15) + "\"}";
}

void path_of(uint64_t h, std::string& out) {
    uint32_t t = (uint32_t)(h % 10000);
    uint64_t g = splitmix64(0xBADC0FFEEull ^ t);
    int depth = 1 + (int)(g & 3);
    for (int d = 0; d < depth; ++d) { out.push_back('/'); out += WORDS[(g >> (4 + 5 * d)) & 31]; }
    if ((g >> 40) & 1) { out.push_back('.'); out += EXTS[(g >> 41) & 7]; }
}

void gen_line(const Cfg& c, uint64_t L, std::string& out) {
    uint64_t h = splitmix64(c.seed ^ L);
    uint64_t h2 = splitmix64(h);
    uint64_t h3 = splitmix64(h2);
    // client address
    uint32_t r = (uint32_t)(h % 1000);
    uint32_t n_ipdb = c.n_ip + c.n_cidr;
    if (r < c.hit_permille && n_ipdb) {
        uint32_t j = (uint32_t)((h >> 12) % n_ipdb);
        if (j < c.n_ip) ioc_key(c, 0, j, out);
        else {
            std::string k;
            ioc_key(c, 1, j - c.n_ip, k);
            unsigned a, b, cc, d, p;
            sscanf(k.c_str(), "%u.%u.%u.%u/%u", &a, &b, &cc, &d, &p);
            uint32_t net = (a << 24) | (b << 16) | (cc << 8) | d;
            uint32_t host = (uint32_t)(h2 >> 7) & (p == 32 ? 0u : ((1u << (32 - p)) - 1));
            ipv4_str(net | host, out);
        }
    } else if (r >= 990) {
        char b[64];
        snprintf(b, sizeof(b), "2001:db8:%x::%x", (unsigned)((h >> 20) & 0xFFFF), (unsigned)((h >> 36) & 0xFFFF) | 1u);
        out += b;
    } else {
        ipv4_str(public_v4(splitmix64(0x77ull ^ ((h >> 12) % 1000000))), out);
    }
    // timestamp
    uint64_t sec = L / 50;  // 50 requests per second
    char ts[64];
    snprintf(ts, sizeof(ts), " - - [%02u/%s/2026:%02u:%02u:%02u +0000] \"", (unsigned)(1 + (sec / 86400) % 28), MONTHS[(sec / (86400 * 28)) % 12],
             (unsigned)((sec / 3600) % 24), (unsigned)((sec / 60) % 60), (unsigned)(sec % 60));
    out += ts;
    out += (h2 & 7) ? "GET " : "POST ";
    path_of(h2 >> 3, out);
    if ((h2 >> 20) % 1000 == 0 && c.n_hash) {  // 0.1 %: hash in query string, 10 % of those from the DB
        out += "?h=";
        if ((h2 >> 32) % 10 == 0) ioc_key(c, 3, (uint32_t)((h2 >> 36) % c.n_hash), out);
        else hex_of(h3, ((h2 >> 40) & 1) ? 32 : 64, out);
    }
    static const int STATUS[8] = {200, 200, 200, 200, 301, 304, 404, 500};
    char sb[64];
    snprintf(sb, sizeof(sb), " HTTP/1.1\" %d %u \"", STATUS[(h3 >> 3) & 7], (unsigned)(200 + (h3 >> 8) % 50000));
    out += sb;
    // referer
    uint32_t rr = (uint32_t)((h3 >> 24) % 1000);
    if (rr < 600) out += "-";
    else {
        out += "https://";
        uint32_t pick = (uint32_t)((h3 >> 34) % 1000);
        uint32_t half = c.hit_permille / 2 ? c.hit_permille / 2 : 1;
        if (pick < half && c.n_dom) ioc_key(c, 2, (uint32_t)((h3 >> 44) % c.n_dom), out);
        else if (pick < 2 * half && c.n_glob) {
            std::string g;
            ioc_key(c, 4, (uint32_t)((h3 >> 44) % c.n_glob), g);
            out += std::string(WORDS[(h3 >> 50) & 31]) + g.substr(1);  // "<word>.<glob domain>"
        } else {
            uint64_t d = splitmix64(0x99ull ^ ((h3 >> 40) % 200000));
            out += std::string("site") + std::to_string((unsigned)(d % 200000)) + "." + TLDS[(d >> 20) % 20];
        }
        path_of(h3 >> 5, out);
    }
    out += "\" \"";
    out += UAS[(h >> 50) % 20];
    out += "\"\n";
}

// ---- other log shapes (bench.py --log-shape, BASELINE.md: a scan engine's number is only as good as its worst common input).
// Every line is still a pure function of (seed, line index); database indicators are planted at the same per-mille rate.
void planted_or_benign_ip(const Cfg& c, uint64_t h, std::string& out) {
    const uint32_t n_ipdb = c.n_ip + c.n_cidr;
    if ((uint32_t)(h % 1000) < c.hit_permille && c.n_ip) ioc_key(c, 0, (uint32_t)((h >> 12) % c.n_ip), out);
    else if ((uint32_t)(h % 1000) < c.hit_permille && n_ipdb) {
        std::string k;
        ioc_key(c, 1, (uint32_t)((h >> 12) % c.n_cidr), k);
        unsigned a, b, cc, d, p;
        sscanf(k.c_str(), "%u.%u.%u.%u/%u", &a, &b, &cc, &d, &p);
        ipv4_str((a << 24) | (b << 16) | (cc << 8) | d, out);
    } else ipv4_str(public_v4(splitmix64(0x77ull ^ ((h >> 12) % 1000000))), out);
}
void planted_or_benign_domain(const Cfg& c, uint64_t h, std::string& out) {
    const uint32_t pick = (uint32_t)(h % 1000), half = c.hit_permille / 2 ? c.hit_permille / 2 : 1;
    if (pick < half && c.n_dom) ioc_key(c, 2, (uint32_t)((h >> 12) % c.n_dom), out);
    else if (pick < 2 * half && c.n_glob) {
        std::string g;
        ioc_key(c, 4, (uint32_t)((h >> 12) % c.n_glob), g);
        out += std::string(WORDS[(h >> 50) & 31]) + g.substr(1);
    } else {
        const uint64_t d = splitmix64(0x99ull ^ ((h >> 12) % 200000));
        if ((d >> 30) & 1) out += ((d >> 31) & 1) ? "www." : "cdn-assets.";
        out += std::string("site") + std::to_string((unsigned)(d % 200000)) + "." + TLDS[(d >> 20) % 20];
    }
}
void planted_or_random_hash(const Cfg& c, uint64_t h, int nchars, std::string& out) {
    if ((uint32_t)(h % 1000) < c.hit_permille && c.n_hash) ioc_key(c, 3, (uint32_t)((h >> 12) % c.n_hash), out);
    else hex_of(splitmix64(h ^ 0xABCDull), nchars, out);
}
void iso_time(uint64_t L, std::string& out) {
    const uint64_t sec = L / 50;
    char ts[64];
    snprintf(ts, sizeof(ts), "2026-01-%02uT%02u:%02u:%02u.%03uZ", (unsigned)(1 + (sec / 86400) % 28), (unsigned)((sec / 3600) % 24), (unsigned)((sec / 60) % 60),
             (unsigned)(sec % 60), (unsigned)((L % 50) * 20));
    out += ts;
}
// 1: application log as JSON lines (structured fields, quoted strings, few dots outside timestamps / addresses)
void gen_jsonl(const Cfg& c, uint64_t L, std::string& out) {
    static const char* LEVEL[4] = {"info", "info", "warn", "error"};
    static const char* SVC[8] = {"auth", "billing", "search", "gateway", "mailer", "orders", "profile", "ingest"};
    const uint64_t h = splitmix64(c.seed ^ (0x1111ull << 48) ^ L), h2 = splitmix64(h), h3 = splitmix64(h2);
    out += "{\"ts\":\"";
    iso_time(L, out);
    out += std::string("\",\"level\":\"") + LEVEL[h & 3] + "\",\"service\":\"" + SVC[(h >> 2) & 7] + "\",\"client_ip\":\"";
    planted_or_benign_ip(c, h2, out);
    out += "\",\"host\":\"";
    planted_or_benign_domain(c, h3, out);
    out += "\",\"user\":\"";
    if ((h >> 8) % 4 == 0) { out += std::string(WORDS[(h >> 12) & 31]) + "." + WORDS[(h >> 17) & 31] + "@"; planted_or_benign_domain(c, splitmix64(h3), out); }
    else out += std::string("u") + std::to_string((unsigned)((h >> 12) % 1000000));
    out += std::string("\",\"msg\":\"") + WORDS[(h2 >> 20) & 31] + " " + WORDS[(h2 >> 25) & 31] + " " + WORDS[(h2 >> 30) & 31] + " took " + std::to_string((unsigned)((h2 >> 35) % 5000)) + "ms\",\"trace_id\":\"";
    hex_of(h3, 32, out);
    out += std::string("\",\"status\":") + std::to_string(200 + (unsigned)((h3 >> 40) % 4) * 100) + "}\n";
}
// 2: firewall / flow log: several addresses per line, hardly anything else
void gen_ip_dense(const Cfg& c, uint64_t L, std::string& out) {
    const uint64_t h = splitmix64(c.seed ^ (0x2222ull << 48) ^ L);
    char b[96];
    snprintf(b, sizeof(b), "Jan %2u %02u:%02u:%02u fw7 kernel: ACCEPT IN=eth%u OUT=eth%u ", (unsigned)(1 + (L / 4320000) % 28), (unsigned)((L / 180000) % 24),
             (unsigned)((L / 3000) % 60), (unsigned)((L / 50) % 60), (unsigned)(h & 3), (unsigned)((h >> 2) & 3));
    out += b;
    uint64_t g = h;
    out += "SRC="; planted_or_benign_ip(c, g = splitmix64(g), out);
    out += " DST="; planted_or_benign_ip(c, g = splitmix64(g), out);
    out += " NAT="; planted_or_benign_ip(c, g = splitmix64(g), out);
    out += ",";     planted_or_benign_ip(c, g = splitmix64(g), out);
    snprintf(b, sizeof(b), " LEN=%u TTL=%u PROTO=TCP SPT=%u DPT=%u via ", (unsigned)(40 + (h >> 8) % 1400), (unsigned)(32 + (h >> 20) % 96), (unsigned)(1024 + (h >> 28) % 60000), (unsigned)((h >> 44) % 1024));
    out += b;
    planted_or_benign_ip(c, g = splitmix64(g), out);
    out += " ";
    planted_or_benign_ip(c, g = splitmix64(g), out);
    if ((h >> 60) == 0) { snprintf(b, sizeof(b), " fe80::%x 2001:db8:%x::%x", (unsigned)(h >> 12) & 0xFFFF, (unsigned)(h >> 20) & 0xFFFF, (unsigned)((h >> 36) & 0xFFFF) | 1u); out += b; }
    out += "\n";
}
// 3: proxy log with long URLs: many host names and file names with extensions per line
void gen_url_heavy(const Cfg& c, uint64_t L, std::string& out) {
    const uint64_t h = splitmix64(c.seed ^ (0x3333ull << 48) ^ L);
    char b[64];
    snprintf(b, sizeof(b), "%llu.%03u %6u ", (unsigned long long)(1767225600ull + L / 50), (unsigned)((L % 50) * 20), (unsigned)(h % 900000));
    out += b;
    uint64_t g = h;
    planted_or_benign_ip(c, g = splitmix64(g), out);
    out += " TCP_MISS/200 GET https://";
    planted_or_benign_domain(c, g = splitmix64(g), out);
    path_of(g >> 7, out);
    out += "?ref=http://";
    planted_or_benign_domain(c, g = splitmix64(g), out);
    path_of(g >> 9, out);
    out += "&img=https://static.";
    planted_or_benign_domain(c, g = splitmix64(g), out);
    out += std::string("/") + WORDS[(g >> 40) & 31] + "_" + WORDS[(g >> 45) & 31] + ".min.js&next=";
    planted_or_benign_domain(c, g = splitmix64(g), out);
    out += "/index.html - DIRECT/";
    planted_or_benign_domain(c, g = splitmix64(g), out);
    out += " text/html\n";
}
// 4: endpoint-detection style log: two or three file hashes per line
void gen_hash_dense(const Cfg& c, uint64_t L, std::string& out) {
    const uint64_t h = splitmix64(c.seed ^ (0x4444ull << 48) ^ L);
    iso_time(L, out);
    out += std::string(" host-") + std::to_string((unsigned)(h % 5000)) + " proc=" + WORDS[(h >> 13) & 31] + ".exe pid=" + std::to_string((unsigned)((h >> 18) % 65536)) + " sha256=";
    uint64_t g = h;
    planted_or_random_hash(c, g = splitmix64(g), 64, out);
    out += " md5=";
    planted_or_random_hash(c, g = splitmix64(g), 32, out);
    if ((h >> 40) & 1) { out += " sha1="; planted_or_random_hash(c, g = splitmix64(g), 40, out); }
    out += std::string(" path=C:/Users/") + WORDS[(h >> 41) & 31] + "/AppData/" + WORDS[(h >> 46) & 31] + "/" + WORDS[(h >> 51) & 31] + ".dll parent=";
    hex_of(splitmix64(g), 16, out);   // 16 hex digits: too short for any hash type
    out += "\n";
}
// 5: skewed halves: of every `period` lines the first half is prose without a dot, a colon or a digit run (nothing to extract:
// the streaming pass runs its front end only), the second half is the address-dense shape — statically assigned equal
// segments then finish at very different times
void gen_skewed(const Cfg& c, uint64_t L, uint64_t period, std::string& out) {
    if (period < 2) period = 2;
    if (L % period >= period / 2) { gen_ip_dense(c, L, out); return; }
    const uint64_t h = splitmix64(c.seed ^ (0x5555ull << 48) ^ L);
    uint64_t g = h;
    const int words = 18 + (int)(h & 15);
    for (int k = 0; k < words; ++k) {
        if ((k & 7) == 0) g = splitmix64(g);
        if (k) out.push_back(' ');
        out += WORDS[(g >> (5 * (k & 7))) & 31];
    }
    out += "\n";
}
void gen_shape(const Cfg& c, int shape, uint64_t param, uint64_t L, std::string& out) {
    switch (shape) {
        case 1: gen_jsonl(c, L, out); break;
        case 2: gen_ip_dense(c, L, out); break;
        case 3: gen_url_heavy(c, L, out); break;
        case 4: gen_hash_dense(c, L, out); break;
        case 5: gen_skewed(c, L, param, out); break;
        default: gen_line(c, L, out); break;
    }
}

}  // namespace

extern "C" {

struct synth_cfg_t { uint64_t seed; uint32_t n_ip, n_cidr, n_dom, n_hash, n_glob, hit_permille, cidr_mode; };

static Cfg to_cfg(const synth_cfg_t* c) { return Cfg{c->seed, c->n_ip, c->n_cidr, c->n_dom, c->n_hash, c->n_glob, c->hit_permille, c->cidr_mode}; }

// Writes lines [first, first+n) into out (capacity cap). Returns bytes written, or the required size if cap is too small.
size_t synth_log(const synth_cfg_t* cfg, uint64_t first, uint64_t n, uint8_t* out, size_t cap) {
    Cfg c = to_cfg(cfg);
    size_t pos = 0;
    std::string line;
    for (uint64_t L = first; L < first + n; ++L) {
        line.clear();
        gen_line(c, L, line);
        if (pos + line.size() <= cap) memcpy(out + pos, line.data(), line.size());
        pos += line.size();
    }
    return pos;
}
// The same for another log shape: 0 nginx (synth_log), 1 jsonl-app, 2 ip-dense, 3 url-heavy, 4 hash-dense, 5 skewed-halves (param = period in lines)
size_t synth_log_shape(const synth_cfg_t* cfg, int shape, uint64_t param, uint64_t first, uint64_t n, uint8_t* out, size_t cap) {
    Cfg c = to_cfg(cfg);
    size_t pos = 0;
    std::string line;
    for (uint64_t L = first; L < first + n; ++L) {
        line.clear();
        gen_shape(c, shape, param, L, line);
        if (pos + line.size() <= cap) memcpy(out + pos, line.data(), line.size());
        pos += line.size();
    }
    return pos;
}

size_t synth_ioc_key(const synth_cfg_t* cfg, int kind, uint32_t i, char* out, size_t cap) {
    std::string s;
    ioc_key(to_cfg(cfg), kind, i, s);
    if (s.size() + 1 <= cap) memcpy(out, s.c_str(), s.size() + 1);
    return s.size();
}
size_t synth_ioc_data(const synth_cfg_t* cfg, int kind, uint32_t i, char* out, size_t cap) {
    std::string s;
    ioc_data(to_cfg(cfg), kind, i, s);
    if (s.size() + 1 <= cap) memcpy(out, s.c_str(), s.size() + 1);
    return s.size();
}
// Feeds every indicator of the config, in CSV row order, to `add(builder, key, json)` (the product's matchy_builder_add);
// `glob_prefix` writes string keys as glob:{key}. Returns the number of entries fed, or -(index + 1) of the first rejected one.
typedef int32_t (*synth_add_fn)(void* builder, const char* key, const char* json);
long long synth_ioc_feed(const synth_cfg_t* cfg, int glob_prefix, synth_add_fn add, void* builder) {
    const Cfg c = to_cfg(cfg);
    const uint32_t counts[5] = {c.n_ip, c.n_cidr, c.n_dom, c.n_hash, c.n_glob};
    long long fed = 0;
    std::string k, d;
    for (int kind = 0; kind < 5; ++kind)
        for (uint32_t i = 0; i < counts[kind]; ++i) {
            k.clear(); d.clear();
            if (glob_prefix && kind >= 2) k = "glob:";
            ioc_key(c, kind, i, k);
            ioc_data(c, kind, i, d);
            if (add(builder, k.c_str(), d.c_str()) != 0) return -(fed + 1);
            ++fed;
        }
    return fed;
}

}  // extern "C"
