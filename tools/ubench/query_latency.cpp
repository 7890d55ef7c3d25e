// Latency of matchy_query through the C ABI: uncached (cache_capacity 0), cached (every query repeated; default
// cache), and T threads on one handle. Usage: query_latency DB.mxy QUERIES.txt [threads]
// Build: g++ -O2 -std=c++17 -I include tools/ubench/query_latency.cpp -o /tmp/query_latency -L matchy_amd/lib -lmatchy_amd -lpthread -Wl,-rpath,$PWD/matchy_amd/lib
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "matchy_amd.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: query_latency DB QUERIES [threads]\n"); return 2; }
    const int threads = argc > 3 ? atoi(argv[3]) : 8;
    std::vector<std::string> qs;
    { std::ifstream f(argv[2]); std::string l; while (std::getline(f, l)) qs.push_back(l); }
    if (qs.empty()) return 2;
    matchy_open_options_t o;
    matchy_init_open_options(&o);
    o.cache_capacity = 0;
    matchy_t* db0 = matchy_open_with_options(argv[1], &o);
    matchy_t* db1 = matchy_open(argv[1]);
    if (!db0 || !db1) { fprintf(stderr, "open failed: %s\n", matchy_amd_last_error()); return 1; }
    auto run = [&](matchy_t* db, size_t from, size_t to, int reps, size_t& found) {
        for (int r = 0; r < reps; ++r)
            for (size_t i = from; i < to; ++i) {
                matchy_result_t res = matchy_query(db, qs[i].c_str());
                found += res.found;
                matchy_free_result(&res);
            }
    };
    size_t found = 0;
    run(db0, 0, std::min<size_t>(qs.size(), 200), 1, found);   // warm-up (first query builds the host tables / the query scanner)
    found = 0;
    double t0 = now();
    run(db0, 0, qs.size(), 1, found);
    double t1 = now();
    printf("uncached, 1 thread : %9.3f us/query  %10.0f q/s  (%zu queries, %zu found)\n", (t1 - t0) / qs.size() * 1e6, qs.size() / (t1 - t0), qs.size(), found);
    // cached: the first 5000 queries over and over (default cache holds 10 000)
    const size_t nc = std::min<size_t>(qs.size(), 5000);
    found = 0;
    run(db1, 0, nc, 1, found);
    t0 = now();
    run(db1, 0, nc, 10, found);
    t1 = now();
    printf("cached,   1 thread : %9.3f us/query  %10.0f q/s\n", (t1 - t0) / (nc * 10) * 1e6, nc * 10 / (t1 - t0));
    // T threads, uncached, one handle
    std::vector<std::thread> th;
    std::vector<size_t> fnd(threads, 0);
    t0 = now();
    for (int t = 0; t < threads; ++t) th.emplace_back([&, t] { run(db0, qs.size() * t / threads, qs.size() * (t + 1) / threads, 4, fnd[t]); });
    for (auto& x : th) x.join();
    t1 = now();
    printf("uncached, %d threads: %9.3f us/query (wall / queries)  %10.0f q/s\n", threads, (t1 - t0) / (qs.size() * 4) * 1e6, qs.size() * 4 / (t1 - t0));
    matchy_close(db0); matchy_close(db1);
    return 0;
}
