// CPU check: Xxh64Stream (bytes pushed one at a time) == xxh64() over the same bytes, all lengths 0..200.
#include <cstdio>
#include <cstdint>
#include <vector>
#include "hashes.h"
using namespace mxy;
int main() {
    int bad = 0;
    std::vector<uint8_t> buf(300);
    for (size_t i = 0; i < buf.size(); ++i) buf[i] = (uint8_t)(i * 131 + 7);
    for (size_t n = 0; n <= 200; ++n)
        for (uint64_t seed : {0ull, 12345ull}) {
            Xxh64Stream st(seed);
            for (size_t i = 0; i < n; ++i) st.push(buf[i]);
            if (st.finish(seed) != xxh64<false>(buf.data(), n, seed)) { ++bad; printf("mismatch n=%zu\n", n); }
        }
    if (xxh64<false>((const uint8_t*)"abc", 3, 0) != 0x44BC2CF5AD770999ull) ++bad;
    if (bad) { printf("FAILED %d\n", bad); return 1; }
    printf("xxh64 stream ok\n");
    return 0;
}
