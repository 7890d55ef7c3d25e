// Prints XXH64 (plain and with ASCII lower-casing folded in) of every prefix length 0..260 of a fixed byte pattern: tests/test_host_units.py
// compares the lines with the independent `xxhash` module (the short-key and tail paths of hashes.h read overlapping words).
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "hashes.h"
int main(){ std::vector<uint8_t> b(300); for(size_t i=0;i<b.size();++i) b[i]=(uint8_t)(i*131+7); for(size_t n=0;n<=260;++n){ printf("%zu %016llx %016llx\n", n, (unsigned long long)mxy::xxh64<false>(b.data(), n, 0), (unsigned long long)mxy::xxh64<true>(b.data(), n, 0)); } }
