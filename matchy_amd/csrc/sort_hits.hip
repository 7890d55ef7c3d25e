// Canonical order of the final hit records on the GPU (rocPRIM radix sort through hipCUB): by start offset, then in
// the chunk-path extractor order (IPv6, IPv4, e-mail, domain, hashes, BTC, ETH, XMR: matchy-extractor/src/lib.rs:449-485),
// then by length. The reference's own result order is unspecified (per-worker vectors concatenated); a deterministic
// order is what the C ABI promises for `matchy_scanner_scan` and fetch_mode 3.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "scan_types.h"

namespace mxy {

__device__ __forceinline__ uint32_t d_type_rank(uint32_t t) {
    switch (t) {
        case IT_IPV6: return 0; case IT_IPV4: return 1; case IT_EMAIL: return 2; case IT_DOMAIN: return 3;
        case IT_MD5: case IT_SHA1: case IT_SHA256: case IT_SHA384: case IT_SHA512: return 4;
        case IT_BITCOIN: return 5; case IT_ETHEREUM: return 6; case IT_MONERO: return 7;
    }
    return 8;
}

__global__ void k_sort_keys(const FinalHit* fin, uint32_t n, unsigned long long* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FinalHit h = fin[i];
    // start < 2^31 and length < 2^24 for one launch (engine.cpp checks the batch size)
    keys[i] = ((unsigned long long)h.start << 32) | ((unsigned long long)d_type_rank(h.len_type >> 24) << 24) | (h.len_type & 0xFFFFFFull);
    vals[i] = i;
}

__global__ void k_sort_gather(const FinalHit* fin, const uint32_t* order, uint32_t n, FinalHit* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fin[order[i]];
}

size_t sort_hits_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                            (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n);
    return bytes;
}

// keys/vals: 2 * n entries each (in | out halves); temp: sort_hits_temp_bytes(n); out: n records
hipError_t sort_hits(const FinalHit* fin, uint32_t n, unsigned long long* keys, uint32_t* vals, void* temp, size_t temp_bytes,
                     FinalHit* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const int blocks = (int)((n + 255) / 256);
    hipLaunchKernelGGL(k_sort_keys, dim3(blocks), dim3(256), 0, stream, fin, n, keys, vals);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const unsigned long long*)keys, keys + n, (const uint32_t*)vals,
                                                      vals + n, (int)n, 0, 64, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sort_gather, dim3(blocks), dim3(256), 0, stream, fin, (const uint32_t*)(vals + n), n, out);
    return hipGetLastError();
}

}  // namespace mxy
