// Canonical order of the final hit records on the GPU: by start offset, then in the chunk-path extractor order (IPv6, IPv4,
// e-mail, domain, hashes, BTC, ETH, XMR: matchy-extractor/src/lib.rs:449-485), then by length. The reference's own result
// order is unspecified (per-worker vectors concatenated); a deterministic order is what the C ABI promises for
// `matchy_scanner_scan` and fetch_mode 3.
//
// A stable LSD radix sort of (64-bit key, record index) pairs, written for this job (round 3; rounds 1-2 called rocPRIM through
// hipCUB): 8-bit digits, one histogram pass over all eight digits while the keys are built, and per digit
//   k_sort_block_hist   digit counts of every 2048-key tile,
//   k_sort_scan         tile offsets per digit (one workgroup per 256 tiles, thread d = digit d: running sums down the columns),
//   k_sort_scatter      tiles again, 256 keys per round in index order; the rank of a key among the keys of its round with the
//                       same digit comes from eight ballots (match-any over the digit's bits) + a count of earlier waves in LDS.
// Digits that are the same in every key (the high bytes of short lengths and of offsets below the batch size: usually three or
// four of the eight) are found by k_sort_plan from the global histogram and their passes return at once; the ping-pong halves each
// pass reads and writes are planned there too, so nothing comes back to the host between the launches.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "scan_types.h"

namespace mxy {

namespace {

constexpr uint32_t SORT_THREADS = 256, SORT_ITEMS = 8, SORT_TILE = SORT_THREADS * SORT_ITEMS, SORT_BINS = 256, SORT_PASSES = 8;

// temp layout (32-bit words): plan | global histogram | tile histograms / offsets
struct SortPlan {
    uint32_t skip[SORT_PASSES];   // 1: every key has the same digit here
    uint32_t src[SORT_PASSES];    // half (0 / 1) of keys / vals the pass reads; it writes the other
    uint32_t final_half;          // half that holds the sorted pairs after the last pass
    uint32_t pad[3];
};
constexpr size_t PLAN_WORDS = sizeof(SortPlan) / 4, GHIST_WORDS = SORT_PASSES * SORT_BINS;

__device__ __forceinline__ uint32_t d_type_rank(uint32_t t) {
    switch (t) {
        case IT_IPV6: return 0; case IT_IPV4: return 1; case IT_EMAIL: return 2; case IT_DOMAIN: return 3;
        case IT_MD5: case IT_SHA1: case IT_SHA256: case IT_SHA384: case IT_SHA512: return 4;
        case IT_BITCOIN: return 5; case IT_ETHEREUM: return 6; case IT_MONERO: return 7;
    }
    return 8;
}

// keys + identity permutation into half 0, and the histogram of all eight digits of all keys
__global__ __launch_bounds__(SORT_THREADS) void k_sort_keys(const FinalHit* fin, uint32_t n, unsigned long long* keys, uint32_t* vals, uint32_t* ghist) {
    __shared__ uint32_t h[SORT_PASSES * SORT_BINS];
    for (uint32_t i = threadIdx.x; i < SORT_PASSES * SORT_BINS; i += SORT_THREADS) h[i] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * SORT_THREADS + threadIdx.x; i < n; i += gridDim.x * SORT_THREADS) {
        const FinalHit f = fin[i];
        // start < 2^31 and length < 2^24 for one launch (engine.cpp checks the batch size)
        const unsigned long long k = ((unsigned long long)f.start << 32) | ((unsigned long long)d_type_rank(f.len_type >> 24) << 24) | (f.len_type & 0xFFFFFFull);
        keys[i] = k;
        vals[i] = i;
#pragma unroll
        for (uint32_t p = 0; p < SORT_PASSES; ++p) atomicAdd(&h[p * SORT_BINS + ((uint32_t)(k >> (8 * p)) & 0xFFu)], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < SORT_PASSES * SORT_BINS; i += SORT_THREADS) if (h[i]) atomicAdd(&ghist[i], h[i]);
}

// which passes have work, and which half each of them reads
__global__ void k_sort_plan(const uint32_t* ghist, uint32_t n, SortPlan* plan) {
    __shared__ uint32_t skip[SORT_PASSES];
    if (threadIdx.x < SORT_PASSES) skip[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < SORT_PASSES * SORT_BINS; i += blockDim.x) if (ghist[i] == n) skip[i / SORT_BINS] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t cur = 0;
        for (uint32_t p = 0; p < SORT_PASSES; ++p) {
            plan->skip[p] = skip[p];
            plan->src[p] = cur;
            if (!skip[p]) cur ^= 1u;
        }
        plan->final_half = cur;
    }
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_block_hist(const unsigned long long* keys, uint32_t n, uint32_t pass, const SortPlan* plan, uint32_t* bhist) {
    if (plan->skip[pass]) return;
    __shared__ uint32_t h[SORT_BINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long* src = keys + (size_t)plan->src[pass] * n;
    const uint32_t base = blockIdx.x * SORT_TILE;
#pragma unroll
    for (uint32_t r = 0; r < SORT_ITEMS; ++r) {
        const uint32_t i = base + r * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(src[i] >> (8 * pass)) & 0xFFu], 1u);
    }
    __syncthreads();
    bhist[(size_t)blockIdx.x * SORT_BINS + threadIdx.x] = h[threadIdx.x];
}

// tile histograms -> first output index of every (tile, digit): keys with a smaller digit first, then earlier tiles. One workgroup per
// chunk of SCAN_CHUNK tiles, thread d = digit d: the chunk's starting value is the digit's global prefix plus the counts of all tiles in
// front of the chunk (read again by every workgroup: a few thousand independent coalesced loads at the most), then a running sum over
// the chunk's own tiles.
constexpr uint32_t SCAN_CHUNK = 256;
__global__ __launch_bounds__(SORT_BINS) void k_sort_scan(const uint32_t* __restrict__ ghist, uint32_t pass, const SortPlan* __restrict__ plan,
                                                         const uint32_t* __restrict__ bhist, uint32_t* __restrict__ boff, uint32_t nblocks) {
    if (plan->skip[pass]) return;
    __shared__ uint32_t s[SORT_BINS];
    const uint32_t d = threadIdx.x;
    const uint32_t own = ghist[pass * SORT_BINS + d];
    s[d] = own;
    __syncthreads();
    uint32_t incl = own;   // inclusive prefix over the 256 digit totals
    for (uint32_t off = 1; off < SORT_BINS; off <<= 1) {
        const uint32_t t = d >= off ? s[d - off] : 0u;
        __syncthreads();
        incl += t;
        s[d] = incl;
        __syncthreads();
    }
    uint32_t run = incl - own;
    const uint32_t first = blockIdx.x * SCAN_CHUNK, last = min(first + SCAN_CHUNK, nblocks);
    uint32_t before = 0;
#pragma unroll 8
    for (uint32_t b = 0; b < first; ++b) before += bhist[(size_t)b * SORT_BINS + d];
    run += before;
    for (uint32_t b = first; b < last; ++b) {
        boff[(size_t)b * SORT_BINS + d] = run;
        run += bhist[(size_t)b * SORT_BINS + d];
    }
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(unsigned long long* keys, uint32_t* vals, uint32_t n, uint32_t pass, const SortPlan* plan, const uint32_t* boff) {
    if (plan->skip[pass]) return;
    __shared__ uint32_t run[SORT_BINS];            // keys of this tile already placed, per digit
    __shared__ uint32_t cnt[SORT_THREADS / 64][SORT_BINS];   // keys of the current round, per wave and digit
    const uint32_t half = plan->src[pass];
    const unsigned long long* ksrc = keys + (size_t)half * n;
    const uint32_t* vsrc = vals + (size_t)half * n;
    unsigned long long* kdst = keys + (size_t)(half ^ 1u) * n;
    uint32_t* vdst = vals + (size_t)(half ^ 1u) * n;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t lt = (1ull << lane) - 1ull;
    run[threadIdx.x] = boff[(size_t)blockIdx.x * SORT_BINS + threadIdx.x];
    const uint32_t base = blockIdx.x * SORT_TILE;
    for (uint32_t r = 0; r < SORT_ITEMS; ++r) {
#pragma unroll
        for (uint32_t w = 0; w < SORT_THREADS / 64; ++w) cnt[w][threadIdx.x] = 0;
        __syncthreads();
        const uint32_t i = base + r * SORT_THREADS + threadIdx.x;
        const bool valid = i < n;
        unsigned long long k = 0;
        uint32_t v = 0, digit = 0;
        if (valid) { k = ksrc[i]; v = vsrc[i]; digit = (uint32_t)(k >> (8 * pass)) & 0xFFu; }
        // lanes of this wave with the same digit (match-any from eight ballots)
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (uint32_t bit = 0; bit < 8; ++bit) {
            const bool mine = (digit >> bit) & 1u;
            const uint64_t bal = __ballot(valid && mine);
            peers &= mine ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & lt);
        if (valid && rank == 0) cnt[wave][digit] = (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t o = run[digit] + rank;
            for (uint32_t w = 0; w < wave; ++w) o += cnt[w][digit];
            kdst[o] = k;
            vdst[o] = v;
        }
        __syncthreads();
        uint32_t add = 0;
#pragma unroll
        for (uint32_t w = 0; w < SORT_THREADS / 64; ++w) add += cnt[w][threadIdx.x];
        run[threadIdx.x] += add;
        __syncthreads();
    }
}

__global__ void k_sort_gather(const FinalHit* fin, const uint32_t* vals, uint32_t n, const SortPlan* plan, FinalHit* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fin[vals[(size_t)plan->final_half * n + i]];
}

}  // namespace

size_t sort_hits_temp_bytes(uint32_t n) {
    const size_t nblocks = ((size_t)n + SORT_TILE - 1) / SORT_TILE;
    return (PLAN_WORDS + GHIST_WORDS + 2 * nblocks * SORT_BINS) * 4;   // plan, digit totals, tile histograms, tile offsets
}

// keys/vals: 2 * n entries each (two halves); temp: sort_hits_temp_bytes(n); out: n records
hipError_t sort_hits(const FinalHit* fin, uint32_t n, unsigned long long* keys, uint32_t* vals, void* temp, size_t temp_bytes,
                     FinalHit* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (temp_bytes < sort_hits_temp_bytes(n)) return hipErrorInvalidValue;
    uint32_t* words = reinterpret_cast<uint32_t*>(temp);
    SortPlan* plan = reinterpret_cast<SortPlan*>(words);
    uint32_t* ghist = words + PLAN_WORDS;
    uint32_t* bhist = ghist + GHIST_WORDS;
    const uint32_t nblocks = (n + SORT_TILE - 1) / SORT_TILE;
    uint32_t* boff = bhist + (size_t)nblocks * SORT_BINS;
    hipError_t e = hipMemsetAsync(ghist, 0, GHIST_WORDS * 4, stream);
    if (e != hipSuccess) return e;
    const int kb = (int)std::min<uint32_t>((n + SORT_THREADS - 1) / SORT_THREADS, 1024u);
    hipLaunchKernelGGL(k_sort_keys, dim3(kb), dim3(SORT_THREADS), 0, stream, fin, n, keys, vals, ghist);
    hipLaunchKernelGGL(k_sort_plan, dim3(1), dim3(256), 0, stream, (const uint32_t*)ghist, n, plan);
    for (uint32_t p = 0; p < SORT_PASSES; ++p) {
        hipLaunchKernelGGL(k_sort_block_hist, dim3(nblocks), dim3(SORT_THREADS), 0, stream, (const unsigned long long*)keys, n, p, (const SortPlan*)plan, bhist);
        hipLaunchKernelGGL(k_sort_scan, dim3((nblocks + SCAN_CHUNK - 1) / SCAN_CHUNK), dim3(SORT_BINS), 0, stream, (const uint32_t*)ghist, p,
                           (const SortPlan*)plan, (const uint32_t*)bhist, boff, nblocks);
        hipLaunchKernelGGL(k_sort_scatter, dim3(nblocks), dim3(SORT_THREADS), 0, stream, keys, vals, n, p, (const SortPlan*)plan, (const uint32_t*)boff);
    }
    hipLaunchKernelGGL(k_sort_gather, dim3((n + 255) / 256), dim3(256), 0, stream, fin, (const uint32_t*)vals, n, (const SortPlan*)plan, out);
    return hipGetLastError();
}

}  // namespace mxy
