// Direct-lookup tables of the IPv4 part of the MMDB search tree, built ON THE DEVICE when a database is opened (DeviceDb::upload):
// the tree itself is uploaded once (one uint2 {left, right} per node), then
//   k_ip_l24    one thread per /24 prefix walks the first 24 levels of SearchTree::lookup_v4 (matchy-format/src/mmdb/tree.rs:46-90)
//               and writes the outcome (DevDb::ip_l24) and the /24 occupancy bit (DevDb::ip_bm24);
//   k_ip_leaf_* every /24 that is not decided after 24 levels gets a leaf table with the outcome for each of its 256 addresses
//               (DevDb::ip_leaf), so that a lookup is two dependent loads (device_shared.h trie_v4).
// 16.8 M walks of <= 24 dependent loads on an L2-resident tree take about a millisecond here; the same loops on the host cost
// ~0.1 s of every matchy_open.
#include <hip/hip_runtime.h>

#include "scan_types.h"

namespace mxy {

// outcome of walking `levels` levels from `node` along the top bits of `bits` (left-aligned in 32 bits): entry encoding of
// DevDb::ip_l1 / ip_l24 — x = kind | prefix << 8 (kind 0 continue at node y, 1 not found, 2 found with data offset y)
__device__ __forceinline__ uint2 walk_levels(const uint2* nodes, uint32_t node_count, uint32_t node, uint32_t bits, uint32_t depth0, uint32_t levels) {
    for (uint32_t k = 0; k < levels; ++k) {
        const uint2 nd = nodes[node];
        const uint32_t rec = ((bits >> (31 - k)) & 1u) ? nd.y : nd.x;
        if (rec == node_count) return make_uint2(1u, 0u);
        if (rec < node_count) node = rec;
        else {
            const uint32_t off = rec - node_count;
            return off < 16 ? make_uint2(1u, 0u) : make_uint2(2u | ((depth0 + k + 1) << 8), off - 16);
        }
    }
    return make_uint2(0u, node);
}

__global__ __launch_bounds__(256) void k_ip_l24(const uint2* nodes, uint32_t node_count, uint32_t v4_start, uint2* l24, uint32_t* bm24, uint32_t* n_undecided) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;   // grid covers exactly 2^24 prefixes
    const uint2 e = node_count ? walk_levels(nodes, node_count, v4_start, v << 8, 0u, 24u) : make_uint2(1u, 0u);
    l24[v] = e;
    // bit v of the /24 bitmap: clear iff every address under the prefix is "not found" within the first 24 levels
    const unsigned long long m = __ballot((e.x & 0xFFu) != 1u);
    if ((threadIdx.x & 63u) == 0) { bm24[v >> 5] = (uint32_t)m; bm24[(v >> 5) + 1] = (uint32_t)(m >> 32); }
    const unsigned long long u = __ballot((e.x & 0xFFu) == 0u);
    if ((threadIdx.x & 63u) == 0 && u) atomicAdd(n_undecided, (uint32_t)__popcll(u));
}

// undecided /24s get leaf indices (in no particular order); those beyond `cap` keep kind 0 (the lookup walks the tree from there)
__global__ __launch_bounds__(256) void k_ip_leaf_assign(uint2* l24, uint32_t* next, uint32_t cap, uint32_t* leaf_node) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    const uint2 e = l24[v];
    if ((e.x & 0xFFu) != 0u) return;
    const uint32_t idx = atomicAdd(next, 1u);
    if (idx >= cap) return;
    leaf_node[idx] = e.y;
    l24[v] = make_uint2(3u, idx);
}

__global__ __launch_bounds__(256) void k_ip_leaf_fill(const uint2* nodes, uint32_t node_count, const uint32_t* leaf_node, uint2* leaf) {
    const uint32_t idx = blockIdx.x, a = threadIdx.x;   // one workgroup per leaf table, one thread per address of the /24
    uint2 e = walk_levels(nodes, node_count, leaf_node[idx], a << 24, 24u, 8u);
    if ((e.x & 0xFFu) == 0u) e = make_uint2(1u, 0u);   // a node below bit 32 answers nothing (tree.rs:46-90: the walk ends after 32 bits)
    leaf[(size_t)idx * 256 + a] = e;
}

void launch_ip_l24(const uint2* nodes, uint32_t node_count, uint32_t v4_start, uint2* l24, uint32_t* bm24, uint32_t* n_undecided, hipStream_t s) {
    hipLaunchKernelGGL(k_ip_l24, dim3((1u << 24) / 256), dim3(256), 0, s, nodes, node_count, v4_start, l24, bm24, n_undecided);
    check_launch("k_ip_l24");
}
void launch_ip_leaf(const uint2* nodes, uint32_t node_count, uint2* l24, uint32_t* next, uint32_t n_leaf, uint32_t* leaf_node, uint2* leaf, hipStream_t s) {
    hipLaunchKernelGGL(k_ip_leaf_assign, dim3((1u << 24) / 256), dim3(256), 0, s, l24, next, n_leaf, leaf_node);
    check_launch("k_ip_leaf_assign");
    hipLaunchKernelGGL(k_ip_leaf_fill, dim3(n_leaf), dim3(256), 0, s, nodes, node_count, leaf_node, leaf);
    check_launch("k_ip_leaf_fill");
}

}  // namespace mxy
