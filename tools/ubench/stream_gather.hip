// Micro-benchmark (gfx950): can the L2 stand in for k_anchor's LDS window?
// Every wave streams its own segment of a large buffer in 2 KiB blocks (8 coalesced dword loads per lane-row, next block prefetched),
// runs `alu` x 8 cheap vector instructions per block, and — the question — fetches the context of "anchors" found 1..AGE blocks ago
// either from a per-wave LDS window (mode 0: 8 ds_write_b32 per block + ds_read gathers) or from global memory again
// (mode 1: G unaligned global_load_dwordx4 per block at random byte offsets inside the last AGE blocks, consumed one block later).
// Reports ms and GB/s per variant at several occupancies (dynamic LDS pads a workgroup to the wanted number per CU); run under
// rocprofv3 --pmc FETCH_SIZE to see whether the re-reads reach the fabric.
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_gather stream_gather.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr uint32_t BLK = 2048;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_u;   // byte-aligned 16-byte load: one global_load_dwordx4

template <int MODE, bool NT>
__global__ __launch_bounds__(256) void k_stream(const uint8_t* __restrict__ log, uint64_t len, uint32_t seg_bytes, uint32_t alu, uint32_t gathers,
                                                uint32_t age, uint32_t win_bytes, uint32_t* __restrict__ out) {
    extern __shared__ uint32_t lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * 4 + wave;
    const uint64_t seg_start = (uint64_t)gw * seg_bytes;
    if (seg_start >= len) return;
    const uint64_t seg_end = seg_start + seg_bytes < len ? seg_start + seg_bytes : len;
    uint32_t* win = lds + wave * (win_bytes / 4);
    uint32_t acc[8] = {lane, 1, 2, 3, 4, 5, 6, 7};
    uint32_t nx[8];
    const uint32_t* src = reinterpret_cast<const uint32_t*>(log + seg_start) + lane;
#pragma unroll
    for (int q = 0; q < 8; ++q) nx[q] = NT ? __builtin_nontemporal_load(src + 64 * q) : src[64 * q];
    uint32_t rnd = lane * 2654435761u + gw;
    u32x4 pend[4] = {};
    for (uint64_t blk = seg_start; blk < seg_end; blk += BLK) {
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(nx[q]));
        // consume the gathers of the previous block
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] ^= pend[g].x + pend[g].y + pend[g].z + pend[g].w;
        uint32_t w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) w[q] = nx[q];
        if (blk + BLK < seg_end) {
            const uint32_t* s2 = reinterpret_cast<const uint32_t*>(log + blk + BLK) + lane;
#pragma unroll
            for (int q = 0; q < 8; ++q) nx[q] = NT ? __builtin_nontemporal_load(s2 + 64 * q) : s2[64 * q];
        }
        if (MODE == 0) {
            uint32_t* dst = win + (((uint32_t)blk & (win_bytes - 1)) >> 2) + lane;
#pragma unroll
            for (int q = 0; q < 8; ++q) dst[64 * q] = w[q];
        }
        // "anchors": random byte offsets in the last `age` blocks
        const uint64_t span = (uint64_t)age * BLK;
        if (blk >= seg_start + span) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < (int)gathers) {
                    rnd = rnd * 1664525u + 1013904223u;
                    const uint32_t back = 16 + (rnd >> 8) % (uint32_t)(span - 32);
                    if (MODE == 1) {
                        const uint8_t* a = log + blk + BLK - 1 - back;
                        pend[g] = *reinterpret_cast<const u32x4_u*>(a);
                    } else {
                        const uint32_t off = ((uint32_t)(blk + BLK - 1 - back)) & (win_bytes - 1);
                        const uint32_t* q = win + (off >> 2);
                        const uint32_t sh = off & 3;
                        uint32_t t[5];
#pragma unroll
                        for (int i = 0; i < 5; ++i) t[i] = q[i];   // may run past the window end by 16 bytes: the allocation has a tail
                        pend[g].x = __builtin_amdgcn_alignbyte(t[1], t[0], sh); pend[g].y = __builtin_amdgcn_alignbyte(t[2], t[1], sh);
                        pend[g].z = __builtin_amdgcn_alignbyte(t[3], t[2], sh); pend[g].w = __builtin_amdgcn_alignbyte(t[4], t[3], sh);
                    }
                }
            }
        }
        // stand-in for the bit-sliced front end and the drains: alu x 8 cheap independent vector instructions
        for (uint32_t i = 0; i < alu; ++i) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = (acc[q] ^ w[q]) + 0x9E3779B9u;
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(acc[q]));
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) r ^= acc[q];
#pragma unroll
    for (int g = 0; g < 4; ++g) r ^= pend[g].x;
    if (r == 0x12345678u) out[gw] = r;
}

int main(int argc, char** argv) {
    const uint64_t len = argc > 1 ? strtoull(argv[1], nullptr, 0) : 1800000000ull;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint8_t* log = nullptr;
    uint32_t* out = nullptr;
    CHECK(hipMalloc(&log, len + 4096));
    CHECK(hipMalloc(&out, 1 << 20));
    {
        std::vector<uint32_t> h((len + 4096) / 4);
        uint32_t x = 12345;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
        CHECK(hipMemcpy(log, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct Var { const char* name; int mode; bool nt; uint32_t gathers, age, wgs_per_cu, alu; };
    std::vector<Var> vars;
    for (uint32_t alu : {0u, 24u, 40u}) {
        for (uint32_t wg : {4u, 5u, 6u, 8u}) {
            vars.push_back({"stream only        ", 2, true, 0, 3, wg, alu});
            vars.push_back({"L2 gather x1  nt   ", 1, true, 1, 3, wg, alu});
            vars.push_back({"L2 gather x2  nt   ", 1, true, 2, 3, wg, alu});
            vars.push_back({"L2 gather x2  plain", 1, false, 2, 3, wg, alu});
            vars.push_back({"L2 gather x2 age6 p", 1, false, 2, 6, wg, alu});
        }
        vars.push_back({"LDS window x1      ", 0, true, 1, 3, 4, alu});
        vars.push_back({"LDS window x2      ", 0, true, 2, 3, 4, alu});
    }
    for (const Var& v : vars) {
        const uint32_t nwaves = cus * v.wgs_per_cu * 4;
        uint32_t seg = (uint32_t)((len / nwaves) / BLK) * BLK;
        const uint32_t win_bytes = 8192;
        // dynamic LDS pads the workgroup so that exactly wgs_per_cu fit a CU (160 KiB)
        const uint32_t lds_bytes = v.mode == 0 ? 4 * win_bytes + 64 : 163840 / v.wgs_per_cu - 512;
        const uint32_t lds_use = v.mode == 0 ? (163840 / v.wgs_per_cu - 512 > lds_bytes ? 163840 / v.wgs_per_cu - 512 : lds_bytes) : lds_bytes;
        auto launch = [&]() {
            const dim3 g(cus * v.wgs_per_cu), b(256);
#define L(M, N) hipLaunchKernelGGL((k_stream<M, N>), g, b, lds_use, 0, log, (uint64_t)seg * nwaves, seg, v.alu, v.gathers, v.age, win_bytes, out)
            if (v.mode == 0) L(0, true);
            else if (v.mode == 2) L(2, true);
            else if (v.nt) L(1, true);
            else L(1, false);
#undef L
        };
        if (v.mode == 0) CHECK(hipFuncSetAttribute((const void*)k_stream<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        CHECK(hipFuncSetAttribute((const void*)k_stream<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        CHECK(hipFuncSetAttribute((const void*)k_stream<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        CHECK(hipFuncSetAttribute((const void*)k_stream<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int reps = 5;
        for (int r = 0; r < reps; ++r) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double bytes = (double)seg * nwaves;
        printf("alu=%2u x8  wg/CU=%u  %s  %.3f ms  %.0f GB/s\n", v.alu, v.wgs_per_cu, v.name, ms, bytes / ms * 1e-6);
        fflush(stdout);
    }
    return 0;
}
