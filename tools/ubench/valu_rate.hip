// Micro-benchmark (gfx950): issue rate of the integer VALU / DPP / SDWA / cross-lane ops k_anchor is built from.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
// Every kernel runs `iters` x 64 independent instances of one instruction per wave; 1, 2 or 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define OPS(X) \
  X(0, "v_and_b32", "v_and_b32 %0, %1, %2", 2) \
  X(1, "v_or_b32", "v_or_b32 %0, %1, %2", 2) \
  X(2, "v_xor_b32", "v_xor_b32 %0, %1, %2", 2) \
  X(3, "v_xnor_b32", "v_xnor_b32 %0, %1, %2", 2) \
  X(4, "v_not_b32", "v_not_b32 %0, %1", 1) \
  X(5, "v_mov_b32", "v_mov_b32 %0, %1", 1) \
  X(6, "v_add_u32", "v_add_u32 %0, %1, %2", 2) \
  X(7, "v_sub_u32", "v_sub_u32 %0, %1, %2", 2) \
  X(8, "v_lshlrev_b32 imm", "v_lshlrev_b32 %0, 4, %1", 1) \
  X(9, "v_lshrrev_b32 imm", "v_lshrrev_b32 %0, 4, %1", 1) \
  X(10, "v_lshlrev_b32 vv", "v_lshlrev_b32 %0, %1, %2", 2) \
  X(11, "v_and_b32 literal", "v_and_b32 %0, 0x55555555, %1", 1) \
  X(12, "v_and_b32 sgpr", "v_and_b32 %0, s10, %1", 1) \
  X(13, "v_min_u32", "v_min_u32 %0, %1, %2", 2) \
  X(14, "v_mul_u32_u24", "v_mul_u32_u24 %0, %1, %2", 2) \
  X(15, "v_cndmask_b32 vcc", "v_cndmask_b32 %0, %1, %2, vcc", 2) \
  X(16, "v_bfi_b32", "v_bfi_b32 %0, %1, %2, %3", 3) \
  X(17, "v_bfi_b32 sgpr mask", "v_bfi_b32 %0, s10, %1, %2", 2) \
  X(18, "v_and_or_b32", "v_and_or_b32 %0, %1, %2, %3", 3) \
  X(19, "v_or3_b32", "v_or3_b32 %0, %1, %2, %3", 3) \
  X(20, "v_lshl_or_b32", "v_lshl_or_b32 %0, %1, 8, %2", 2) \
  X(21, "v_lshl_add_u32", "v_lshl_add_u32 %0, %1, 1, %2", 2) \
  X(22, "v_add3_u32", "v_add3_u32 %0, %1, %2, %3", 3) \
  X(23, "v_alignbyte_b32", "v_alignbyte_b32 %0, %1, %2, 3", 2) \
  X(24, "v_alignbit_b32", "v_alignbit_b32 %0, %1, %2, 7", 2) \
  X(25, "v_perm_b32", "v_perm_b32 %0, %1, %2, %3", 3) \
  X(26, "v_bfe_u32", "v_bfe_u32 %0, %1, 8, 8", 1) \
  X(27, "v_bcnt_u32_b32", "v_bcnt_u32_b32 %0, %1, %2", 2) \
  X(28, "v_ffbl_b32", "v_ffbl_b32 %0, %1", 1) \
  X(29, "v_mbcnt_lo", "v_mbcnt_lo_u32_b32 %0, %1, %2", 2) \
  X(30, "v_mad_u32_u24", "v_mad_u32_u24 %0, %1, %2, %3", 3) \
  X(31, "v_mul_lo_u32", "v_mul_lo_u32 %0, %1, %2", 2) \
  X(32, "v_sad_u8", "v_sad_u8 %0, %1, %2, %3", 3) \
  X(33, "v_dot4_u32_u8", "v_dot4_u32_u8 %0, %1, %2, %3", 3) \
  X(34, "v_pk_add_u16", "v_pk_add_u16 %0, %1, %2", 2) \
  X(35, "v_mov_b32_dpp wave_shr1", "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf", 1) \
  X(36, "v_and_b32_dpp wave_shr1", "v_and_b32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf", 2) \
  X(37, "v_mov_b32_dpp quad_perm", "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 1) \
  X(38, "v_mov_b32_dpp row_shr1", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", 1) \
  X(39, "v_or_b32_sdwa bytes", "v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2", 2) \
  X(40, "v_mov_b32_sdwa byte", "v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2", 1) \
  X(41, "v_lshlrev_b32_sdwa", "v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1", 2) \
  X(42, "v_and_b32_sdwa dst byte", "v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0", 12) \
  X(43, "v_fma_f32", "v_fma_f32 %0, %1, %2, %3", 3) \
  X(44, "v_cmp_ne_u32 vcc", "v_cmp_ne_u32 vcc, %1, %2", 20) \
  X(45, "v_cmp_ne_u32 sgpr", "v_cmp_ne_u32 s[12:13], %1, %2", 20) \
  X(46, "v_readlane_b32", "v_readlane_b32 s12, %1, 63", 20) \
  X(47, "v_readfirstlane", "v_readfirstlane_b32 s12, %1", 20) \
  X(48, "v_writelane_b32", "v_writelane_b32 %0, s10, 5", 10) \
  X(49, "ds_bpermute_b32", "ds_bpermute_b32 %0, %1, %2", 32) \
  X(50, "ds_read_u8", "ds_read_u8 %0, %1", 31) \
  X(51, "ds_read_b32", "ds_read_b32 %0, %1", 31) \
  X(52, "ds_write_b32", "ds_write_b32 %1, %2", 30) \
  X(53, "v_xad_u32", "v_xad_u32 %0, %1, %2, %3", 3) \
  X(54, "v_sub_u32 imm", "v_sub_u32 %0, 48, %1", 1) \
  X(55, "v_ashrrev_i32", "v_ashrrev_i32 %0, 31, %1", 1) \
  X(56, "v_or_b32 literal", "v_or_b32 %0, 0x80808080, %1", 1) \
  X(57, "v_add_u32 self (shl1)", "v_add_u32 %0, %1, %1", 1) \
  X(58, "v_pk_lshlrev_b16", "v_pk_lshlrev_b16 %0, 1, %1", 1) \
  X(59, "v_cmp_lt + v_cndmask", "v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc", 2)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t lds[1024];
    lds[threadIdx.x] = seed; lds[threadIdx.x + 256] = seed; lds[threadIdx.x + 512] = seed; lds[threadIdx.x + 768] = seed;
    __syncthreads();
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = (seed * (i + 1) + threadIdx.x) & 0x3FC;
    asm volatile("s_mov_b32 s10, 0x33333333" ::: "s10");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                uint32_t a = r[i], b = r[(i + 5) & 15], c = r[(i + 11) & 15];
#define X(N, NAME, ASM, KIND) \
                if constexpr (OP == N) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b), "v"(c) : "vcc", "s12", "s13", "memory");
                OPS(X)
#undef X
            }
        }
        if (OP >= 49 && OP <= 52) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t x = lds[threadIdx.x];
#pragma unroll
    for (int i = 0; i < 16; ++i) x ^= r[i];
    if (x == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <int OP>
int run(const char* name, uint32_t* d, int cus) {
    const int iters = 1000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double cyc[3];
    int wi = 0;
    for (int w : {1, 2, 4}) {
        hipLaunchKernelGGL(k<OP>, dim3(cus * w), dim3(256), 0, 0, d, 10, 1u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(cus * w), dim3(256), 0, 0, d, iters, 1u);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = (double)w * iters * 64;
        cyc[wi++] = ms * 1e6 / instr_per_simd;   // ns per wave-instruction per SIMD
    }
    printf("%-28s ns/instr/SIMD at 1,2,4 waves/SIMD: %6.3f %6.3f %6.3f   (cycles @2.1GHz: %5.2f %5.2f %5.2f)\n", name, cyc[0], cyc[1], cyc[2],
           cyc[0] * 2.1, cyc[1] * 2.1, cyc[2] * 2.1);
    return 0;
}

int main() {
    uint32_t* d; CHECK(hipMalloc(&d, 1 << 26));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
#define X(N, NAME, ASM, KIND) run<N>(NAME, d, cus);
    OPS(X)
#undef X
    return 0;
}
