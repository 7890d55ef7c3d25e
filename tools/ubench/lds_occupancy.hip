// How many 256-thread workgroups with N bytes of LDS does a CU of gfx950 hold? (hipOccupancy... and a timing check: 5 x CUs workgroups
// that each spin 200 us finish in one spin period only if all are resident at once.)  hipcc --offload-arch=gfx950 -O2 lds_occupancy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES>
__global__ __launch_bounds__(256) void spin(uint32_t* out, long long ticks) {
    __shared__ uint32_t s[BYTES / 4];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0) out[blockIdx.x] = s[(blockIdx.x * 7) % (BYTES / 4)];
}
template <int BYTES>
void probe(uint32_t* out, int cus) {
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin<BYTES>, 256, 0);
    for (int per = 4; per <= 6; ++per) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        spin<BYTES><<<cus * per, 256>>>(out, 20000);   // 100 MHz clock: 200 us
        hipDeviceSynchronize();
        hipEventRecord(a);
        spin<BYTES><<<cus * per, 256>>>(out, 20000);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        printf("lds %6d B: api %d wg/cu; %d wg/cu launched -> %.3f ms\n", BYTES, nb, per, ms);
    }
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    uint32_t* out; hipMalloc(&out, 1 << 20);
    printf("CUs %d, LDS per CU (maxSharedMemoryPerMultiProcessor) %zu\n", pr.multiProcessorCount, (size_t)pr.maxSharedMemoryPerMultiProcessor);
    probe<32768>(out, pr.multiProcessorCount);
    probe<32256>(out, pr.multiProcessorCount);
    probe<31744>(out, pr.multiProcessorCount);
    probe<30720>(out, pr.multiProcessorCount);
    probe<27136>(out, pr.multiProcessorCount);
    probe<26624>(out, pr.multiProcessorCount);
    return 0;
}
