// tools/mb_occupancy.hip — how many workgroups of W waves does a CU of the MI355X hold at a given VGPR count and LDS size?  (round 4: the lean sphere kernel
// fits 96 VGPRs, i.e. five waves per SIMD - as two 10-wave workgroups per CU?)  Every workgroup spins for a fixed 200 us of wall time; a grid of 2 x CUs
// workgroups then takes 200 us if two are resident per CU and 400 us if one is.  Per-SIMD wave counts from HW_ID.
//   hipcc --offload-arch=gfx950 -O2 tools/mb_occupancy.hip -o build/mb/mb_occupancy && build/mb/mb_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int VGPRS>
__global__ void __launch_bounds__(1024) k_spin(unsigned long long* out, int spin_100mhz_ticks) {
    extern __shared__ unsigned char smem[];
    if (VGPRS == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if (VGPRS == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    if (VGPRS == 104) asm volatile("v_mov_b32 v103, 0" ::: "v103");
    if (VGPRS == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    smem[threadIdx.x] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_100mhz_ticks) __builtin_amdgcn_s_sleep(16);
    if ((threadIdx.x & 63) == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        unsigned long long* w = out + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4;
        w[0] = t0; w[1] = __builtin_amdgcn_s_memrealtime(); w[2] = hw; w[3] = xcc;
    }
}

template <int VGPRS>
static void run(int waves, int lds, int cus) {
    const int grid = 2 * cus + cus / 2, ticks = 20000;          // 200 us each; 2.5 workgroups per CU
    unsigned long long* d;
    const size_t n = (size_t)grid * waves * 4;
    CK(hipMalloc(&d, n * 8));
    CK(hipMemset(d, 0, n * 8));
    CK(hipFuncSetAttribute((const void*)k_spin<VGPRS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_spin<VGPRS>, dim3(grid), dim3(64 * waves), lds, 0, d, ticks);
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(n);
    CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    // waves that started within the first 50 us = the first residency wave; count them per SIMD
    unsigned long long tmin = ~0ull;
    for (size_t k = 0; k < n; k += 4) if (h[k] && h[k] < tmin) tmin = h[k];
    int first = 0, per_simd_max = 0;
    std::vector<int> cnt(8 * 64 * 4 * 16, 0);
    for (size_t k = 0; k < n; k += 4) {
        if (h[k] - tmin > 5000) continue;
        first++;
        const unsigned hw = (unsigned)h[k + 2], xcc = (unsigned)h[k + 3] & 0xF;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        int& c = cnt[(((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd];
        c++; if (c > per_simd_max) per_simd_max = c;
    }
    printf("VGPRs %3d  workgroup %2d waves  LDS %6d B : kernel %.3f ms (200 us per residency round; %d workgroups on %d CUs) first round: %d waves = %.2f workgroups per CU, max %d waves on one SIMD\n",
           VGPRS, waves, lds, ms, grid, cus, first, (double)first / waves / cus, per_simd_max);
    CK(hipFree(d));
}

int main() {
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    run<96>(10, 72 * 1024, cus);
    run<96>(10, 32 * 1024, cus);
    run<96>(10, 1024, cus);
    run<96>(5, 36 * 1024, cus);
    run<96>(4, 30 * 1024, cus);
    run<96>(12, 60 * 1024, cus);
    run<96>(8, 60 * 1024, cus);
    run<80>(12, 70 * 1024, cus);
    run<80>(12, 1024, cus);
    run<80>(8, 50 * 1024, cus);
    run<104>(16, 100 * 1024, cus);
    run<128>(16, 100 * 1024, cus);
    run<128>(8, 70 * 1024, cus);
    return 0;
}
