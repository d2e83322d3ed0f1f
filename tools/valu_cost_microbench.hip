// tools/valu_cost_microbench.hip — issue cost (cycles per wave-instruction per SIMD) of the VALU instruction FORMS the two render
// kernels are made of, on gfx950 at 4 waves per SIMD (their occupancy): 16 independent chains per lane, inline asm.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_cost_microbench.hip -o build/exp/valu_cost && build/exp/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 16384;
constexpr int UNROLL = 16;

#define FORMS(X) \
    X(0,  "v_add_f32 v,v,v (VOP2)",            "v_add_f32 %0, %0, %3", 1, 0) \
    X(1,  "v_mul_f32 v,s,v (VOP2)",            "v_mul_f32 %0, %4, %0", 1, 0) \
    X(2,  "v_fma_f32 v,v,v,v",                 "v_fma_f32 %0, %0, %3, %2", 1, 0) \
    X(3,  "v_fma_f32 v,v,s,v",                 "v_fma_f32 %0, %0, %4, %3", 1, 0) \
    X(4,  "v_fmac_f32 v,v,v (VOP2)",           "v_fmac_f32 %0, %2, %3", 1, 0) \
    X(5,  "v_max_f32 v,v,v (VOP2)",            "v_max_f32 %0, %0, %3", 1, 0) \
    X(6,  "v_max3_f32 v,v,v,v",                "v_max3_f32 %0, %0, %2, %3", 1, 0) \
    X(7,  "v_med3_f32 v,v,v,v",                "v_med3_f32 %0, %0, %2, %3", 1, 0) \
    X(8,  "v_cndmask_b32_e32 (vcc)",           "v_cndmask_b32_e32 %0, %0, %3, vcc", 1, 1) \
    X(9,  "v_cndmask_b32_e64 (sgpr pair)",     "v_cndmask_b32_e64 %0, %0, %3, %5", 1, 0) \
    X(10, "v_cmp_lt_f32_e32 (-> vcc)",         "v_cmp_lt_f32_e32 vcc, %0, %3", 1, 1) \
    X(11, "v_cmp_lt_f32_e64 (-> sgpr pair)",   "v_cmp_lt_f32_e64 s[40:41], %0, %3", 1, 0) \
    X(12, "v_cmp + v_cndmask vcc pair",        "v_cmp_lt_f32_e32 vcc, %0, %3\n\tv_cndmask_b32_e32 %0, %0, %2, vcc", 2, 1) \
    X(13, "v_add_u32 (VOP2)",                  "v_add_u32_e32 %0, %0, %3", 1, 0) \
    X(14, "v_lshl_add_u32",                    "v_lshl_add_u32 %0, %0, 1, %3", 1, 0) \
    X(15, "v_mul_u32_u24 (VOP2)",              "v_mul_u32_u24_e32 %0, %0, %3", 1, 0) \
    X(16, "v_mul_lo_u32",                      "v_mul_lo_u32 %0, %0, %3", 1, 0) \
    X(17, "v_mad_u64_u32",                     "v_mad_u64_u32 %1, vcc, %0, %3, %1", 1, 1) \
    X(18, "v_lshl_add_u64",                    "v_lshl_add_u64 %1, %1, 0, %6", 1, 0) \
    X(19, "v_xor_b32 (VOP2)",                  "v_xor_b32_e32 %0, %0, %3", 1, 0) \
    X(20, "v_lshlrev_b32 (VOP2)",              "v_lshlrev_b32_e32 %0, 13, %0", 1, 0) \
    X(21, "v_alignbit_b32",                    "v_alignbit_b32 %0, %0, %3, 31", 1, 0) \
    X(22, "v_mov_b32 (VOP1)",                  "v_mov_b32_e32 %0, %3", 1, 0) \
    X(23, "v_mov_b32 dpp quad_perm",           "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 1, 0) \
    X(24, "v_add_f32 dpp row_shr:1",           "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", 1, 0) \
    X(25, "v_readlane_b32",                    "v_readlane_b32 s40, %0, 3", 1, 2) \
    X(26, "v_writelane_b32",                   "v_writelane_b32 %0, s42, 3", 1, 2) \
    X(27, "v_rcp_f32",                         "v_rcp_f32_e32 %0, %0", 1, 0) \
    X(28, "v_sqrt_f32",                        "v_sqrt_f32_e32 %0, %0", 1, 0) \
    X(29, "v_div_scale_f32",                   "v_div_scale_f32 %0, vcc, %0, %3, %0", 1, 1) \
    X(30, "v_div_fmas_f32",                    "v_div_fmas_f32 %0, %0, %2, %3", 1, 0) \
    X(31, "v_div_fixup_f32",                   "v_div_fixup_f32 %0, %0, %2, %3", 1, 0) \
    X(32, "v_fma_f64",                         "v_fma_f64 %1, %1, %6, %6", 1, 0) \
    X(33, "v_mul_f64",                         "v_mul_f64 %1, %1, %6", 1, 0) \
    X(34, "v_add_f64",                         "v_add_f64 %1, %1, %6", 1, 0) \
    X(35, "v_cvt_f32_u32",                     "v_cvt_f32_u32_e32 %0, %0", 1, 0) \
    X(36, "v_cvt_f64_f32",                     "v_cvt_f64_f32_e32 %1, %0", 1, 0) \
    X(37, "v_cvt_f32_f64",                     "v_cvt_f32_f64_e32 %0, %1", 1, 0) \
    X(38, "v_mbcnt_lo + v_mbcnt_hi",           "v_mbcnt_lo_u32_b32 %0, s40, 0\n\tv_mbcnt_hi_u32_b32 %0, s41, %0", 2, 2) \
    X(39, "v_ffbl_b32",                        "v_ffbl_b32_e32 %0, %0", 1, 0) \
    X(40, "v_sub_f32 v,v,v + v_mul_f32 v,v,v", "v_sub_f32 %0, %0, %3\n\tv_mul_f32 %0, %2, %0", 2, 0) \
    X(41, "v_min3_f32 v,v,v,v",                "v_min3_f32 %0, %0, %2, %3", 1, 0) \
    X(42, "v_exp_f32",                         "v_exp_f32_e32 %0, %0", 1, 0) \
    X(43, "v_sin_f32",                         "v_sin_f32_e32 %0, %0", 1, 0) \
    X(44, "v_pk_mul_f32",                      "v_pk_mul_f32 %1, %1, %6", 1, 0) \
    X(45, "s_nop 0 (issue slot)",              "s_nop 0", 1, 0) \
    X(46, "v_and_or_b32",                      "v_and_or_b32 %0, %0, %2, %3", 1, 0) \
    X(47, "v_bfe_u32",                         "v_bfe_u32 %0, %0, 3, 8", 1, 0)

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, unsigned long long m, unsigned long long* cyc, int lanes) {
    if ((int)(threadIdx.x & 63) >= lanes) return;
    float x[UNROLL];
    double d[UNROLL];
    for (int i = 0; i < UNROLL; i++) { x[i] = threadIdx.x * 1e-6f + i + 1.0f; d[i] = x[i]; }
    const float bv = 1.0001f + threadIdx.x * 1e-9f, cv = 0.5f + threadIdx.x * 1e-9f;
    const double dv = 1.0000001 + threadIdx.x * 1e-12;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
#define X(ID, NAME, ASM, N, C) if (MODE == ID) { \
                if (C == 0) asm volatile(ASM : "+v"(x[i]), "+v"(d[i]) : "v"(bv), "v"(cv), "s"(a), "s"(m), "v"(dv)); \
                if (C == 1) asm volatile(ASM : "+v"(x[i]), "+v"(d[i]) : "v"(bv), "v"(cv), "s"(a), "s"(m), "v"(dv) : "vcc"); \
                if (C == 2) asm volatile(ASM : "+v"(x[i]), "+v"(d[i]) : "v"(bv), "v"(cv), "s"(a), "s"(m), "v"(dv) : "vcc", "s40", "s41", "s42"); }
            FORMS(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < UNROLL; i++) s += x[i] + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int n_inst, int waves_per_simd, int cus = 256, int lanes = 64) {
    const int blocks = cus * waves_per_simd;
    float* out;
    unsigned long long* dcyc; unsigned long long hcyc = 0;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMalloc(&dcyc, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0x5555aaaa5555aaaaull, dcyc, lanes);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0x5555aaaa5555aaaaull, dcyc, lanes);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_insts = (double)blocks * 4 * ITERS * UNROLL * n_inst;
    const double cyc = ms * 1e-3 * 2.4e9 / (wave_insts / (cus * 4.0));
    CHECK(hipMemcpy(&hcyc, dcyc, 8, hipMemcpyDeviceToHost));
    const double clk = (double)hcyc / ((double)ITERS * UNROLL * n_inst * waves_per_simd);
    printf("%-36s CUs %3d lanes %2d waves/SIMD=%d  %5.2f cycles per wave-instruction per SIMD at 2.4 GHz wall | %5.2f by the wave's own s_memtime | %.0f MHz implied\n", name, cus, lanes, waves_per_simd, cyc, clk, (double)hcyc / (ms * 1e-3) * 1e-6);
    fflush(stdout);
    CHECK(hipFree(out));
}

int main() {
    for (int cus : { 256, 128, 64, 32, 8 }) for (int lanes : { 64, 32, 8 }) for (int w : { 4, 2, 1 }) {
        run<1>("v_mul_f32 v,s,v (VOP2)", 1, w, cus, lanes);
        run<19>("v_xor_b32 (VOP2)", 1, w, cus, lanes);
        run<40>("v_sub_f32 + v_mul_f32", 2, w, cus, lanes);
    }
    for (int w : { 4 }) {
#define X(ID, NAME, ASM, N, C) run<ID>(NAME, N, w);
        FORMS(X)
#undef X
    }
    return 0;
}
