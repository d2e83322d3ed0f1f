// valu_microbench.hip — measures the fp32 VALU issue rates of gfx950 that the sphere kernel's design
// rests on: plain v_fma_f32 / v_mul+v_add, packed v_pk_fma_f32 / v_pk_mul+v_pk_add, with VGPR and
// SGPR operands, at 1..8 waves per SIMD.  Prints TFLOP/s and cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;     // independent chains

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float a, float b) {
    float x[UNROLL];
    f2 p[UNROLL];
    for (int i = 0; i < UNROLL; i++) { x[i] = threadIdx.x * 1e-6f + i; p[i] = f2{ x[i], x[i] + 0.5f }; }
    const f2 pa = f2{ a, a }, pb = f2{ b, b };
    const float bv = b + threadIdx.x * 1e-9f;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            // inline asm so the compiler can neither SLP-pack the plain forms nor unpack the packed ones
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(bv));
            if (MODE == 1) asm volatile("v_mul_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0" : "+v"(x[i]) : "s"(a), "v"(bv));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
            if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
            if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2" : "+v"(p[i]) : "s"(pa), "v"(pb));
            if (MODE == 5) asm volatile("v_sub_f32 %0, %0, %1\n\tv_alignbit_b32 %0, %0, %2, 31" : "+v"(x[i]) : "s"(a), "v"(bv));
        }
    }
    float s = 0;
    for (int i = 0; i < UNROLL; i++) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd, double flops_per_inst_lane, int insts_per_iter) {
    const int blocks = 256 * waves_per_simd;      // 256-thread blocks = 4 waves = 1 per SIMD
    float* out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.0001f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.0001f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_insts = (double)blocks * 4 * ITERS * UNROLL * insts_per_iter;
    const double flops = wave_insts * 64 * flops_per_inst_lane;
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal
    const double cyc = ms * 1e-3 * 2.4e9 / (wave_insts / (256.0 * 4));
    printf("%-22s waves/SIMD=%d  %8.3f ms  %7.2f TFLOP/s  %5.2f cyc/wave-inst/SIMD(@2.4GHz)\n", name, waves_per_simd, ms, flops / ms * 1e-9, cyc);
    CHECK(hipFree(out));
}

int main() {
    for (int w : { 1, 2, 4, 8 }) {
        run<0>("v_fma_f32", w, 2, 1);
        run<1>("v_mul_f32+v_add_f32", w, 1, 2);
        run<2>("v_pk_fma_f32", w, 4, 1);
        run<3>("v_pk_mul+v_pk_add", w, 2, 2);
        run<4>("v_pk_mul(sgpr)+v_pk_add", w, 2, 2);
        run<5>("v_sub+v_alignbit", w, 1, 2);
    }
    return 0;
}
