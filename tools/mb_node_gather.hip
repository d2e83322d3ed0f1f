// tools/mb_node_gather.hip — what does ONE dependent node-record fetch per ray cost on gfx950, by how the lanes fetch it?
// Every wave runs `steps` dependent steps; each ACTIVE lane owns a chain: fetch a record at a data-dependent random
// index, derive the next index from the data.  Records are 64-byte aligned, 48 bytes used.
//   mode 0: the owner lane fetches its record itself: 3 x global_load_dwordx4 (what k_render_mesh_queue does)
//   mode 1: the owner's QUAD fetches it: round r serves owner lane 4q+r, lane 4q+j loads piece j with ONE
//           global_load_dwordx4 (4 lanes -> one 64-byte line); data handed to the owner with quad DPP
//   mode 2: as 1, but the pieces are written straight into LDS (global_load_lds_dwordx4) and the owner reads its
//           48 bytes back with 3 x ds_read_b128
//   mode 3: the owner fetches ONE dwordx4 only (cost per lane-instruction)
// Printed: cycles per step per wave (wall time x 2.4 GHz / steps) and records per cycle per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/mb_node_gather.hip -o build/exp/mb_node_gather && build/exp/mb_node_gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int CTRL> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t quad_bcast_r(uint32_t v, int r) {
    return r == 0 ? quad_bcast<0x00>(v) : r == 1 ? quad_bcast<0x55>(v) : r == 2 ? quad_bcast<0xAA>(v) : quad_bcast<0xFF>(v);
}

template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const float4* __restrict__ tab, uint32_t nrec_mask, int steps, int active, float* out) {
    extern __shared__ float4 s_dyn[];                  // occupancy limiter + (mode 2) landing area: 4 KB per wave at the start
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, j = lane & 3u;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    float acc = 0.0f;
    const bool mine = (int)lane < active;
    float4* land = s_dyn + wave * 256;                 // [4 rounds][64 lanes] float4
    const uint32_t land_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4*)land;
    for (int s = 0; s < steps; s++) {
        float y = 0.0f;
        if (MODE == 0 || MODE == 3) {
            if (mine) {
                const float4* n = tab + (size_t)(idx & nrec_mask) * 4;
                if (MODE == 0) { const float4 a = n[0], b = n[1], c = n[2]; y = a.x + b.y + c.z; }
                else { const float4 a = n[0]; y = a.x + a.y; }
            }
        } else if (MODE == 1) {
            float4 v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t oidx = quad_bcast_r(idx, r);
                const bool oact = (int)((lane & ~3u) + r) < active;
                v[r] = make_float4(0, 0, 0, 0);
                if (oact && j != 3u) v[r] = tab[(size_t)(oidx & nrec_mask) * 4 + j];
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t x = __float_as_uint(v[r].x + v[r].y);
                const float p = __uint_as_float(quad_bcast<0x00>(x)) + __uint_as_float(quad_bcast<0x55>(x)) + __uint_as_float(quad_bcast<0xAA>(x));
                if (j == (uint32_t)r) y = p;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t oidx = quad_bcast_r(idx, r);
                const bool oact = (int)((lane & ~3u) + r) < active;
                if (oact && j != 3u) {
                    const float4* src = tab + (size_t)(oidx & nrec_mask) * 4 + j;
                    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(land_addr + (uint32_t)r * 1024u));
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (mine) {
                const float4* rec = land + j * 64u + (lane & ~3u);      // round j, quad's four slots
                const float4 a = rec[0], b = rec[1], c = rec[2];
                y = a.x + b.y + c.z;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (mine) {
            acc += y;
            idx = idx * 1664525u + 1013904223u + (__float_as_uint(y) & 0xFFFFu);
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main() {
    const int cus = 256;
    float4* tab; float* out;
    const uint32_t max_rec = 1u << 16;                 // 4 MB
    if (hipMalloc(&tab, (size_t)max_rec * 64) != hipSuccess) return 1;
    std::vector<float> h((size_t)max_rec * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(rand() & 0xFFFF) * 1e-3f;
    if (hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
    if (hipMalloc(&out, (size_t)cus * 8 * 256 * 4) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int steps = 20000;
    printf("mode table_KB waves/CU active | cycles per step per wave | records per cycle per CU\n");
    for (int mode = 0; mode < 4; mode++)
        for (uint32_t nrec : { 1u << 14 })
            for (int wg_per_cu : { 2, 4 })                  // 8, 16 waves per CU
                for (int active : { 16, 32, 48, 64 }) {
                    const size_t lds = (size_t)(150 * 1024 / wg_per_cu) & ~255ull;     // occupancy is limited through dynamic LDS
                    auto kern = mode == 0 ? k_gather<0> : mode == 1 ? k_gather<1> : mode == 2 ? k_gather<2> : k_gather<3>;
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    for (int rep = 0; rep < 2; rep++) {
                        (void)hipEventRecord(e0);
                        hipLaunchKernelGGL(kern, dim3(cus * wg_per_cu), dim3(256), lds, 0, tab, nrec - 1, steps, active, out);
                        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    }
                    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                    const double cyc = ms * 1e-3 * 2.4e9 / steps;
                    printf("%4d %8u %8d %6d | %10.0f | %6.2f\n", mode, nrec * 64 / 1024, wg_per_cu * 4, active, cyc, wg_per_cu * 4 * active / cyc);
                    fflush(stdout);
                }
    // self-check of the LDS-DMA form: the same chains as mode 0 must give the same sums
    {
        const size_t n = (size_t)cus * 2 * 256, lds = 64 * 1024;
        std::vector<float> r0(n), r2(n);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gather<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gather<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_gather<0>, dim3(cus * 2), dim3(256), lds, 0, tab, (1u << 14) - 1, 7, 37, out);
        (void)hipMemcpy(r0.data(), out, n * 4, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_gather<2>, dim3(cus * 2), dim3(256), lds, 0, tab, (1u << 14) - 1, 7, 37, out);
        (void)hipMemcpy(r2.data(), out, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0, nz = 0;
        for (size_t i = 0; i < n; i++) { bad += r0[i] != r2[i]; nz += r0[i] != 0.0f; }
        printf("self-check mode 2 vs mode 0: %zu of %zu differ (%zu non-zero)\n", bad, n, nz);
    }
    return 0;
}
