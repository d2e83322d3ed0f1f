// rt_kernels_spheres.hip — the render() hot path for SPHERE scenes on gfx950 (wave64).
//
// Replaces /root/reference/kernels.cu:535-569 (render) + :396-533 (color) + :325-360 (hit) for the
// README-era sphere scene (README.md:84-104: scene arrays in __constant__, one thread per pixel,
// per-sample loop, brute-force sphere list).  This file is compiled twice by the Makefile:
//   -DRT_MODE_PARITY -ffp-contract=off   -> rt_launch_spheres_parity   (bit-exact vs the CPU oracle)
//   -DRT_MODE_FAST   -ffp-contract=fast  -> rt_launch_spheres_fast     (FMA; tolerance parity)
//
// MI355X design:
//   * one lane owns one pixel and its RNG stream (kernels.cu:541-542: one xorshift32 stream per
//     PIXEL running across all its samples, so a pixel's samples are inherently sequential);
//   * FLATTENED loop with in-lane refill: one loop iteration = one ray per lane; a lane whose path
//     ends starts its pixel's next sample in the same iteration, so lanes idle only at the very end
//     of their pixel instead of at every short path (the reference's warp efficiency was 41 %);
//   * the sphere array (cx,cy,cz,r*r) and the materials are staged once per workgroup into LDS
//     (instead of __constant__); the scan reads them with wave-uniform ds_read_b128 broadcasts;
//   * the closest-hit scan is split in two phases.  Phase 1 evaluates only the sign of the
//     discriminant for 32 spheres at a time (17 VALU ops/sphere, no branch, no compare: the sign
//     bit is shifted into a per-lane 32-bit mask with one v_alignbit).  Phase 2 walks the set bits
//     in index order and runs the full sphereHit (IEEE sqrt + divide) only for those candidates.
//     Because a non-candidate returns FLT_MAX in the reference and never updates `closest`, and
//     candidates are visited in increasing index with the same strict `<`, the result is
//     bit-identical to the reference's linear scan;
//   * framebuffer stores go through an LDS transpose so a wave writes row-contiguous dwords.
#include "rt_device.h"
#include "rt_params.h"

#include <float.h>

using namespace rtd;

#if defined(RT_MODE_PARITY)
#define RT_LAUNCH_NAME rt_launch_spheres_parity
#elif defined(RT_MODE_FAST)
#define RT_LAUNCH_NAME rt_launch_spheres_fast
#else
#error "define RT_MODE_PARITY or RT_MODE_FAST"
#endif

namespace {

constexpr int kWavesPerWg = 4;              // 4 waves side by side: a 32 x 8 pixel tile per workgroup
constexpr int kThreads = 64 * kWavesPerWg;

__device__ __forceinline__ int global_row(const RtPartition& pt, int lr) {
    const int stripe = lr / pt.stripe_rows;
    return (stripe * pt.world + pt.rank) * pt.stripe_rows + (lr - stripe * pt.stripe_rows);
}

// sphereHit, intersections.h:85-104, on a pre-normalised direction `dn` with a = dot(dn,dn) hoisted
// (same bits every call) and r2 = radius*radius precomputed (same bits).
__device__ __forceinline__ float sphere_hit_exact(float4 s, f3 org, f3 dn, float a, float t_min, float t_max) {
    const f3 oc = org - F3(s.x, s.y, s.z);
    const float b = dot(oc, dn);
    const float c = dot(oc, oc) - s.w;
    const float discriminant = b * b - a * c;
    if (discriminant > 0) {
        const float sq = rt_sqrt(discriminant);
        float temp = (-b - sq) / a;
        if (temp < t_max && temp > t_min) return temp;
        temp = (-b + sq) / a;
        if (temp < t_max && temp > t_min) return temp;
    }
    return FLT_MAX;
}

template <int VARIANT>
__global__ void __launch_bounds__(kThreads) k_render_spheres(const RtSphereParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    float4* s_sph = reinterpret_cast<float4*>(smem);                 // n_padded x (cx,cy,cz,r*r)
    float4* s_mat = s_sph + P.n_padded;                              // n x (r,g,b,param)
    float*  s_fb = reinterpret_cast<float*>(s_mat + P.n);            // kThreads x 3
    int*    s_typ = reinterpret_cast<int*>(s_fb + kThreads * 3);     // n

    for (int k = threadIdx.x; k < P.n_padded; k += kThreads) {
        float4 s = P.spheres[k];
        s.w = s.w * s.w;                                             // intersections.h:89 radius*radius
        s_sph[k] = s;
    }
    for (int k = threadIdx.x; k < P.n; k += kThreads) {
        s_mat[k] = P.mat_color[k];
        s_typ[k] = P.mat_type[k];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i0 = (blockIdx.x * kWavesPerWg + wave) * 8;            // left pixel column of this wave's 8x8 tile
    const int lr0 = blockIdx.y * 8;                                  // first local row of the tile
    const int i = i0 + (lane & 7);
    const int lr = lr0 + (lane >> 3);
    const bool valid = (i < P.nx) && (lr < P.part.local_rows);
    const int j = global_row(P.part, lr);
    const uint32_t pixelId = (uint32_t)(j * P.nx + i);               // kernels.cu:541 (global id -> seed)

    const int ngroups = P.n_padded >> 5;
    const float t_min = P.t_min;
    const int max_depth = P.max_depth;

    uint32_t rng = pixel_seed(pixelId);
    f3 col = F3(0, 0, 0);
    f3 org = F3(0, 0, 0), dir = F3(0, 0, 1), atten = F3(1, 1, 1), pcolor = F3(0, 0, 0);
    int bounce = 0;
    bool inside = false;
    int s = 0;
    uint32_t nrays = 0;

    // starts sample `s` of this lane's pixel: kernels.cu:549-555 + the head of color() :397-398
    auto start_sample = [&]() {
        if (P.rng_mode == RT_RNG_COUNTER) rng = sample_seed(pixelId, (uint32_t)s);
        const float u = ((float)i + rnd(rng)) / (float)P.nx;
        const float v = ((float)j + rnd(rng)) / (float)P.ny;
        f3 d;
        get_ray(P.cam, u, v, rng, org, d);
        dir = unit(d);                                               // ray.h:9 (get_ray returns a ray)
        atten = F3(1.0f, 1.0f, 1.0f);
        pcolor = F3(0, 0, 0);
        bounce = 0;
        inside = false;
    };

    bool active = valid && (P.ns > 0);
    if (active) start_sample();

    while (active) {
        // ---- hit(), kernels.cu:325-360: the ray is rebuilt from the path, which renormalises the direction
        const f3 dn = unit(dir);
        const float a = dot(dn, dn);
        float closest = FLT_MAX;
        int sid = -1;
        nrays++;

        for (int g = 0; g < ngroups; g++) {
            const float4* sp = s_sph + (g << 5);
            uint32_t mask = 0;
#pragma unroll
            for (int kk = 0; kk < 32; kk++) {
                const float4 sph = sp[kk];                           // wave-uniform address: LDS broadcast
                const float ocx = org.x - sph.x;
                const float ocy = org.y - sph.y;
                const float ocz = org.z - sph.z;
                const float b = ocx * dn.x + ocy * dn.y + ocz * dn.z;
                const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - sph.w;
                const float nd = a * c - b * b;                      // == -(b*b - a*c) bit for bit
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(nd), 31);   // mask = mask<<1 | sign(nd)
            }
            while (mask) {                                           // candidates, lowest sphere index first
                const int lz = __clz((int)mask);
                mask &= ~(0x80000000u >> lz);
                const int k = (g << 5) + lz;
                const float t = sphere_hit_exact(s_sph[k], org, dn, a, t_min, closest);
                if (k < P.n && t < closest) { closest = t; sid = k; }
            }
        }

        bool path_done;
        if (sid < 0) {
            pcolor = pcolor + atten * sky_color(P.sky, dir);         // kernels.cu:419-425
            path_done = true;
        } else {
            const float4 sc4 = s_sph[sid];
            const float radius = P.spheres[sid].w;
            const f3 hp = org + closest * dn;                        // ray.h:12 point_at_parameter
            f3 normal = (hp - F3(sc4.x, sc4.y, sc4.z)) / radius;     // intersections.h:95
            if (dot(dn, normal) > 0.0f) normal = -normal;            // kernels.cu:354-355
            const float4 m = s_mat[sid];
            Scatter sc;
            material_scatter(sc, closest, normal, inside, dir, s_typ[sid], F3(m.x, m.y, m.z), m.w, rng);
            org = org + sc.t * dir;                                  // kernels.cu:485-489
            dir = sc.wi;
            atten = atten * sc.throughput;
            inside = sc.refracted ? !inside : inside;
            path_done = false;
            if (P.rr && bounce > 3) {                                // kernels.cu:512-527
                const float mx = max3(atten);
                if (rnd(rng) > mx) {
                    path_done = true;
                } else {
                    const float kk = 1.0f / mx;
                    atten = F3(atten.x * kk, atten.y * kk, atten.z * kk);
                }
            }
            bounce++;
            if (bounce >= max_depth) path_done = true;               // loop bound, kernels.cu:402
        }

        if (path_done) {
            col = col + pcolor;                                      // kernels.cu:558
            s++;
            if (s < P.ns) start_sample();
            else active = false;
        }
    }

    // ---- framebuffer: fb[pixel] = col / ns (kernels.cu:568), transposed through LDS so that
    // consecutive lanes store consecutive dwords of a row segment (8 px * 12 B = 96 B per tile row)
    const f3 out = col / (float)P.ns;
    float* my = s_fb + threadIdx.x * 3;
    my[0] = out.x; my[1] = out.y; my[2] = out.z;
    __syncthreads();
    const float* wfb = s_fb + wave * 192;
    float* fbf = reinterpret_cast<float*>(P.fb);
#pragma unroll
    for (int q = lane; q < 192; q += 64) {
        const int row = q / 24, off = q - row * 24;
        const int px = i0 + off / 3;
        const int r = lr0 + row;
        if (px < P.nx && r < P.part.local_rows)
            fbf[((size_t)r * P.nx + i0) * 3 + off] = wfb[q];
    }

    if (P.counters) {
        atomicAdd(&P.counters->rays, (unsigned long long)nrays);
        atomicAdd(&P.counters->prim_tests, (unsigned long long)nrays * (unsigned long long)P.n);
    }
}

}  // namespace

#if defined(RT_MODE_PARITY)
size_t rt_sphere_kernel_lds_bytes(int n_padded, int n, int threads) {
    (void)threads;
    return (size_t)n_padded * 16 + (size_t)n * 16 + (size_t)kThreads * 3 * 4 + (size_t)n * 4;
}
#endif

hipError_t RT_LAUNCH_NAME(const RtSphereParams& p, int variant, hipStream_t stream) {
    (void)variant;
    const size_t lds = (size_t)p.n_padded * 16 + (size_t)p.n * 16 + (size_t)kThreads * 3 * 4 + (size_t)p.n * 4;
    auto kern = k_render_spheres<0>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const dim3 grid((p.nx + 8 * kWavesPerWg - 1) / (8 * kWavesPerWg), (p.part.local_rows + 7) / 8);
    hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, stream, p);
    return hipGetLastError();
}
