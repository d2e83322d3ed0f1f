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

// LDS image of the scene, staged once per workgroup (README.md:93-103 used __constant__).
struct SceneLds {
    const float4* sph;      // n_padded x (cx, cy, cz, r*r)
    const float4* mat;      // n x (r, g, b, param)
    const int*    typ;      // n
};

__device__ __forceinline__ SceneLds stage_scene(const RtSphereParams& P, unsigned char* smem, float** after) {
    float4* s_sph = reinterpret_cast<float4*>(smem);
    float4* s_mat = s_sph + P.n_padded;
    int*    s_typ = reinterpret_cast<int*>(s_mat + P.n);
    for (int k = threadIdx.x; k < P.n_padded; k += kThreads) {
        float4 s = P.spheres[k];
        s.w = s.w * s.w;                                             // intersections.h:89 radius*radius
        s_sph[k] = s;
    }
    for (int k = threadIdx.x; k < P.n; k += kThreads) {
        s_mat[k] = P.mat_color[k];
        s_typ[k] = P.mat_type[k];
    }
    *after = reinterpret_cast<float*>(s_typ + ((P.n + 3) & ~3));
    __syncthreads();
    return { s_sph, s_mat, s_typ };
}

// Per-lane path state (path, helper_structs.h:48-71, minus what sphere scenes never use).
struct Lane {
    uint32_t rng;
    f3 col;                 // pixel accumulator (kernels.cu:547)
    f3 org, dir, atten, pcolor;
    int bounce;
    bool inside;
    int s;                  // sample index within the pixel
    int i, j;               // global pixel coordinates
    uint32_t pixelId;
};

// kernels.cu:549-555 + the head of color() :397-398: starts sample L.s of the lane's pixel
__device__ __forceinline__ void start_sample(const RtSphereParams& P, Lane& L) {
    if (P.rng_mode == RT_RNG_COUNTER) L.rng = sample_seed(L.pixelId, (uint32_t)L.s);
    const float u = ((float)L.i + rnd(L.rng)) / (float)P.nx;
    const float v = ((float)L.j + rnd(L.rng)) / (float)P.ny;
    f3 d;
    get_ray(P.cam, u, v, L.rng, L.org, d);
    L.dir = unit(d);                                                 // ray.h:9 (get_ray returns a ray)
    L.atten = F3(1.0f, 1.0f, 1.0f);
    L.pcolor = F3(0, 0, 0);
    L.bounce = 0;
    L.inside = false;
}

__device__ __forceinline__ void start_pixel(const RtSphereParams& P, Lane& L, int i, int j) {
    L.i = i; L.j = j;
    L.pixelId = (uint32_t)(j * P.nx + i);                            // kernels.cu:541 (global id -> seed)
    L.rng = pixel_seed(L.pixelId);
    L.col = F3(0, 0, 0);
    L.s = 0;
    start_sample(P, L);
}

struct Hit { float closest; int sid; };

// ---- closest hit, LANE-PARALLEL form: every lane scans all spheres for its own ray --------------------------------
// `dn` is the renormalised direction (hit() rebuilds the ray: kernels.cu:326, ray.h:9), a = dot(dn,dn).
__device__ __forceinline__ Hit scan_lane_parallel(const RtSphereParams& P, const SceneLds& S, f3 org, f3 dn, float a) {
    const float t_min = P.t_min;
    const int ngroups = P.n_padded >> 5;
    Hit h = { FLT_MAX, -1 };
    for (int g = 0; g < ngroups; g++) {
        const float4* sp = S.sph + (g << 5);
        uint32_t mask = 0;
#pragma unroll
        for (int kk = 0; kk < 32; kk++) {
            const float4 sph = sp[kk];                               // wave-uniform address: LDS broadcast
            const float ocx = org.x - sph.x;
            const float ocy = org.y - sph.y;
            const float ocz = org.z - sph.z;
            const float b = ocx * dn.x + ocy * dn.y + ocz * dn.z;
            const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - sph.w;
            const float nd = a * c - b * b;                          // == -(b*b - a*c) bit for bit
            mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(nd), 31);   // mask = mask<<1 | sign(nd)
        }
        while (mask) {                                               // candidates, lowest sphere index first
            const int lz = __clz((int)mask);
            mask &= ~(0x80000000u >> lz);
            const int k = (g << 5) + lz;
            const float t = sphere_hit_exact(S.sph[k], org, dn, a, t_min, h.closest);
            if (k < P.n && t < h.closest) { h.closest = t; h.sid = k; }
        }
    }
    return h;
}

// ---- closest hit, WAVE-COOPERATIVE form: the 64 lanes share ONE ray (that of lane q) ---------------------------------
// Lane l tests spheres l, l+64, l+128, ...; the per-lane results are merged with the reference's tie rule.
// Why this is still the reference's answer: sphereHit's result for sphere k does not depend on the running
// `closest` except for acceptance (the far root is never below the near root), so the linear scan computes the
// lexicographic minimum of (t_k, k) over all spheres — which can be evaluated in any order.
// Must be called by all 64 lanes of the wave in uniform control flow; q is wave-uniform.
__device__ __forceinline__ Hit scan_cooperative(const RtSphereParams& P, const SceneLds& S, int q, f3 org, f3 dn, float a) {
    const int lane = threadIdx.x & 63;
    const f3 O = F3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(org.x), q)),
                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(org.y), q)),
                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(org.z), q)));
    const f3 D = F3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(dn.x), q)),
                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dn.y), q)),
                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dn.z), q)));
    const float A = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), q));
    const float t_min = P.t_min;
    const int rounds = P.n_padded >> 6;
    float bt = FLT_MAX;
    int bk = 0x7fffffff;
    for (int r0 = 0; r0 < rounds; r0 += 32) {
        const int rc = min(32, rounds - r0);
        uint32_t cm = 0;
        for (int r = 0; r < rc; r++) {
            const float4 sph = S.sph[((r0 + r) << 6) + lane];
            const float ocx = O.x - sph.x;
            const float ocy = O.y - sph.y;
            const float ocz = O.z - sph.z;
            const float b = ocx * D.x + ocy * D.y + ocz * D.z;
            const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - sph.w;
            const float nd = A * c - b * b;
            cm = __builtin_amdgcn_alignbit(cm, __float_as_uint(nd), 31);
        }
        cm <<= (32 - rc);                                            // round r0 at bit 31
        while (cm) {
            const int lz = __clz((int)cm);
            cm &= ~(0x80000000u >> lz);
            const int k = ((r0 + lz) << 6) + lane;
            const float t = sphere_hit_exact(S.sph[k], O, D, A, t_min, bt);
            if (k < P.n && t < bt) { bt = t; bk = k; }
        }
    }
    // merge: lexicographic minimum of (t, k) over the lanes that found something.  t > t_min >= 0, so the
    // IEEE bit patterns order like the values and the merge runs on the scalar unit.
    unsigned long long found = __ballot(bt < FLT_MAX);
    uint32_t st = __float_as_uint(FLT_MAX);
    int sk = -1;
    while (found) {
        const int b = __builtin_ctzll(found);
        found &= found - 1;
        const uint32_t tb = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(bt), b);
        const int kb = __builtin_amdgcn_readlane(bk, b);
        if (tb < st || (tb == st && kb < sk)) { st = tb; sk = kb; }
    }
    Hit h = { __uint_as_float(st), sk };
    return h;
}

// ---- shading of one hit / miss: the rest of color()'s loop body (kernels.cu:415-531) -------------------------------
// Returns true when the path ended (the caller accumulates L.pcolor and starts the next sample).
__device__ __forceinline__ bool shade(const RtSphereParams& P, const SceneLds& S, Lane& L, f3 dn, Hit h) {
    if (h.sid < 0) {
        L.pcolor = L.pcolor + L.atten * sky_color(P.sky, L.dir);     // kernels.cu:419-425
        return true;
    }
    const float4 sc4 = S.sph[h.sid];
    const float radius = P.spheres[h.sid].w;
    const f3 hp = L.org + h.closest * dn;                            // ray.h:12 point_at_parameter
    f3 normal = (hp - F3(sc4.x, sc4.y, sc4.z)) / radius;             // intersections.h:95
    if (dot(dn, normal) > 0.0f) normal = -normal;                    // kernels.cu:354-355
    const float4 m = S.mat[h.sid];
    Scatter sc;
    material_scatter(sc, h.closest, normal, L.inside, L.dir, S.typ[h.sid], F3(m.x, m.y, m.z), m.w, L.rng);
    L.org = L.org + sc.t * L.dir;                                    // kernels.cu:485-489
    L.dir = sc.wi;
    L.atten = L.atten * sc.throughput;
    L.inside = sc.refracted ? !L.inside : L.inside;
    bool path_done = false;
    if (P.rr && L.bounce > 3) {                                      // kernels.cu:512-527
        const float mx = max3(L.atten);
        if (rnd(L.rng) > mx) {
            path_done = true;
        } else {
            const float kk = 1.0f / mx;
            L.atten = F3(L.atten.x * kk, L.atten.y * kk, L.atten.z * kk);
        }
    }
    L.bounce++;
    if (L.bounce >= P.max_depth) path_done = true;                   // loop bound, kernels.cu:402
    return path_done;
}

// One iteration of color()'s bounce loop for every lane of the wave that has a ray (`has_ray`).  WAVE-LEVEL: all 64
// lanes must call it together.  With many live lanes each lane scans the sphere list for its own ray; with few
// (the tail of a tile / of the frame, where a handful of glass-trapped pixels need thousands of rays each) the
// whole wave works on one ray at a time, which cuts the latency of a ray ~30x and with it the critical path.
__device__ __forceinline__ bool trace_rays(const RtSphereParams& P, const SceneLds& S, Lane& L, bool has_ray, int coop_below) {
    // ---- hit(), kernels.cu:325-360: the ray is rebuilt from the path, which renormalises the direction
    const f3 dn = unit(L.dir);
    const float a = dot(dn, dn);
    Hit h = { FLT_MAX, -1 };
    const unsigned long long live = __ballot(has_ray);
    if (__popcll(live) >= coop_below) {
        if (has_ray) h = scan_lane_parallel(P, S, L.org, dn, a);
    } else {
        unsigned long long m = live;
        const int lane = threadIdx.x & 63;
        while (m) {
            const int q = __builtin_ctzll(m);
            m &= m - 1;
            const Hit hq = scan_cooperative(P, S, q, L.org, dn, a);
            if (lane == q) h = hq;
        }
    }
    bool done = false;
    if (has_ray) done = shade(P, S, L, dn, h);
    return done;
}

// ---- variant 1: one 8x8 pixel tile per wave, the lane keeps its pixel for the whole kernel ------------------------
__global__ void __launch_bounds__(kThreads) k_render_spheres_tiles(const RtSphereParams P, int coop_below) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* s_fb;
    const SceneLds S = stage_scene(P, smem, &s_fb);                  // s_fb: kThreads x 3 floats

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i0 = (blockIdx.x * kWavesPerWg + wave) * 8;            // left pixel column of this wave's 8x8 tile
    const int lr0 = blockIdx.y * 8;                                  // first local row of the tile
    const int i = i0 + (lane & 7);
    const int lr = lr0 + (lane >> 3);
    const bool valid = (i < P.nx) && (lr < P.part.local_rows);

    Lane L;
    L.col = F3(0, 0, 0);
    L.org = F3(0, 0, 0); L.dir = F3(0, 0, 1);
    uint32_t nrays = 0;
    bool active = valid && (P.ns > 0);
    if (active) start_pixel(P, L, i, global_row(P.part, lr));

    while (__ballot(active) != 0ull) {                               // wave-uniform loop: idle lanes stay to help
        if (active) nrays++;
        const bool done = trace_rays(P, S, L, active, coop_below);
        if (active && done) {
            L.col = L.col + L.pcolor;                                // kernels.cu:558
            L.s++;
            if (L.s < P.ns) start_sample(P, L);
            else active = false;
        }
    }

    // ---- framebuffer: fb[pixel] = col / ns (kernels.cu:568), transposed through LDS so that
    // consecutive lanes store consecutive dwords of a row segment (8 px * 12 B = 96 B per tile row)
    const f3 out = L.col / (float)P.ns;
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

// ---- work-order pre-pass ------------------------------------------------------------------------------------------
// The persistent kernel ends with a tail in which every lane finishes the pixel it happens to hold.  That tail is short
// when the LAST pixels handed out are cheap and alike.  A pixel whose centre ray (no lens offset, no jitter) misses every
// sphere is almost surely a sky pixel: one ray per sample, the cheapest and most uniform work there is.  This kernel
// sorts the tile-major pixel indices into two lists — list A "hits something" (handed out first, scattered), list B
// "sky" (handed out last) — with one atomic per wave and list.  The lists only change WHO renders a pixel and WHEN,
// never the result (the seed depends on the pixel id alone).  P.queue[4] / [5] = lengths of A / B.
__global__ void __launch_bounds__(kThreads) k_classify_spheres(const RtSphereParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* unused;
    const SceneLds S = stage_scene(P, smem, &unused);
    const int tiles_x = (P.nx + 7) >> 3;
    const int tiles_y = (P.part.local_rows + 7) >> 3;
    const uint32_t total = (uint32_t)tiles_x * (uint32_t)tiles_y * 64u;
    const uint32_t p = blockIdx.x * kThreads + threadIdx.x;
    const uint32_t tile = p >> 6, within = p & 63u;
    const int ty = (int)(tile / (uint32_t)tiles_x), tx = (int)(tile - (uint32_t)ty * (uint32_t)tiles_x);
    const int i = tx * 8 + (int)(within & 7u);
    const int lr = ty * 8 + (int)(within >> 3);
    const bool valid = p < total && i < P.nx && lr < P.part.local_rows;
    bool hits = false;
    if (valid) {
        const int j = global_row(P.part, lr);
        const float u = ((float)i + 0.5f) / (float)P.nx, v = ((float)j + 0.5f) / (float)P.ny;
        const f3 org = ld3(P.cam.origin);
        const f3 dn = unit(ld3(P.cam.lower_left_corner) + u * ld3(P.cam.horizontal) + v * ld3(P.cam.vertical) - org);
        for (int k = 0; k < P.n; k++) {
            const float4 sph = S.sph[k];
            const f3 oc = org - F3(sph.x, sph.y, sph.z);
            const float b = dot(oc, dn);
            const float c = dot(oc, oc) - sph.w;
            hits = hits || ((b * b - c > 0.0f) && (b < 0.0f || c < 0.0f));
        }
    }
    const unsigned long long ma = __ballot(valid && hits), mb = __ballot(valid && !hits);
    uint32_t base_a = 0, base_b = 0;
    if ((threadIdx.x & 63) == 0) {
        if (ma) base_a = atomicAdd(P.queue + 4, (uint32_t)__popcll(ma));
        if (mb) base_b = atomicAdd(P.queue + 5, (uint32_t)__popcll(mb));
    }
    base_a = __builtin_amdgcn_readfirstlane(base_a);
    base_b = __builtin_amdgcn_readfirstlane(base_b);
    if (valid) {
        const unsigned long long m = hits ? ma : mb;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (hits) P.order[base_a + rank] = p;
        else P.order[total + base_b + rank] = p;
    }
}

// ---- variant 0 (default): persistent waves + pixel queue ----------------------------------------------------------
// A pixel's samples are sequential (one RNG stream per pixel), so the pixel is the atom of work.  Pixels are
// numbered tile-major (8x8 tiles, row-major tiles, row-major inside a tile) and handed out from ONE global
// counter; in a wave, the lanes whose pixel is finished are counted with a wave64 ballot, ONE atomic reserves that
// many queue positions, and an mbcnt prefix sum gives each idle lane its own position (dense allocation).  Every lane
// therefore traces a ray in (almost) every iteration until the queue is empty; which lane renders which pixel is
// irrelevant to the result because the seed is a function of the global pixel id only.

__global__ void __launch_bounds__(kThreads) k_render_spheres_queue(const RtSphereParams P, int coop_below, uint32_t stride, int classified) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* unused;
    const SceneLds S = stage_scene(P, smem, &unused);

    const int tiles_x = (P.nx + 7) >> 3;
    const int tiles_y = (P.part.local_rows + 7) >> 3;
    const uint32_t padded = (uint32_t)tiles_x * (uint32_t)tiles_y * 64u;
    // classified order: queue positions [0, nA) walk list A in a scattered order, [nA, nA+nB) walk list B
    const uint32_t nA = classified ? P.queue[4] : 0u, nB = classified ? P.queue[5] : 0u;
    const uint32_t total = classified ? nA + nB : padded;
    if (classified && nA > 64u) {                                    // stride ~ 0.618 nA, coprime with nA (wave-uniform)
        uint32_t c = ((uint32_t)((unsigned long long)nA * 2654435769ull >> 32)) | 1u;
        for (;;) {
            uint32_t a = c, b = nA;
            while (b) { const uint32_t t = a % b; a = b; b = t; }
            if (a == 1u) break;
            c += 2u;
        }
        stride = c % nA;
    }

    Lane L;
    L.col = F3(0, 0, 0);
    L.org = F3(0, 0, 0); L.dir = F3(0, 0, 1);
    int lr = 0;                         // local row of the lane's pixel (framebuffer row)
    uint32_t nrays = 0;
    bool have_pixel = false;            // lane owns an unfinished pixel
    bool exhausted = false;             // wave-uniform: the global queue is empty
    float* fbf = reinterpret_cast<float*>(P.fb);
    // diagnostics (only when P.wave_dbg): 100 MHz time stamps and iteration counts of this wave
    const unsigned long long dbg_t0 = P.wave_dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long dbg_tex = 0ull;
    uint32_t dbg_iters = 0, dbg_coop_iters = 0, dbg_coop_rays = 0;

    while (true) {
        // ---- refill idle lanes --------------------------------------------------------------------------------
        while (!exhausted) {
            const unsigned long long need = __ballot(!have_pixel);
            if (need == 0ull) break;
            // exactly as many pixels as there are idle lanes: nothing is hoarded in a wave while other waves idle
            const uint32_t cnt = (uint32_t)__popcll(need);
            uint32_t base = 0;
            if ((threadIdx.x & 63) == 0) base = atomicAdd(P.queue, cnt);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= total) { exhausted = true; break; }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            if (base + cnt >= total) exhausted = true;               // this grab took the last pixels
            if (!have_pixel && base + rank < total) {
                // queue position -> pixel: a multiplicative permutation (stride coprime with total) scatters
                // neighbouring pixels over different waves, so the few very long pixels (paths trapped inside
                // glass for 50 bounces, clustered on sphere rims) never share a wave; stride 1 = tile-major order
                const uint32_t pos = base + rank;
                uint32_t p;
                if (!classified) p = (uint32_t)(((unsigned long long)pos * stride) % padded);
                else if (pos < nA) p = P.order[nA > 64u ? (uint32_t)(((unsigned long long)pos * stride) % nA) : pos];
                else p = P.order[padded + (pos - nA)];
                const uint32_t tile = p >> 6, within = p & 63u;
                const int ty = (int)(tile / (uint32_t)tiles_x), tx = (int)(tile - (uint32_t)ty * (uint32_t)tiles_x);
                const int i = tx * 8 + (int)(within & 7u);
                lr = ty * 8 + (int)(within >> 3);
                if (i < P.nx && lr < P.part.local_rows) {            // pixels of partial edge tiles are skipped
                    start_pixel(P, L, i, global_row(P.part, lr));
                    have_pixel = true;
                }
            }
        }
        const unsigned long long live_now = __ballot(have_pixel);
        if (live_now == 0ull) break;                                 // wave-uniform exit: idle lanes stay to help
        if (P.wave_dbg) {
            if (exhausted && dbg_tex == 0ull) dbg_tex = __builtin_amdgcn_s_memrealtime();
            dbg_iters++;
            if (__popcll(live_now) < coop_below) { dbg_coop_iters++; dbg_coop_rays += (uint32_t)__popcll(live_now); }
        }

        // ---- one ray per live lane ----------------------------------------------------------------------------
        if (have_pixel) nrays++;
        const bool done = trace_rays(P, S, L, have_pixel, coop_below);
        if (have_pixel && done) {
            L.col = L.col + L.pcolor;                                // kernels.cu:558
            L.s++;
            if (L.s < P.ns) {
                start_sample(P, L);
            } else {
                const f3 out = L.col / (float)P.ns;                  // kernels.cu:568
                float* dst = fbf + ((size_t)lr * P.nx + L.i) * 3;
                dst[0] = out.x; dst[1] = out.y; dst[2] = out.z;
                have_pixel = false;
            }
        }
    }

    if (P.counters) {
        atomicAdd(&P.counters->rays, (unsigned long long)nrays);
        atomicAdd(&P.counters->prim_tests, (unsigned long long)nrays * (unsigned long long)P.n);
    }
    if (P.wave_dbg && (threadIdx.x & 63) == 0) {
        unsigned long long* w = P.wave_dbg + ((size_t)blockIdx.x * kWavesPerWg + (threadIdx.x >> 6)) * 8;
        w[0] = dbg_t0; w[1] = dbg_tex; w[2] = __builtin_amdgcn_s_memrealtime();
        w[3] = dbg_iters; w[4] = dbg_coop_iters; w[5] = dbg_coop_rays;
        w[6] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID, HW_ID
        w[7] = 1;
    }
}

}  // namespace

static size_t lds_bytes(int n_padded, int n) {
    return (size_t)n_padded * 16 + (size_t)n * 16 + (size_t)((n + 3) & ~3) * 4 + (size_t)kThreads * 3 * 4;
}

#if defined(RT_MODE_PARITY)
size_t rt_sphere_kernel_lds_bytes(int n_padded, int n, int threads) {
    (void)threads;
    return lds_bytes(n_padded, n);
}
#endif

// variant: bits 0..7   kernel: 0 = persistent waves + pixel queue (default), 1 = one tile per wave;
//          bits 8..15  workgroups per CU of the persistent kernel (0 = default 4);
//          bits 16..23 switch to the wave-cooperative scan when fewer than this many lanes of a wave have a ray
//                      (0 = default 24; 1 = never cooperative, 65 = always cooperative);
//          bits 24..25 work order of the persistent kernel: 0 = classified (hit-something pixels scattered, sky
//                      pixels last; needs the classify pre-pass), 1 = tile-major, 2 = scattered only.
hipError_t RT_LAUNCH_NAME(const RtSphereParams& p, int variant, hipStream_t stream) {
    const size_t lds = lds_bytes(p.n_padded, p.n);
    const int kind = variant & 0xFF;
    const void* kern = (kind == 1) ? reinterpret_cast<const void*>(k_render_spheres_tiles)
                                   : reinterpret_cast<const void*>(k_render_spheres_queue);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_classify_spheres), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int coop_below = (variant >> 16) & 0xFF;
    if (coop_below == 0) coop_below = 24;
    if (kind == 1) {
        const dim3 grid((p.nx + 8 * kWavesPerWg - 1) / (8 * kWavesPerWg), (p.part.local_rows + 7) / 8);
        hipLaunchKernelGGL(k_render_spheres_tiles, grid, dim3(kThreads), lds, stream, p, coop_below);
        return hipGetLastError();
    }
    if (!p.queue) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(p.queue, 0, 64, stream);
    if (e != hipSuccess) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int wg_per_cu = (variant >> 8) & 0xFF;
    if (wg_per_cu == 0) wg_per_cu = 4;
    const long long total_px = (long long)((p.nx + 7) / 8) * ((p.part.local_rows + 7) / 8) * 64;
    long long blocks = (long long)cus * wg_per_cu;
    const long long useful = (total_px + kThreads - 1) / kThreads;      // never more lanes than pixels
    if (blocks > useful) blocks = useful;
    if (blocks < 1) blocks = 1;
    // scattered order: stride ~ 0.618 * total, coprime with total
    uint32_t stride = 1;
    const int order_mode = (variant >> 24) & 3;
    const int classified = (order_mode == 0 && p.order != nullptr) ? 1 : 0;
    if (classified) {
        hipLaunchKernelGGL(k_classify_spheres, dim3((unsigned)((total_px + kThreads - 1) / kThreads)), dim3(kThreads), lds, stream, p);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (order_mode != 1 && total_px > 64) {
        auto gcd = [](unsigned long long a, unsigned long long b) { while (b) { const unsigned long long t = a % b; a = b; b = t; } return a; };
        unsigned long long cand = (unsigned long long)((double)total_px * 0.6180339887) | 1ull;
        while (gcd(cand, (unsigned long long)total_px) != 1ull) cand += 2;
        stride = (uint32_t)(cand % (unsigned long long)total_px);
    }
    hipLaunchKernelGGL(k_render_spheres_queue, dim3((unsigned)blocks), dim3(kThreads), lds, stream, p, coop_below, stride, classified);
    return hipGetLastError();
}
