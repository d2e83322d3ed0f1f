// rt_kernels_mesh.hip — the render() hot path for TRIANGLE-MESH scenes on gfx950 (wave64).
//
// Replaces /root/reference/kernels.cu:535-569 (render), :396-533 (color), :325-360 (hit),
// :296-323 (hitMesh), :154-224 (hitBvh, DUAL_NODES), :148-152 (pop_bitstack) and :363-393
// (generateShadowRay), plus the primitive tests of intersections.h:7-41,54-104.
// Compiled twice (see rt_kernels_spheres.hip): PARITY (-ffp-contract=off: bit-exact vs the oracle, light sampling included - cosf / sinf / powf are
// glibc's algorithms restated, rt_glibc_sincosf.h / rt_glibc_powf.h) and FAST (FMA contraction).
//
// MI355X design:
//   * one lane = one pixel + its RNG stream; persistent waves, a pixel queue, and a per-lane ray-JOB state machine (closest-hit or shadow ray, same code):
//     a wave alternates between TRAVERSE (node steps for the lanes at an internal node, or (ray, triangle) pair rounds for the lanes waiting at a leaf)
//     and PROCESS (scatter, next-event estimation, Russian roulette in the reference's order for the lanes whose traversal ended);
//   * BVH child pairs are read from axis-grouped 96-byte records (per axis the four planes in both orders: the `if (invD < 0) swap` of
//     intersections.h:30 is a choice of address), three global_load_dwordx4 per node visit (the reference used a 1D texture object: kernels.cu:166-173);
//   * leaves are read as compact 48-byte records (v0 and the two edges, rounded as intersections.h:56-57 rounds them; the count of real triangles per
//     leaf in the LDS): 1.5 MB of node records + 1.7 MB of leaf records touched per frame of the benchmark tree - one XCD's L2 holds them;
//   * 1/direction is computed once per ray instead of per box per axis (intersections.h:28) - the same quotient each time, so bit-identical; the slab
//     test runs all three axes branch-free (t_min only grows, t_max only shrinks: the final `t_max < t_min` equals the per-axis early-out);
//   * a lean instantiation (template LEAN) for untextured scenes of the three basic materials: fewer registers, no (u, v) through the traversal.
#include "rt_device.h"
#include "rt_params.h"

#include <float.h>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdio>

using namespace rtd;

#if defined(RT_MODE_PARITY)
#define RT_LAUNCH_NAME rt_launch_mesh_parity
#elif defined(RT_MODE_FAST)
#define RT_LAUNCH_NAME rt_launch_mesh_fast
#else
#error "define RT_MODE_PARITY or RT_MODE_FAST"
#endif

namespace {

#ifndef RT_MESH_WG_WAVES
#define RT_MESH_WG_WAVES 4          // waves per workgroup (experiment: 16 = one workgroup per CU, one LDS copy of the tables)
#endif
constexpr int kWavesPerWg = RT_MESH_WG_WAVES;
constexpr int kThreads = 64 * kWavesPerWg;

__device__ __forceinline__ int global_row(const RtPartition& pt, int lr) {
    const int stripe = lr / pt.stripe_rows;
    return (stripe * pt.world + pt.rank) * pt.stripe_rows + (lr - stripe * pt.stripe_rows);
}

struct Tri { f3 v0, v1, v2; float tc[6]; int meshID; };

__device__ __forceinline__ Tri load_tri(const rt_triangle* tris, uint32_t id) {
    const float4* p = reinterpret_cast<const float4*>(tris + id);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    Tri t;
    t.v0 = F3(a.x, a.y, a.z); t.v1 = F3(a.w, b.x, b.y); t.v2 = F3(b.z, b.w, c.x);
    t.tc[0] = c.y; t.tc[1] = c.z; t.tc[2] = c.w; t.tc[3] = d.x; t.tc[4] = d.y; t.tc[5] = d.z;
    t.meshID = (int)(__float_as_uint(d.w) & 0xFFu);
    return t;
}

struct TravStats { uint32_t nodes, tests; };
typedef float f4u __attribute__((ext_vector_type(4)));

// hitBvh (DUAL_NODES), kernels.cu:154-224
__device__ __forceinline__ float hit_bvh(const RtMeshParams& P, const Ray& r, float t_min, float t_max, bool is_shadow,
                                         uint32_t& triId, float& hu, float& hv, TravStats& st) {
    int idx = 1;
    float closest = t_max;
    uint32_t bitStack = 1;
    while (idx) {
        if ((uint32_t)idx < P.first_leaf) {
            const int idx2 = idx << 1;
            const float4* n = P.bvh4 + (size_t)idx * 3;
            const float4 na = n[0], nb = n[1], nc = n[2];
            st.nodes++;
            const float leftHit = hit_bbox_dist(F3(na.x, na.y, na.z), F3(na.w, nb.x, nb.y), r, closest);
            const bool traverseLeft = leftHit < closest;
            const float rightHit = hit_bbox_dist(F3(nb.z, nb.w, nc.x), F3(nc.y, nc.z, nc.w), r, closest);
            const bool traverseRight = rightHit < closest;
            const bool swap = rightHit < leftHit;
            if (traverseLeft && traverseRight) {
                idx = idx2 + (swap ? 1 : 0);
                bitStack = (bitStack << 1) + 1;
            } else if (traverseLeft || traverseRight) {
                idx = idx2 + (swap ? 1 : 0);
                bitStack = bitStack << 1;
            } else {
                const int m = __ffs((int)bitStack) - 1;              // pop_bitstack, kernels.cu:148-152
                bitStack = (bitStack >> m) ^ 1u;
                idx = (idx >> m) ^ 1;
            }
        } else {
            const uint32_t first = ((uint32_t)idx - P.first_leaf) * P.nppl;
            for (uint32_t i = 0; i < P.nppl; i++) {
                const float4* p = reinterpret_cast<const float4*>(P.tris + first + i);
                const float4 a = p[0], b = p[1];
                if (isinf(a.x)) break;                               // kernels.cu:202 sentinel
                const float cx = p[2].x;
                float u, v;
                st.tests++;
                const float hitT = triangle_hit(F3(a.x, a.y, a.z), F3(a.w, b.x, b.y), F3(b.z, b.w, cx), r, t_min, closest, u, v);
                if (hitT < closest) {
                    if (is_shadow) return 0.0f;
                    closest = hitT;
                    triId = first + i;
                    hu = u; hv = v;
                }
            }
            const int m = __ffs((int)bitStack) - 1;
            bitStack = (bitStack >> m) ^ 1u;
            idx = (idx >> m) ^ 1;
        }
    }
    return closest;
}

// hitMesh, kernels.cu:296-323
__device__ __forceinline__ float hit_mesh(const RtMeshParams& P, const Ray& r, float t_min, float t_max, bool is_shadow,
                                          uint32_t& triId, float& hu, float& hv, TravStats& st) {
    if (!hit_bbox(ld3(P.bounds.min), ld3(P.bounds.max), r, t_max)) return FLT_MAX;
    return hit_bvh(P, r, t_min, t_max, is_shadow, triId, hu, hv, st);
}

template <int VARIANT>
__global__ void __launch_bounds__(kThreads) k_render_mesh(const RtMeshParams P) {
    __shared__ float s_fb[kThreads * 3];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i0 = (blockIdx.x * kWavesPerWg + wave) * 8;
    const int lr0 = blockIdx.y * 8;
    const int i = i0 + (lane & 7);
    const int lr = lr0 + (lane >> 3);
    const bool valid = (i < P.nx) && (lr < P.part.local_rows);
    const int j = global_row(P.part, lr);
    const uint32_t pixelId = (uint32_t)(j * P.nx + i);

    const float eps = P.t_min;
    const f3 lightC = ld3(P.light.center);
    const float lightR = P.light.radius;

    uint32_t rng = pixel_seed(pixelId);
    f3 col = F3(0, 0, 0);
    f3 org = F3(0, 0, 0), dir = F3(0, 0, 1), atten = F3(1, 1, 1), pcolor = F3(0, 0, 0);
    int bounce = 0;
    bool inside = false, specular = false;
    int s = 0;
    TravStats st = { 0, 0 };
    uint32_t nrays = 0, nshadow = 0;

    auto start_sample = [&]() {                                      // kernels.cu:549-555, 397-398
        if (P.rng_mode == RT_RNG_COUNTER) rng = sample_seed(pixelId, (uint32_t)s);
        const float u = ((float)i + rnd(rng)) / (float)P.nx;
        const float v = ((float)j + rnd(rng)) / (float)P.ny;
        f3 d;
        get_ray(P.cam, u, v, rng, org, d);
        dir = unit(d);
        atten = F3(1.0f, 1.0f, 1.0f);
        pcolor = F3(0, 0, 0);
        bounce = 0;
        inside = false;
        specular = false;
    };

    bool active = valid && (P.ns > 0);
    if (active) start_sample();

    while (active) {
        // ---- hit(context, p, FLT_MAX, false, inters), kernels.cu:325-360
        const Ray r = make_ray(org, dir);
        uint32_t triId = 0;
        float hu = 0.0f, hv = 0.0f;
        nrays++;
        const float t = hit_mesh(P, r, eps, FLT_MAX, false, triId, hu, hv, st);
        bool path_done = false;
        if (t < FLT_MAX) {
            const Tri tri = load_tri(P.tris, triId);                 // kernels.cu:334
            f3 normal = unit(cross(tri.v1 - tri.v0, tri.v2 - tri.v0));
            const float w0 = 1 - hu - hv;
            const float tcu = (hu * tri.tc[2] + hv * tri.tc[4] + w0 * tri.tc[0]);
            const float tcv = (hu * tri.tc[3] + hv * tri.tc[5] + w0 * tri.tc[1]);
            if (dot(r.d, normal) > 0.0f) normal = -normal;           // kernels.cu:354-355

            const rt_material mat = P.materials[tri.meshID];         // kernels.cu:452-480
            f3 albedo;
            if (mat.texId != -1) {
                const int width = P.tex_width[mat.texId];
                const int height = P.tex_height[mat.texId];
                float tu = tcu; tu = tu - floorf(tu);
                float tv = tcv; tv = tv - floorf(tv);
                const int tx = (int)((float)(width - 1) * tu);
                const int ty = (int)((float)(height - 1) * tv);
                const int tIdx = ty * width + tx;
                const float* d = P.tex_data[mat.texId];
                albedo = F3(d[tIdx * 3 + 0], d[tIdx * 3 + 1], d[tIdx * 3 + 2]);
            } else {
                albedo = ld3(mat.color);
            }
            Scatter sc;
            material_scatter(sc, t, r.o + t * r.d, normal, inside, dir, mat.type, albedo, mat.param, rng);
            org = org + sc.t * dir;                                  // kernels.cu:485-489
            dir = sc.wi;
            atten = atten * sc.throughput;
            specular = sc.specular;
            inside = sc.refracted ? !inside : inside;

            if (P.nee && !specular) {                                // generateShadowRay, kernels.cu:363-393 (rt_device.h)
                ShadowSample sh;
                if (generate_shadow_ray(lightC, lightR, ld3(P.lightColor), org, atten, normal, rng, sh)) {
                    const Ray sr = make_ray(org, sh.dir);
                    uint32_t tid2 = 0; float u2, v2;
                    nshadow++;
                    const float ts = hit_mesh(P, sr, eps, sh.dist, true, tid2, u2, v2, st);
                    if (!(ts < sh.dist)) pcolor = pcolor + sh.contrib;       // kernels.cu:504-510
                }
            }
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
            if (bounce >= P.max_depth) path_done = true;
        } else if (specular && sphere_hit(lightC, lightR, r, eps, FLT_MAX) < FLT_MAX) {   // kernels.cu:346
            if (!P.nee) pcolor = pcolor + atten * ld3(P.lightColor);                      // kernels.cu:440-446
            path_done = true;
        } else {
            pcolor = pcolor + atten * sky_color(P.sky, dir);         // kernels.cu:419-425
            path_done = true;
        }

        if (path_done) {
            col = col + pcolor;
            s++;
            if (s < P.ns) start_sample();
            else active = false;
        }
    }

    const f3 out = col / (float)P.ns;                                // kernels.cu:568
    float* my = s_fb + threadIdx.x * 3;
    my[0] = out.x; my[1] = out.y; my[2] = out.z;
    __syncthreads();
    const float* wfb = s_fb + wave * 192;
    float* fbf = reinterpret_cast<float*>(P.fb);
#pragma unroll
    for (int q = lane; q < 192; q += 64) {
        const int row = q / 24, off = q - row * 24;
        const int px = i0 + off / 3;
        const int rr = lr0 + row;
        if (px < P.nx && rr < P.part.local_rows)
            fbf[((size_t)rr * P.nx + i0) * 3 + off] = wfb[q];
    }

    if (P.counters) {
        atomicAdd(&P.counters->rays, (unsigned long long)nrays);
        atomicAdd(&P.counters->shadow_rays, (unsigned long long)nshadow);
        atomicAdd(&P.counters->prim_tests, (unsigned long long)st.tests);
        atomicAdd(&P.counters->node_visits, (unsigned long long)st.nodes);
    }
}


// ---- variant 0 (default): persistent waves, pixel queue, ray-job state machine ---------------------------------------
// The first kernel (above) nests two traversals (closest hit, then the shadow ray) inside one loop iteration and lets a
// lane whose traversal ended wait for the slowest lane of the wave: rocprof showed 17 % of the lanes active per VALU
// instruction.  Here every lane owns a JOB = one ray to traverse (closest-hit or shadow, same code) and the wave
// alternates between two phases:
//   TRAVERSE  while at least `kMinTraversing` lanes still have nodes to visit: "while-while" steps — all lanes descend
//             internal nodes together until each sits on a leaf (or is done), then all lanes test their leaf's triangles.
//             The per-lane visiting order is exactly hitBvh's (kernels.cu:154-224), only the control flow is regrouped;
//   PROCESS   the lanes whose traversal has ended consume the result (scatter + next-event estimation + Russian roulette
//             exactly in the reference's order, kernels.cu:415-527) and set up their next job: the shadow ray, the next
//             bounce, the next sample, or the next pixel from the global queue (ballot + mbcnt allocation).
// Lanes that are still traversing simply keep their state across a PROCESS phase.
constexpr int kMinTraversing = 40;
constexpr uint32_t kLeafCntLds = 32768;     // leaves whose triangle counts the default kernel keeps in the LDS (one byte each, beside its 5 KB of pair-round scratch)

struct Job {
    f3 d, inv;              // unit direction and its reciprocal (ray.h:9, intersections.h:28); the ORIGIN is the path's `org`: every job of a path starts there,
                            // and `org` only moves in PROCESS, when no job is running
    int idx;                // 0 = traversal finished
    uint32_t bitStack;
    float closest;          // running closest t (t_max at start)
    uint32_t triId;
    float hu, hv;           // (only a textured material reads them: the lean instantiation does not carry them)
    bool shadow;
    uint32_t ax, ay, az;    // byte offsets of the ray's (near_L, near_R, far_L, far_R) on each axis inside a bvh_axis record
};

__device__ __forceinline__ void job_start(const RtMeshParams& P, Job& J, f3 org, f3 dir, float t_max, bool shadow) {
    const Ray r = make_ray(org, dir);
    J.d = r.d; J.inv = r.inv;
    J.shadow = shadow;
    J.closest = t_max;
    J.bitStack = 1;
    J.triId = 0; J.hu = 0.0f; J.hv = 0.0f;
    J.ax = r.inv.x < 0.0f ? 16u : 0u;                    // `if (invD < 0.0f) swap(t0, t1)`, intersections.h:30, as a choice of address
    J.ay = r.inv.y < 0.0f ? 48u : 32u;
    J.az = r.inv.z < 0.0f ? 80u : 64u;
    // hitMesh, kernels.cu:296-323: scene bounds first; a miss reports FLT_MAX
    if (hit_bbox(ld3(P.bounds.min), ld3(P.bounds.max), r, t_max)) {
        J.idx = 1;
    } else {
        J.idx = 0;
        J.closest = FLT_MAX;
    }
}

// TRAV 0: thresholded while-while (default); TRAV 1: classic while-while (all lanes descend to a leaf, then all test their leaf)
// STATS: the reference's `#ifdef STATS` counters (kernels.cu:47-67,399-561) as device atomics - the counting instantiation only.
// (Round 2 built and measured a form with TWO concurrent jobs per lane - the shadow ray beside the next bounce ray, both known
// after a diffuse hit; Russian roulette and the light sample draw in the reference's order, the shadow result is added before
// anything else: bit-exact.  One PROCESS per bounce instead of two (-31 % process steps), 33 instead of 31 lanes per node step -
// but every node step had to select the lane's job (+28 % cycles per step): 573-590 against 609 Msamples/s.  Not kept.)
// LEAN (the launcher's promise: every material is RT_DIFFUSE / RT_METAL / RT_GLASS without a texture - what scene_materials.h:13-20 at HEAD can produce -, no
// floor plane, pair rounds available): PROCESS without the preset tables, textures and the plane, no (u, v) carried through the traversal - fewer registers,
// so more waves per SIMD (RT_MESH_LEAN_WAVES) to hide the dependent node loads behind.
#ifndef RT_MESH_HEAVY_CLS
#define RT_MESH_HEAVY_CLS 5         // PHASE 2: cost classes counted as expensive (>= 1.6 x the mean pixel) and spread over the first fills (0 = off) ...
#define RT_MESH_SPREAD_ROUNDS 2     // ... of this many times the lanes in flight.  C4, A/B in one call (profiles/r04_sweep_mesh_spread*.txt): lists as they lie 907, 5 classes over
                                    // 1 / 2 / 3 / 5 fills 972 / 937-987 / 954 / 938, 4 classes 969 / 967, 6 classes 903 / 945, 7 classes 905-919 Msamples/s
#endif
#ifndef RT_MESH_CHAIN_THR
#define RT_MESH_CHAIN_THR 448       // PHASE 2: list 0 = pixels from 16 x this many cost units per sample (the mean pixel of C4 has ~100): the chains, see "Chain waves"
#endif
#ifndef RT_MESH_CHAIN_LANES
#define RT_MESH_CHAIN_LANES 6       // pixels of list 0 per chain wave (0 = no chain waves)
#endif
#ifndef RT_MESH_TAIL_DIAG
#define RT_MESH_TAIL_DIAG 0
#endif
#if RT_MESH_TAIL_DIAG
__device__ uint32_t* g_diag_items;      // per pixel: (start, end) of its second-dispatch item, microseconds
#endif
#ifndef RT_MESH_LEAN_WAVES
#define RT_MESH_LEAN_WAVES 4
#endif
// PHASE (the cost-ordered frame, rt_params.h): 0 = whole pixels in one dispatch (scattered order); 1 = samples [0, s_split) of every pixel, then park the pixel;
// 2 = resume the parked pixels in the order of P.order (longest first; since round 4 from one set of lists and counters per XCD, with 32-byte records - the sphere
// kernel's traffic forms, DESIGN.md 3.8: here they halve the reads of the parked state, the writes still leave the L2 as partial lines).  A frame ends one pixel-time after its queue runs empty, and a pixel of C4 is 1/8 of a
// lane's frame: in scattered order the 1920x1080x256 frame ran at 782 Msamples/s where the same samples as a 3840x2160x64 frame ran at 1106
// (profiles/r04_mesh_tail_probe.txt) - the expensive pixels were still running when everything else was done.
template <int TRAV, bool DBG, bool STATS, bool LEAN = false, int PHASE = 0>
__global__ void __launch_bounds__(kThreads, LEAN ? RT_MESH_LEAN_WAVES : (TRAV == 0 ? 4 : 5)) k_render_mesh_queue(const RtMeshParams P, uint32_t stride, int min_traversing, int leaf_thr) {
    const int tiles_x = (P.nx + 7) >> 3;
    const int tiles_y = (P.part.local_rows + 7) >> 3;
    const uint32_t total = (uint32_t)tiles_x * (uint32_t)tiles_y * 64u;
    const float eps = P.t_min;
#if RT_MESH_TAIL_DIAG
    const unsigned long long t_start_diag = __builtin_amdgcn_s_memrealtime();
#endif
    // leaf phase of TRAV 0: per-wave LDS scratch and this lane's fixed role in a pair round (ray j / nppl, triangle j % nppl)
    __shared__ uint32_t s_owner[kThreads];
    __shared__ unsigned long long s_best[kThreads];
    __shared__ float2 s_uv[kThreads];
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* w_owner = s_owner + (threadIdx.x & ~63u);
    unsigned long long* w_best = s_best + (threadIdx.x & ~63u);
    float2* w_uv = s_uv + (threadIdx.x & ~63u);
    // (pair rounds need the compact leaf records of the host - RtMeshParams::leaf_tri / leaf_ofs, built when no leaf has a real triangle behind a sentinel)
    const int pair_per = (TRAV == 0 && P.leaf_sentinels_trailing && P.leaf_ofs != nullptr && P.leaf_tri != nullptr && P.nppl >= 1u && P.nppl <= 16u &&
                          P.first_leaf <= kLeafCntLds) ? (int)(64u / P.nppl) : 0;     // rays per pair round
    const uint32_t pair_r = pair_per ? lane / P.nppl : 0u;
    const uint32_t pair_k = pair_per ? lane - pair_r * P.nppl : 0u;
    const bool pair_ok = pair_per && (int)pair_r < pair_per;
    // Compact leaf records (rt_params.h): a pair lane reads the 48 bytes triangleHit needs (v0 and the two edges) of a REAL triangle only - the number
    // of real triangles of every leaf sits in the LDS (one byte per leaf, staged once per workgroup), so the sentinel slots of a leaf (on the
    // benchmark tree 2.8 of its 5) are never requested: 1.7 MB of triangle data touched instead of 5.2 MB.
    extern __shared__ __align__(16) unsigned char s_leaf_cnt[];
    if (pair_per > 0) {
        const uint32_t words = (P.first_leaf + 3u) >> 2;
        for (uint32_t k = threadIdx.x; k < words; k += kThreads) reinterpret_cast<uint32_t*>(s_leaf_cnt)[k] = P.leaf_ofs[k];
        __syncthreads();
    }
    const f3 lightC = ld3(P.light.center);
    const float lightR = P.light.radius;
    float* fbf = reinterpret_cast<float*>(P.fb);
    // PHASE 2 work order.  P.order holds the pixels by descending cost class.  Taken as it lies, the first fill gives every wave 64 pixels of the top class - and a
    // pixel advances one node step per step of its wave, whose time grows with the lanes that are traversing (the loads, not their latency, bound a step): a wave
    // full of long traversals makes each of them ~1.5x longer, and the frame ends with them.  So the expensive lists (the first `heavy_cls` classes: nH pixels)
    // are SPREAD evenly over the first S queue positions, S = spread_rounds x (lanes in flight): position p takes an expensive pixel iff floor((p + 1) nH / S) >
    // floor(p nH / S), else the next of the rest (also by descending cost).  (leaf_thr >> 8: heavy_cls | spread_rounds << 4, from the launcher.)
    // One set of lists and counters for the machine, or (P.xcd_queues, as in the sphere kernel: rt_params.h) one per XCD: a wave serves the queue of the XCD it runs
    // on - the numbers below are then those of that XCD's lists and of an eighth of the grid - and when that is empty takes what the others have left (s_mx, `stolen`).
    __shared__ uint32_t s_m[4];         // [0] nH  [1] S  [2] E  [3] chain lanes (PHASE 2; of this wave's own queue)
    __shared__ uint32_t s_mx[8 * kXcdQueues];      // per queue: [0] nH [1] S [2] E [4] first position of its lists in P.order [5] items of its general part
    const uint32_t xq = (PHASE == 2 && P.xcd_queues == kXcdQueues) ? (uint32_t)kXcdQueues : 1u;
    const uint32_t myx = xq > 1u ? ((uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) & (uint32_t)(kXcdQueues - 1)) : 0u;       // XCC_ID
    if (PHASE == 2 && threadIdx.x < xq) {
        const uint32_t* const Q = P.queue + (size_t)threadIdx.x * kXcdQueueWords;
        const uint32_t grid_x = xq > 1u ? max(gridDim.x / xq, 1u) : gridDim.x;      // workgroups that serve this queue first
        uint32_t pixels = 0;
        for (int c = 0; c < 18; c++) pixels += Q[4 + c];
        const int heavy_cls = (leaf_thr >> 8) & 0xF, rounds = (leaf_thr >> 12) & 0xF;
        uint32_t nH = 0;
        for (int c = 0; c < heavy_cls; c++) nH += Q[4 + c];
        // the chains: list 0 (the launcher's threshold: pixels several times the mean) goes, `chain_lanes` pixels per wave, to the first waves that ask, and those
        // waves take nothing else while these run (below); at most one wave in eight
        uint32_t chain_lanes = (uint32_t)(leaf_thr >> 16) & 0xFFu;
        uint32_t E = chain_lanes ? Q[4] : 0u;
        if (E > nH) E = nH;
        if (E > (pixels >> ((leaf_thr >> 24) & 0xF))) E = 0u;
        // The chains end the frame only when a pixel of 10-20 x the mean (at about half the time per node visit of a cheap one) outlasts the whole queue, i.e. when
        // the frame is less than ~16 fills of the machine (C4: 7.9; 3840x2160: 31.6 - chain waves measured -3 % there, +12 % on C4; profiles/r04_mesh_chain_scenes.txt)
        if (((leaf_thr >> 24) & 0xF) != 0 && (uint32_t)P.part.local_rows * (uint32_t)P.nx > 16u * gridDim.x * blockDim.x) E = 0u;      // list 0 is not "a few pixels" in this scene (its threshold is absolute): no chain waves
        const uint32_t max_waves = grid_x * (blockDim.x >> 6) >> 3;
        if (E > max_waves * chain_lanes) chain_lanes = min(64u, (E + max_waves - 1u) / max_waves);      // (a long list: more of its pixels per wave, not fewer of them in chain waves)
        if (E > max_waves * chain_lanes) E = max_waves * chain_lanes;
        const uint32_t N = pixels - E;
        nH -= E;
        uint32_t S = (uint32_t)rounds * grid_x * blockDim.x;
        if (S > N) S = N;
        if (heavy_cls == 0 || rounds == 0 || nH == 0u || nH > S || S - nH > N - nH) { nH = 0u; S = 0u; }      // (off: the lists as they lie)
        uint32_t* const sm = s_mx + 8 * threadIdx.x;
        sm[0] = nH; sm[1] = S; sm[2] = E; sm[4] = xq > 1u ? Q[3] : 0u; sm[5] = N;
        if (threadIdx.x == myx) { s_m[0] = nH; s_m[1] = S; s_m[2] = E; s_m[3] = chain_lanes; }
    }
    if (PHASE == 2) __syncthreads();
    const int chain_lanes = PHASE == 2 ? (int)s_m[3] : 0;
    leaf_thr &= 0xFF;

    // path state (path, helper_structs.h:48-71)
    uint32_t rng = 1;
    f3 col = F3(0, 0, 0), org = F3(0, 0, 0), dir = F3(0, 0, 1), atten = F3(1, 1, 1), pcolor = F3(0, 0, 0);
    f3 pend_contrib = F3(0, 0, 0);      // lightContribution of the pending shadow ray
    float pend_dist = 0.0f;
    int bounce = 0, s = 0, pi = 0, pj = 0, lr = 0;
    bool inside = false, specular = false;
    uint32_t pixelId = 0;
    Job J;
    J.idx = 0; J.shadow = false; J.closest = FLT_MAX; J.bitStack = 1; J.triId = 0; J.hu = J.hv = 0.0f;
    J.d = F3(0, 0, 1); J.inv = F3(0, 0, 1);
    bool have_pixel = false, exhausted = false;
    TravStats st = { 0, 0 };
    uint32_t nrays = 0, nshadow = 0;
    uint32_t pix_jobs = 0;              // PHASE 1: node visits of this pixel so far: its measured cost
    const int s_end = PHASE == 1 ? P.s_split : P.ns;
    const uint32_t n_chain = PHASE == 2 ? s_m[2] : 0u;
    const uint32_t n_items = PHASE == 2 ? s_mx[8 * myx + 5] : total;      // the lists hold the valid pixels only
    const uint32_t seg_own = PHASE == 2 ? s_mx[8 * myx + 4] : 0u;          // first position of this wave's own lists in P.order
    uint32_t* const Qown = P.queue + (size_t)myx * kXcdQueueWords;
    uint32_t stolen = 0;                // (a queue per XCD) stages beyond this wave's own general queue known to be empty: stage 2k + 1 = what is left of the chain
                                        // pixels of queue (myx + k) mod 8, stage 2k + 2 = the general part of queue (myx + k + 1) mod 8
    const bool p1seg = PHASE == 1 && (min_traversing & 256) != 0;      // first dispatch: the permutation moves row segments of 8 pixels (a line of px_state per 8 lanes)
    min_traversing &= 255;
    // diagnostics (P.dbg): cycles and active lanes per phase, per wave; summed over the waves at the end
    unsigned long long g_cyc[4] = { 0, 0, 0, 0 };   // process, refill, node loop, leaf
    unsigned long long g_act[4] = { 0, 0, 0, 0 }, g_it[4] = { 0, 0, 0, 0 };   // active lanes summed over steps; steps
    const bool dbg = DBG && P.dbg != nullptr;       // DBG = false: the diagnostics (and their SGPR pressure) compile away
    constexpr bool COUNT = STATS || DBG;            // ray / node / triangle counters: only the instantiations launched when counters are asked for
    auto stat = [&](int k) { if (STATS && P.counters) atomicAdd(&P.counters->ref_stats[k], 1ull); };
    bool from_mesh = false;                         // the path's previous hit was the triangle mesh (STATS, kernels.cu:399-432)

    // A lane needs a new ray job in four places (new sample, new pixel, next bounce, shadow ray); job_start (ray set-up with
    // three IEEE divides + the scene-bounds test) is called at ONE place for all of them, after PROCESS and the refill:
    // as separate call sites a wave ran it once per place.  1 = closest-hit ray along `dir`, 2 = shadow ray along `shadow_dir`.
    int want_job = 0;
    bool need_sample = false;
    f3 shadow_dir = F3(0, 0, 1);
    auto start_sample = [&]() {                                      // kernels.cu:549-555, 397-398
        if (P.rng_mode == RT_RNG_COUNTER) rng = sample_seed(pixelId, (uint32_t)s);
        const float u = ((float)pi + rnd(rng)) / (float)P.nx;
        const float v = ((float)pj + rnd(rng)) / (float)P.ny;
        f3 d;
        get_ray(P.cam, u, v, rng, org, d);
        dir = unit(d);
        atten = F3(1.0f, 1.0f, 1.0f);
        pcolor = F3(0, 0, 0);
        bounce = 0;
        inside = false;
        specular = false;
        from_mesh = false;
        nrays++;
        stat(RT_STAT_PRIMARY);
        want_job = 1;                                               // hit(context, p, FLT_MAX, false, ...)
    };

    auto resume_pixel = [&](uint32_t pos) {                          // PHASE 2: the pixel's stream continues where the first dispatch left it
        uint32_t packed;
        float4 st4;
        if (P.ord_rec) {                                             // one 32-byte record per queue position (RtSphereParams::ord_rec)
            st4 = P.ord_rec[2 * (size_t)pos];
            packed = __float_as_uint(P.ord_rec[2 * (size_t)pos + 1].x);
        } else {
            packed = P.order[pos];
            st4 = P.ord_state[pos];
        }
        pi = (int)(packed & 0xFFFFu); lr = (int)(packed >> 16);
        pj = global_row(P.part, lr);
        pixelId = (uint32_t)(pj * P.nx + pi);
        rng = __float_as_uint(st4.w);
        col = F3(st4.x, st4.y, st4.z);
        s = P.s_split;
        have_pixel = true;
        need_sample = true;
#if RT_MESH_TAIL_DIAG
        g_diag_items[((size_t)lr * P.nx + pi) * 2] = (uint32_t)(__builtin_amdgcn_s_memrealtime() / 100ull);
#endif
    };
    // Chain waves (PHASE 2).  The frame cannot end before its most expensive pixel has traced its 250 samples one after the other (the stream is
    // sequential), and in a wave of 64 busy lanes a pixel advances only in the steps of its own kind (node / leaf) and waits through the others: measured,
    // the ~60 most expensive pixels of C4 ended 100 ms after everything else (profiles/r04_mesh_tail_diag.txt).  So the first waves to ask take `chain_lanes`
    // pixels of list 0 each and nothing else until those are done: few lanes, every step is theirs (leaf phase as soon as one lane waits, PROCESS as soon as one
    // traversal ends); then the wave joins the others at the queue.
    const int leaf_thr0 = leaf_thr, min_traversing0 = min_traversing;
    bool chain_wave = false;                                         // (wave-uniform) holds pixels of list 0 only; when they are done it joins the others
    bool chain_ask = PHASE == 2 && n_chain != 0u;                    // (asked at the first refill: a pixel is taken where the others are, after PROCESS)

    while (true) {
        // ================= PROCESS: lanes without a running traversal =======================================
        unsigned long long c0 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
        if (dbg) { const int n_ = (int)__popcll(__ballot(have_pixel && J.idx == 0)); g_act[0] += (unsigned long long)n_; g_it[0] += n_ ? 1ull : 0ull; }
        if (have_pixel && J.idx == 0) {
            bool path_done = false;
            bool next_ray = false;                                   // continue the path with a new closest-hit job
            if (!J.shadow) {
                // ---- the result of hit(context, p, FLT_MAX, false, inters), kernels.cu:325-360
                const bool primary = bounce == 0;
                const float t = J.closest;
                int obj = 0;                                         // 0 none, 1 triangle mesh, 2 floor, 3 light
                float t_hit = t;
                f3 normal = F3(0, 1, 0);
                f3 albedo = F3(0, 0, 0);
                int mtype = RT_FLOOR_DIFFUSE;                        // floor_diffuse_scatter, scene_materials.h:30-33 (kernels.cu:481-482)
                float mparam = 0.0f;
                if (t < FLT_MAX) {
                    obj = 1;
                    // kernels.cu:334 re-loads the triangle.  With compact leaf records the normal and the material come from the record the leaf test read
                    // (the two edges ARE v1 - v0 and v2 - v0 as kernels.cu:336 computes them; meshID rides in the record): the caller's 64-byte array is
                    // touched only for the texture coordinates of a TEXTURED material - the traversal's working set (1.5 MB of node records + 1.7 MB of
                    // records) then fits one XCD's L2.
                    f3 e1, e2;
                    int mesh_id;
                    if (pair_per > 0) {
                        const float4* rec = P.leaf_tri + (size_t)J.triId * 3;
                        const float4 ra = rec[0], rb = rec[1], rc = rec[2];
                        e1 = F3(ra.w, rb.x, rb.y); e2 = F3(rb.z, rb.w, rc.x);
                        mesh_id = (int)(__float_as_uint(rc.y) & 0xFFu);
                    } else {
                        const Tri tri = load_tri(P.tris, J.triId);
                        e1 = tri.v1 - tri.v0; e2 = tri.v2 - tri.v0;
                        mesh_id = tri.meshID;
                    }
                    normal = unit(cross(e1, e2));
                    const rt_material mat = P.materials[mesh_id];    // kernels.cu:452-480
                    mtype = mat.type; mparam = mat.param;
                    if (!LEAN && mat.texId != -1) {
                        const float* tc = P.tris[J.triId].texCoords;
                        const float w0 = 1 - J.hu - J.hv;
                        const float tcu = (J.hu * tc[2] + J.hv * tc[4] + w0 * tc[0]);
                        const float tcv = (J.hu * tc[3] + J.hv * tc[5] + w0 * tc[1]);
                        const int width = P.tex_width[mat.texId];
                        const int height = P.tex_height[mat.texId];
                        float tu = tcu; tu = tu - floorf(tu);
                        float tv = tcv; tv = tv - floorf(tv);
                        const int tx = (int)((float)(width - 1) * tu);
                        const int ty = (int)((float)(height - 1) * tv);
                        const int tIdx = ty * width + tx;
                        const float* d = P.tex_data[mat.texId];
                        albedo = F3(d[tIdx * 3 + 0], d[tIdx * 3 + 1], d[tIdx * 3 + 2]);
                    } else {
                        albedo = ld3(mat.color);
                    }
                } else {
                    float tp = FLT_MAX;
                    Ray jr; jr.o = org; jr.d = J.d; jr.inv = J.inv;
                    if (!LEAN && P.floor_on) tp = plane_hit(ld3(P.floor.norm), ld3(P.floor.point), jr, eps, FLT_MAX);   // kernels.cu:341-345, re-enabled by rt_render_options.floor
                    if (tp < FLT_MAX) {
                        obj = 2; t_hit = tp; normal = ld3(P.floor.norm);
                    } else if (specular && sphere_hit(lightC, lightR, jr, eps, FLT_MAX) < FLT_MAX) {             // kernels.cu:346
                        obj = 3;
                    }
                }
                if (obj == 0) {
                    if (primary) stat(RT_STAT_PRIMARY_NOHITS); else stat(from_mesh ? RT_STAT_SECONDARY_MESH_NOHIT : RT_STAT_SECONDARY_NOHIT);
                    pcolor = pcolor + atten * sky_color(P.sky, dir);                                            // kernels.cu:419-425
                    path_done = true;
                } else {
                    from_mesh = obj == 1;                                                                       // kernels.cu:428-432
                    if (primary) stat(from_mesh ? RT_STAT_PRIMARY_HIT_MESH : RT_STAT_PRIMARY_NOHITS);
                    if (obj == 3) {
                        if (!P.nee) pcolor = pcolor + atten * ld3(P.lightColor);                                // kernels.cu:440-446
                        path_done = true;
                    } else {
                        if (dot(J.d, normal) > 0.0f) normal = -normal;                                          // kernels.cu:354-355
                        Scatter sc;
                        material_scatter<LEAN>(sc, t_hit, org + t_hit * J.d, normal, inside, dir, mtype, albedo, mparam, rng);
                        org = org + sc.t * dir;                      // kernels.cu:485-489
                        dir = sc.wi;
                        atten = atten * sc.throughput;
                        specular = sc.specular;
                        inside = sc.refracted ? !inside : inside;

                        bool shadow_job = false;
                        if (P.nee && !specular) {                    // generateShadowRay, kernels.cu:363-393 (rt_device.h)
                            ShadowSample sh;
                            if (generate_shadow_ray(lightC, lightR, ld3(P.lightColor), org, atten, normal, rng, sh)) {
                                pend_contrib = sh.contrib;
                                pend_dist = sh.dist;
                                nshadow++;
                                stat(RT_STAT_SHADOWS);
                                shadow_dir = sh.dir;
                                want_job = 2;                        // hit(context, p, lightDist, true, ...)
                                shadow_job = true;
                            }
                        }
                        if (!shadow_job) next_ray = true;            // falls through to Russian roulette below
                    }
                }
            } else {
                // shadow traversal finished: hitMesh returned J.closest; "hit" means closest < lightDist (kernels.cu:331,504-510)
                if (!(J.closest < pend_dist)) { pcolor = pcolor + pend_contrib; stat(RT_STAT_SHADOWS_NOHITS); }
                J.shadow = false;
                next_ray = true;
            }
            if (next_ray) {
                if (P.rr && bounce > 3) {                            // kernels.cu:512-527
                    const float mx = max3(atten);
                    if (rnd(rng) > mx) {
                        stat(RT_STAT_RUSSIAN_KILL);
                        path_done = true;
                    } else {
                        const float kk = 1.0f / mx;
                        atten = F3(atten.x * kk, atten.y * kk, atten.z * kk);
                    }
                }
                if (!path_done) {
                    bounce++;
                    if (bounce >= P.max_depth) { stat(RT_STAT_EXCEED_MAX_BOUNCE); path_done = true; }          // loop bound, kernels.cu:402,529-531
                }
                if (!path_done) {
                    nrays++;
                    stat(RT_STAT_SECONDARY);                                                                    // kernels.cu:403-408
                    if (from_mesh) stat(RT_STAT_SECONDARY_MESH);
                    if (STATS && len(atten) < 0.01f) stat(RT_STAT_LOW_POWER);
                    want_job = 1;
                }
            }
            if (STATS && path_done && (isnan(pcolor.x) || isnan(pcolor.y) || isnan(pcolor.z))) stat(RT_STAT_NAN);   // kernels.cu:559-561
            if (path_done) {
                col = col + pcolor;                                  // kernels.cu:558
                s++;
                if (s < s_end) {
                    need_sample = true;
                } else if (PHASE == 1) {                             // first samples done: park the pixel (colour sum, stream position, cost)
                    const size_t px = (size_t)lr * P.nx + pi;
                    P.px_state[px] = make_float4(col.x, col.y, col.z, __uint_as_float(rng));
                    P.px_rays[px] = (pix_jobs + 12u) / 24u;         // in the ordering pass's unit: ~150 node visits per sample are ~6 "rays" per sample (class thresholds of cost_class)
                    have_pixel = false;
                } else {
                    const f3 out = col / (float)P.ns;                // kernels.cu:568
                    // one 12-byte store (global_store_dwordx3): a lane finishes its pixel on its own, three dword stores are three partial-sector writes
                    *reinterpret_cast<float3*>(fbf + ((size_t)lr * P.nx + pi) * 3) = make_float3(out.x, out.y, out.z);
                    have_pixel = false;
#if RT_MESH_TAIL_DIAG
                    if (PHASE == 2) g_diag_items[((size_t)lr * P.nx + pi) * 2 + 1] = (uint32_t)(__builtin_amdgcn_s_memrealtime() / 100ull);
#endif
                }
            }
        }

        if (dbg) { const unsigned long long c1 = __builtin_amdgcn_s_memtime(); g_cyc[0] += c1 - c0; c0 = c1; }
        // ================= refill idle lanes from the global pixel queue ======================================
        if (PHASE == 2 && chain_ask) {
            chain_ask = false;
            uint32_t c = 0;
            if ((threadIdx.x & 63) == 0) c = atomicAdd(Qown + 1, (uint32_t)chain_lanes);
            c = __builtin_amdgcn_readfirstlane(c);
            if (c < n_chain) {
                chain_wave = true;
                leaf_thr = 1; min_traversing = 63;
                const uint32_t l = threadIdx.x & 63u;
                if (l < (uint32_t)chain_lanes && c + l < n_chain) resume_pixel(seg_own + c + l);
            }
        }
        if (PHASE == 2 && chain_wave) {
            if (__ballot(have_pixel) != 0ull) goto refilled;         // a chain wave takes nothing while a chain runs
            chain_wave = false; leaf_thr = leaf_thr0; min_traversing = min_traversing0;
        }
        while (!exhausted) {
            const unsigned long long need = __ballot(!have_pixel);
            if (need == 0ull) break;
            const uint32_t cnt = (uint32_t)__popcll(need);
            uint32_t base = 0;
            if ((threadIdx.x & 63) == 0) base = atomicAdd(Qown, cnt);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= n_items) { exhausted = true; break; }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            if (base + cnt >= n_items) exhausted = true;
#if RT_MESH_TAIL_DIAG
            if (PHASE == 2 && exhausted && (threadIdx.x & 63) == 0) atomicMin(reinterpret_cast<unsigned long long*>(P.queue + 48), __builtin_amdgcn_s_memrealtime());      // the queue ran empty
#endif
            if (!have_pixel && base + rank < n_items) {
                if (PHASE == 2) {                                    // resume: the pixel's stream continues where the first dispatch left it
                    uint32_t pos = (stride == 0xFFFFFFFFu) ? n_items - 1u - (base + rank) : base + rank;      // (stride ~0: experiment, cheapest first)
                    const uint32_t nH = s_m[0], S = s_m[1];
                    if (nH != 0u && stride != 0xFFFFFFFFu) {
                        if (pos < S) {
                            const uint32_t before = (uint32_t)(((unsigned long long)pos * nH) / S);
                            const uint32_t after = (uint32_t)(((unsigned long long)(pos + 1u) * nH) / S);
                            pos = after > before ? before : nH + (pos - before);
                        }                                            // (pos >= S: every expensive pixel is out: the rest list at nH + (pos - nH) = pos)
                    }
                    resume_pixel(seg_own + n_chain + pos);
                } else {
                const uint32_t it = base + rank;
                const uint32_t p = p1seg ? (((uint32_t)(((unsigned long long)(it >> 3) * stride) % (total >> 3)) << 3) | (it & 7u))
                                         : (uint32_t)(((unsigned long long)it * stride) % total);
                const uint32_t tile = p >> 6, within = p & 63u;
                const int ty = (int)(tile / (uint32_t)tiles_x), tx = (int)(tile - (uint32_t)ty * (uint32_t)tiles_x);
                const int i = tx * 8 + (int)(within & 7u);
                lr = ty * 8 + (int)(within >> 3);
                if (i < P.nx && lr < P.part.local_rows) {
                    pi = i; pj = global_row(P.part, lr);
                    pixelId = (uint32_t)(pj * P.nx + pi);
                    rng = pixel_seed(pixelId);
                    col = F3(0, 0, 0);
                    s = 0;
                    pix_jobs = 0;
                    have_pixel = true;
                    need_sample = true;
                }
                }
            }
        }
        // (a queue per XCD) this wave's own queue is empty and lanes are idle: what the other stages have left - chain pixels nobody took (an XCD without a
        // workgroup of this grid has nobody to take them), then the next XCD's general part, its chain pixels, ... - as ordinary pixels
        while (PHASE == 2 && xq > 1u && exhausted && stolen + 1u < 2u * xq) {
            const unsigned long long need = __ballot(!have_pixel);
            if (need == 0ull) break;
            const uint32_t cnt = (uint32_t)__popcll(need);
            const uint32_t stage = stolen + 1u;
            const uint32_t qx = (myx + (stage >> 1)) & (uint32_t)(kXcdQueues - 1);
            const bool leftovers = (stage & 1u) != 0u;
            const uint32_t* const sm = s_mx + 8 * qx;
            const uint32_t limit = leftovers ? sm[2] : sm[5];
            uint32_t base = 0;
            if ((threadIdx.x & 63) == 0) base = atomicAdd(P.queue + (size_t)qx * kXcdQueueWords + (leftovers ? 1 : 0), cnt);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= limit) { stolen = stage; continue; }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            if (!have_pixel && base + rank < limit) {
                uint32_t pos = base + rank;
                if (!leftovers) {
                    const uint32_t nH = sm[0], S = sm[1];
                    if (nH != 0u && pos < S) {
                        const uint32_t before = (uint32_t)(((unsigned long long)pos * nH) / S);
                        const uint32_t after = (uint32_t)(((unsigned long long)(pos + 1u) * nH) / S);
                        pos = after > before ? before : nH + (pos - before);
                    }
                    pos += sm[2];
                }
                resume_pixel(sm[4] + pos);
            }
        }
    refilled:
        if (__ballot(have_pixel) == 0ull) break;
        if (need_sample) { start_sample(); need_sample = false; }    // one site for "path ended" and "new pixel"
        if (want_job) {
            const bool sh = want_job == 2;
            job_start(P, J, org, sh ? shadow_dir : dir, sh ? pend_dist : FLT_MAX, sh);
            if (STATS && J.idx == 0) stat(sh ? RT_STAT_SHADOWS_BBOX_NOHITS : (bounce == 0 ? RT_STAT_PRIMARY_BBOX_NOHITS : RT_STAT_SECONDARY_BBOX_NOHIT));   // kernels.cu:298-301
            want_job = 0;
        }
        if (dbg) { const unsigned long long c1 = __builtin_amdgcn_s_memtime(); g_cyc[1] += c1 - c0; c0 = c1; }

        // ================= TRAVERSE ================================================================================
        // at least once; keep going while enough lanes have nodes left (the others wait for the next PROCESS phase)
        if (TRAV == 0) {
            // Thresholded while-while.  One wave step is EITHER a node step (kernels.cu:162-195) for the lanes at an
            // internal node OR the leaf loop (kernels.cu:196-214) for the lanes at a leaf.  Classic while-while runs node
            // steps until EVERY lane sits on a leaf: measured 17.7 of 64 lanes busy per node step (tools/mesh_debug.py),
            // because descents between two leaves take 1..30 steps.  Here the leaf loop runs as soon as `leaf_thr` lanes
            // wait at a leaf (or no lane is at a node): lanes waiting for the slowest descent are capped at leaf_thr.
            // Every ray still sees the reference's own sequence of node visits and triangle tests.
            // A lane without a pixel has J.idx == 0 (a pixel ends in PROCESS, where J.idx is 0, and nothing starts a job for it), so the
            // lane masks of a step are two compares of J.idx; `have_pixel` does not change inside the loop.
            const int n_have = (int)__popcll(__builtin_amdgcn_ballot_w64(have_pixel));
            bool first = true;
            for (;;) {
                // ---- node steps (kernels.cu:162-195) for the lanes at an internal node, until the leaf phase is due or the loop is left
                unsigned long long act_m, node_m;
                bool leave = false;
                int n_leaf = 0;
                for (;;) {
                    act_m = __builtin_amdgcn_ballot_w64(J.idx != 0);
                    // back to PROCESS when enough lanes have a finished traversal to consume (or nothing is left to traverse); at least one
                    // step; lanes without a pixel (end of the frame) do not count, so the tail does not bounce between the phases
                    if (act_m == 0ull || (!first && n_have - (int)__popcll(act_m) >= 64 - min_traversing)) { leave = true; break; }
                    first = false;
                    node_m = act_m & __builtin_amdgcn_ballot_w64((uint32_t)J.idx < P.first_leaf);
                    const int n_node = (int)__popcll(node_m);
                    n_leaf = (int)__popcll(act_m & ~node_m);
                    if (n_node == 0 || n_leaf >= leaf_thr) break;
                    if (dbg) { g_act[2] += (unsigned long long)n_node; g_it[2]++; }
                    const bool at_node = J.idx != 0 && (uint32_t)J.idx < P.first_leaf;
                    if (at_node) {
                        if (PHASE == 1) pix_jobs++;                  // the pixel's measured cost: node visits (rays differ little between pixels, their traversals a lot)
                        const int idx2 = J.idx << 1;
                        // child pair of node idx from its axis-grouped record (rt_params.h): per axis (near_L, near_R, far_L, far_R)
                        const char* rec = reinterpret_cast<const char*>(P.bvh_axis);
                        const uint32_t rb = __umul24((uint32_t)J.idx, 96u);
                        const f4u px = *reinterpret_cast<const f4u*>(rec + (rb + J.ax));
                        const f4u py = *reinterpret_cast<const f4u*>(rec + (rb + J.ay));
                        const f4u pz = *reinterpret_cast<const f4u*>(rec + (rb + J.az));
                        if (COUNT) st.nodes++;
                        // leftHit / rightHit of kernels.cu:175-181 are (hit ? entry : FLT_MAX); only compared, so kept as (hit, entry):
                        // `x < closest` = hit && entry < closest (closest <= FLT_MAX), `rightHit < leftHit` = as below when one side is taken.
                        // hit_bbox_dist (intersections.h:25-41) per box: t_min from 0.001f, t_max from closest, v_max / v_min updates (rt_device.h slab)
                        float le = 0.001f, re = 0.001f, lx = J.closest, rx = J.closest;
                        le = fmaxf((px.x - org.x) * J.inv.x, le); lx = fminf((px.z - org.x) * J.inv.x, lx);
                        re = fmaxf((px.y - org.x) * J.inv.x, re); rx = fminf((px.w - org.x) * J.inv.x, rx);
                        le = fmaxf((py.x - org.y) * J.inv.y, le); lx = fminf((py.z - org.y) * J.inv.y, lx);
                        re = fmaxf((py.y - org.y) * J.inv.y, re); rx = fminf((py.w - org.y) * J.inv.y, rx);
                        le = fmaxf((pz.x - org.z) * J.inv.z, le); lx = fminf((pz.z - org.z) * J.inv.z, lx);
                        re = fmaxf((pz.y - org.z) * J.inv.z, re); rx = fminf((pz.w - org.z) * J.inv.z, rx);
                        const bool hl = !(lx < le), hr = !(rx < re);
                        const bool traverseLeft = hl && le < J.closest;
                        const bool traverseRight = hr && re < J.closest;
                        const bool swap = traverseRight && (!traverseLeft || re < le);
                        if (traverseLeft || traverseRight) {
                            if (STATS) stat((traverseLeft && traverseRight) ? RT_STAT_NODES_BOTH : RT_STAT_NODES_SINGLE);   // BVH_COUNT, kernels.cu:184-191
                            J.idx = idx2 + (swap ? 1 : 0);
                            J.bitStack = (J.bitStack << 1) + ((traverseLeft && traverseRight) ? 1u : 0u);
                        } else {
                            const int m = __ffs((int)J.bitStack) - 1;    // pop_bitstack, kernels.cu:148-152
                            J.bitStack = (J.bitStack >> m) ^ 1u;
                            J.idx = (J.idx >> m) ^ 1;
                        }
                    }
                    if (dbg) { const unsigned long long c1 = __builtin_amdgcn_s_memtime(); g_cyc[2] += c1 - c0; c0 = c1; }
                }
                if (leave) break;
                const bool act = J.idx != 0;
                const bool at_node = act && (uint32_t)J.idx < P.first_leaf;
                {
                    if (dbg) { g_act[3] += (unsigned long long)n_leaf; g_it[3]++; }
                    if (pair_per > 0) {
                        // (ray, triangle) pairs: the <= leaf_thr lanes at a leaf have nppl triangles each; tested one triangle
                        // per lane and nppl steps they would keep n_leaf of 64 lanes busy.  Instead lane j of a round takes
                        // triangle j % nppl of the (j / nppl)-th waiting ray: ray fetched with ds_bpermute, result folded into
                        // the owner's slot with one 64-bit LDS atomicMin.  The reference's loop (kernels.cu:196-214) returns the
                        // lexicographic minimum of (t, k) over the triangles before the first sentinel that hit below `closest`
                        // (its running t_max only rejects what the minimum rejects too); a shadow ray stops at the FIRST k that
                        // hits.  Both are what min(key) below is, key = t_bits << 32 | k (t > 0) resp. k for shadow rays.
                        const bool at_leaf = act && !at_node;
                        const unsigned long long leaf_m = act_m & ~node_m;
                        const uint32_t my_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(leaf_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)leaf_m, 0u));
                        if (at_leaf) { w_owner[my_rank] = lane; w_best[lane] = ~0ull; }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t packed = (uint32_t)J.idx | (J.shadow ? 0x80000000u : 0u);
                        // only FULL rounds (pair_per rays x nppl triangles fill the wave); the remaining owners keep waiting at their leaf and
                        // are served by the next leaf phase - a second round with a handful of pairs costs as much as a full one
                        // (16 owners x 5 triangles = 64 + 16 pairs was the common case).  Fewer owners than one round: all of them.
                        const int n_serve = n_leaf >= pair_per ? (n_leaf / pair_per) * pair_per : n_leaf;
                        for (int base = 0; base < n_serve; base += pair_per) {
                            const int r = base + (int)pair_r;
                            const bool pv = pair_ok && r < n_serve;
                            const uint32_t owner = pv ? w_owner[r] : lane;
                            const int src = (int)(owner << 2);
                            const float ox = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(org.x)));
                            const float oy = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(org.y)));
                            const float oz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(org.z)));
                            const float dx = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(J.d.x)));
                            const float dy = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(J.d.y)));
                            const float dz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(J.d.z)));
                            const float o_closest = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(J.closest)));
                            const uint32_t o_packed = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)packed);
                            const bool o_shadow = (o_packed & 0x80000000u) != 0u;
                            bool reached = false, hit = false;
                            float u = 0.0f, v = 0.0f;
                            unsigned long long key = ~0ull;
                            {
                                const uint32_t o_leaf = (o_packed & 0x7FFFFFFFu) - P.first_leaf;
                                reached = pv && pair_k < (uint32_t)s_leaf_cnt[pv ? o_leaf : 0u];     // kernels.cu:202: the loop ends at the leaf's first sentinel = after its real triangles
                                if (reached) {
                                    const float4* pt = P.leaf_tri + (size_t)(o_leaf * P.nppl + pair_k) * 3;
                                    const float4 ta = pt[0], tb = pt[1];
                                    const float tcx = pt[2].x;
                                    Ray pr;
                                    pr.o = F3(ox, oy, oz); pr.d = F3(dx, dy, dz); pr.inv = F3(0, 0, 0);
                                    const float hitT = triangle_hit_edges(F3(ta.x, ta.y, ta.z), F3(ta.w, tb.x, tb.y), F3(tb.z, tb.w, tcx), pr, eps, o_closest, u, v);
                                    hit = hitT < o_closest;
                                    if (hit) {
                                        key = o_shadow ? (unsigned long long)pair_k
                                                       : (((unsigned long long)__float_as_uint(hitT) << 32) | (unsigned long long)pair_k);
                                        atomicMin(&w_best[owner], key);
                                    }
                                }
                            }
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            if (pv) {
                                const unsigned long long best = w_best[owner];
                                // tests the reference executes: every reached triangle; a shadow ray stops after its first hit
                                if (COUNT && reached && (!o_shadow || best == ~0ull || (unsigned long long)pair_k <= best)) st.tests++;
                                if (!LEAN && hit && key == best) w_uv[owner] = make_float2(u, v);
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        unsigned long long my_best = ~0ull;              // owner side: result of my leaf
                        float my_u = 0.0f, my_v = 0.0f;
                        const bool served = at_leaf && (int)my_rank < n_serve;
                        if (served) {
                            my_best = w_best[lane];
                            if (!LEAN) { const float2 uv = w_uv[lane]; my_u = uv.x; my_v = uv.y; }
                        }
                        if (served) {
                            if (my_best != ~0ull && J.shadow) {          // any-hit: hitBvh returns 0.0f (kernels.cu:205)
                                J.closest = 0.0f;
                                J.idx = 0;
                            } else {
                                if (my_best != ~0ull) {
                                    J.closest = __uint_as_float((uint32_t)(my_best >> 32));
                                    J.triId = ((uint32_t)J.idx - P.first_leaf) * P.nppl + (uint32_t)(my_best & 0xFFFFFFFFull);
                                    if (!LEAN) { J.hu = my_u; J.hv = my_v; }
                                }
                                const int m = __ffs((int)J.bitStack) - 1;
                                J.bitStack = (J.bitStack >> m) ^ 1u;
                                J.idx = (J.idx >> m) ^ 1;
                            }
                        }
                    } else
                    if (act && !at_node) {
                        const uint32_t first = ((uint32_t)J.idx - P.first_leaf) * P.nppl;
                        bool occluded = false;
                        for (uint32_t k = 0; k < P.nppl; k++) {
                            const float4* pt = reinterpret_cast<const float4*>(P.tris + first + k);
                            const float4 a = pt[0], b = pt[1];
                            if (isinf(a.x)) break;                       // kernels.cu:202 sentinel
                            const float cx = pt[2].x;
                            float u, v;
                            if (COUNT) st.tests++;
                            const float hitT = triangle_hit(F3(a.x, a.y, a.z), F3(a.w, b.x, b.y), F3(b.z, b.w, cx), Ray{ org, J.d, J.inv }, eps, J.closest, u, v);
                            if (hitT < J.closest) {
                                if (J.shadow) { occluded = true; break; }    // any-hit: hitBvh returns 0.0f (kernels.cu:205)
                                J.closest = hitT;
                                J.triId = first + k;
                                J.hu = u; J.hv = v;
                            }
                        }
                        if (occluded) {
                            J.closest = 0.0f;
                            J.idx = 0;
                        } else {
                            const int m = __ffs((int)J.bitStack) - 1;
                            J.bitStack = (J.bitStack >> m) ^ 1u;
                            J.idx = (J.idx >> m) ^ 1;
                        }
                    }
                    if (dbg) { const unsigned long long c1 = __builtin_amdgcn_s_memtime(); g_cyc[3] += c1 - c0; c0 = c1; }
                }
            }
        } else {
            do {
                // descend: internal nodes (kernels.cu:162-195)
                for (;;) {
                    const bool at_node = have_pixel && J.idx != 0 && (uint32_t)J.idx < P.first_leaf;
                    const unsigned long long node_m = __ballot(at_node);
                    if (node_m == 0ull) break;
                    if (dbg) { g_act[2] += (unsigned long long)__popcll(node_m); g_it[2]++; }
                    if (!at_node) continue;
                    const int idx2 = J.idx << 1;
                    const float4* n = P.bvh4 + (size_t)J.idx * 3;
                    const float4 na = n[0], nb = n[1], nc = n[2];
                    st.nodes++;
                    const float leftHit = hit_bbox_dist(F3(na.x, na.y, na.z), F3(na.w, nb.x, nb.y), Ray{ org, J.d, J.inv }, J.closest);
                    const bool traverseLeft = leftHit < J.closest;
                    const float rightHit = hit_bbox_dist(F3(nb.z, nb.w, nc.x), F3(nc.y, nc.z, nc.w), Ray{ org, J.d, J.inv }, J.closest);
                    const bool traverseRight = rightHit < J.closest;
                    const bool swap = rightHit < leftHit;
                    if (traverseLeft && traverseRight) {
                        stat(RT_STAT_NODES_BOTH);
                        J.idx = idx2 + (swap ? 1 : 0);
                        J.bitStack = (J.bitStack << 1) + 1;
                    } else if (traverseLeft || traverseRight) {
                        stat(RT_STAT_NODES_SINGLE);
                        J.idx = idx2 + (swap ? 1 : 0);
                        J.bitStack = J.bitStack << 1;
                    } else {
                        const int m = __ffs((int)J.bitStack) - 1;        // pop_bitstack, kernels.cu:148-152
                        J.bitStack = (J.bitStack >> m) ^ 1u;
                        J.idx = (J.idx >> m) ^ 1;
                    }
                }
                if (dbg) { const unsigned long long c1 = __builtin_amdgcn_s_memtime(); g_cyc[2] += c1 - c0; c0 = c1; }
                if (dbg) { g_act[3] += (unsigned long long)__popcll(__ballot(have_pixel && J.idx != 0)); g_it[3]++; }
                // leaf (kernels.cu:196-214)
                if (have_pixel && J.idx != 0) {
                    const uint32_t first = ((uint32_t)J.idx - P.first_leaf) * P.nppl;
                    bool occluded = false;
                    for (uint32_t k = 0; k < P.nppl; k++) {
                        const float4* pt = reinterpret_cast<const float4*>(P.tris + first + k);
                        const float4 a = pt[0], b = pt[1];
                        if (isinf(a.x)) break;                           // kernels.cu:202 sentinel
                        const float cx = pt[2].x;
                        float u, v;
                        st.tests++;
                        const float hitT = triangle_hit(F3(a.x, a.y, a.z), F3(a.w, b.x, b.y), F3(b.z, b.w, cx), Ray{ org, J.d, J.inv }, eps, J.closest, u, v);
                        if (hitT < J.closest) {
                            if (J.shadow) { occluded = true; break; }    // any-hit: hitBvh returns 0.0f (kernels.cu:205)
                            J.closest = hitT;
                            J.triId = first + k;
                            J.hu = u; J.hv = v;
                        }
                    }
                    if (occluded) {
                        J.closest = 0.0f;
                        J.idx = 0;
                    } else {
                        const int m = __ffs((int)J.bitStack) - 1;
                        J.bitStack = (J.bitStack >> m) ^ 1u;
                        J.idx = (J.idx >> m) ^ 1;
                    }
                }
                if (dbg) { const unsigned long long c1 = __builtin_amdgcn_s_memtime(); g_cyc[3] += c1 - c0; c0 = c1; }
            } while (__popcll(__ballot(have_pixel && J.idx != 0)) >= min_traversing);
        }
    }
#if RT_MESH_TAIL_DIAG
    if (PHASE == 2 && (threadIdx.x & 63) == 0) {                    // (diagnostic build: when did the queue run empty, when did the last wave end: tools/build_variant.sh -DRT_MESH_TAIL_DIAG=1)
        atomicMin(reinterpret_cast<unsigned long long*>(P.queue + 50), t_start_diag);
        atomicMax(reinterpret_cast<unsigned long long*>(P.queue + 52), __builtin_amdgcn_s_memrealtime());
        atomicAdd(reinterpret_cast<unsigned long long*>(P.queue + 54), __builtin_amdgcn_s_memrealtime() - t_start_diag);      // wave-time
    }
#endif
    if (dbg && (threadIdx.x & 63) == 0) {
        for (int k = 0; k < 4; k++) { atomicAdd(P.dbg + k, g_cyc[k]); atomicAdd(P.dbg + 4 + k, g_act[k]); atomicAdd(P.dbg + 8 + k, g_it[k]); }
        atomicAdd(P.dbg + 12, 1ull);
    }

    if (P.counters) {
        atomicAdd(&P.counters->rays, (unsigned long long)nrays);
        atomicAdd(&P.counters->shadow_rays, (unsigned long long)nshadow);
        atomicAdd(&P.counters->prim_tests, (unsigned long long)st.tests);
        atomicAdd(&P.counters->node_visits, (unsigned long long)st.nodes);
    }
}


}  // namespace

// variant: bits 0..7  0 = persistent state-machine kernel (default), 1 = first kernel (one tile per wave);
//          bits 8..15 workgroups per CU of the persistent kernel (0 = default 4);
//          bits 16..23 keep traversing while at least this many lanes have nodes left (0 = default: 24, classic 40);
//          bits 24..25 traversal of the persistent kernel: 0 = thresholded while-while (default), 1 = classic while-while;
//          bits 26..31 leaf threshold of the former (0 = default 16).
hipError_t RT_LAUNCH_NAME(const RtMeshParams& p, int variant, hipStream_t stream) {
    if ((variant & 0xFF) == 1) {
        const dim3 grid((p.nx + 8 * kWavesPerWg - 1) / (8 * kWavesPerWg), (p.part.local_rows + 7) / 8);
        hipLaunchKernelGGL(k_render_mesh<0>, grid, dim3(kThreads), 0, stream, p);
        return hipGetLastError();
    }
    if (!p.queue) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(p.queue, 0, 64, stream);
    if (e != hipSuccess) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int wg_per_cu = (variant >> 8) & 0xFF;
    const bool classic0 = ((variant >> 24) & 3) == 1;
    static const bool lean_env = !(getenv("RT_MESH_LEAN") && getenv("RT_MESH_LEAN")[0] == '0');          // A/B: RT_MESH_LEAN=0 keeps the general kernel
    const bool lean = lean_env && !classic0 && !p.dbg && !p.counters && p.lean_ok && !p.floor_on && p.leaf_sentinels_trailing && p.leaf_ofs && p.leaf_tri &&
                      p.nppl >= 1u && p.nppl <= 16u && p.first_leaf <= kLeafCntLds;
    if (wg_per_cu == 0) wg_per_cu = lean ? RT_MESH_LEAN_WAVES : (classic0 ? 5 : 4);   // = the launch bounds (96 / 128 VGPRs); the pair rounds spill at 96      // launch bound: 5 waves per SIMD (96 VGPRs); 6 gave the same rate, 8 spills
    const long long total_px = (long long)((p.nx + 7) / 8) * ((p.part.local_rows + 7) / 8) * 64;
    long long blocks = (long long)cus * wg_per_cu * 4 / kWavesPerWg;
    const long long useful = (total_px + kThreads - 1) / kThreads;
    if (blocks > useful) blocks = useful;
    if (blocks < 1) blocks = 1;
    uint32_t stride = 1;
    static const bool tile_order = getenv("RT_MESH_ORDER") && getenv("RT_MESH_ORDER")[0] == 't';       // A/B: tile-major pixel order (stride 1)
    if (total_px > 64 && !tile_order) {
        auto gcd = [](unsigned long long a, unsigned long long b) { while (b) { const unsigned long long t = a % b; a = b; b = t; } return a; };
        unsigned long long cand = (unsigned long long)((double)total_px * 0.6180339887) | 1ull;
        while (gcd(cand, (unsigned long long)total_px) != 1ull) cand += 2;
        stride = (uint32_t)(cand % (unsigned long long)total_px);
    }
    const bool classic = ((variant >> 24) & 3) == 1;
    int min_traversing = (variant >> 16) & 0xFF;
    if (min_traversing == 0) min_traversing = classic ? kMinTraversing : 24;    // measured: 16 -> 409, 20 -> 435, 24 -> 446, 32 -> 429 Msamples/s
    int leaf_thr = (variant >> 26) & 0x3F;
    // default leaf threshold = one full pair round: 64 / nppl waiting rays (12 at 5 triangles per leaf).  Measured on C4 with full rounds
    // only: 8 -> 607, 12 -> 648, 16 -> 616, 24 -> 541 Msamples/s (round 1, with partial rounds: 16 -> 610).
    if (leaf_thr == 0) leaf_thr = (p.nppl >= 1u && p.nppl <= 16u) ? (int)(64u / p.nppl) : 16;
    const dim3 grid((unsigned)blocks), block(kThreads);
    const size_t lds = (!classic && p.leaf_ofs && p.first_leaf <= kLeafCntLds) ? (size_t)((p.first_leaf + 15u) & ~15u) : 0;      // the leaf-count table
    // the counting instantiation (STATS: the reference's ray statistics as device atomics) runs only when counters are asked for
    // The cost-ordered frame in two dispatches (template parameter PHASE): reference RNG stream, no diagnostics, enough samples for the first few to be a small part.
    // RT_MESH_TWO=0: the single scattered dispatch (A/B); RT_MESH_SPLIT=<n>: samples of the first dispatch.
    static const bool two_env = !(getenv("RT_MESH_TWO") && getenv("RT_MESH_TWO")[0] == '0');
    static const int split_env = getenv("RT_MESH_SPLIT") ? atoi(getenv("RT_MESH_SPLIT")) : 2;      // (1: 903, 2: 911, 4: 893, 8: 866 Msamples/s on C4, profiles/r04_ab_mesh_two_b.txt)
    const int split = split_env < 1 ? 1 : split_env;
    if (two_env && !classic && !p.dbg && !p.counters && p.rng_mode == RT_RNG_REFERENCE_STREAM && p.px_state && p.px_rays && p.order && p.ord_state && p.ord_rays &&
        p.ns >= 4 * split && p.nx <= 65535 && p.part.local_rows <= 65535) {
        RtMeshParams q = p;
        q.s_split = split;
        // (p.p1_segments: the first dispatch scatters row segments of 8 pixels - `stride` coprime with total / 8, flag in bit 8 of min_traversing)
        uint32_t stride1 = stride;
        int mt1 = min_traversing;
        if (p.p1_segments && total_px > 512 && !tile_order) {
            auto gcd = [](unsigned long long a, unsigned long long b) { while (b) { const unsigned long long t = a % b; a = b; b = t; } return a; };
            const unsigned long long segs = (unsigned long long)total_px >> 3;
            unsigned long long cand = (unsigned long long)((double)segs * 0.6180339887) | 1ull;
            while (gcd(cand, segs) != 1ull) cand += 2;
            stride1 = (uint32_t)(cand % segs);
            mt1 |= 256;
        }
        if (lean) hipLaunchKernelGGL((k_render_mesh_queue<0, false, false, true, 1>), grid, block, lds, stream, q, stride1, mt1, leaf_thr);
        else hipLaunchKernelGGL((k_render_mesh_queue<0, false, false, false, 1>), grid, block, lds, stream, q, stride1, mt1, leaf_thr);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(p.queue, 0, sizeof(uint32_t) * kXcdQueues * kXcdQueueWords, stream);
        if (e != hipSuccess) return e;
        RtSphereParams o;                                            // what the ordering pass reads (rt_params.h: rt_order_pixels_by_cost)
        memset(&o, 0, sizeof o);
        const int chain_thr_env = getenv("RT_MESH_CHAIN_THR") ? atoi(getenv("RT_MESH_CHAIN_THR")) : RT_MESH_CHAIN_THR;
        const int chain_lanes_env = getenv("RT_MESH_CHAIN_LANES") ? atoi(getenv("RT_MESH_CHAIN_LANES")) : RT_MESH_CHAIN_LANES;
        o.nx = p.nx; o.ny = p.ny; o.part = p.part; o.s_split = split; o.chain_top_thr = chain_thr_env < 17 ? 17 : chain_thr_env;
        o.px_rays = p.px_rays; o.px_state = p.px_state; o.order = p.order; o.ord_state = p.ord_state; o.ord_rays = p.ord_rays; o.queue = p.queue;
        o.ord_rec = p.ord_rec; o.xcd_queues = p.xcd_queues;
        e = rt_order_pixels_by_cost(o, stream);
        if (e != hipSuccess) return e;
        static const bool rev_env = getenv("RT_MESH_REV") && getenv("RT_MESH_REV")[0] == '1';       // experiment: cheapest pixels first
        static const int heavy_env = getenv("RT_MESH_HEAVY") ? atoi(getenv("RT_MESH_HEAVY")) : RT_MESH_HEAVY_CLS;       // expensive classes spread over the first fills
        static const int rounds_env = getenv("RT_MESH_ROUNDS") ? atoi(getenv("RT_MESH_ROUNDS")) : RT_MESH_SPREAD_ROUNDS;
        const uint32_t stride2 = rev_env ? 0xFFFFFFFFu : stride;
        const int chain_frac = getenv("RT_MESH_CHAIN_FRAC") ? (atoi(getenv("RT_MESH_CHAIN_FRAC")) & 0xF) : 8;     // chain waves only while list 0 is below pixels >> this (tests: 0)
        const int lt2 = leaf_thr | ((heavy_env & 0xF) << 8) | ((rounds_env & 0xF) << 12) | ((chain_lanes_env < 0 ? 0 : chain_lanes_env > 64 ? 64 : chain_lanes_env) << 16) | (chain_frac << 24);
#if RT_MESH_TAIL_DIAG
        static uint32_t* d_items = nullptr; const size_t n_px = (size_t)p.part.local_rows * p.nx;
        if (!d_items) { (void)hipMalloc(&d_items, n_px * 8); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_diag_items), &d_items, sizeof d_items); }
        { const unsigned long long init[4] = { ~0ull, ~0ull, 0ull, 0ull }; (void)hipMemcpyAsync(p.queue + 48, init, sizeof init, hipMemcpyHostToDevice, stream); (void)hipStreamSynchronize(stream); }
#endif
        if (lean) hipLaunchKernelGGL((k_render_mesh_queue<0, false, false, true, 2>), grid, block, lds, stream, q, stride2, min_traversing, lt2);
        else hipLaunchKernelGGL((k_render_mesh_queue<0, false, false, false, 2>), grid, block, lds, stream, q, stride2, min_traversing, lt2);
#if RT_MESH_TAIL_DIAG
        { unsigned long long r[4]; (void)hipStreamSynchronize(stream); (void)hipMemcpy(r, p.queue + 48, sizeof r, hipMemcpyDeviceToHost);
          unsigned q4[2]; (void)hipMemcpy(q4, p.queue + 4, 4, hipMemcpyDeviceToHost); fprintf(stderr, "list 0: %u pixels; ", q4[0]);
          fprintf(stderr, "mesh tail diag: queue empty at %.1f ms, last wave ends at %.1f ms, mean wave life %.1f ms (%u waves)\n", (double)(r[0] - r[1]) * 1e-5, (double)(r[2] - r[1]) * 1e-5,
                  (double)r[3] * 1e-5 / ((double)grid.x * (block.x / 64)), grid.x * (block.x / 64));
          if (const char* f = getenv("RT_MESH_DIAG_FILE")) { std::vector<uint32_t> h(n_px * 3); (void)hipMemcpy(h.data(), d_items, n_px * 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(h.data() + n_px * 2, p.px_rays, n_px * 4, hipMemcpyDeviceToHost); if (FILE* o = fopen(f, "wb")) { fwrite(h.data(), 4, h.size(), o); fclose(o); } } }
#endif
        return hipGetLastError();
    }
    if (classic) {
        if (p.dbg) hipLaunchKernelGGL((k_render_mesh_queue<1, true, false>), grid, block, lds, stream, p, stride, min_traversing, leaf_thr);
        else if (p.counters) hipLaunchKernelGGL((k_render_mesh_queue<1, false, true>), grid, block, lds, stream, p, stride, min_traversing, leaf_thr);
        else hipLaunchKernelGGL((k_render_mesh_queue<1, false, false>), grid, block, lds, stream, p, stride, min_traversing, leaf_thr);
    } else {
        if (lean) hipLaunchKernelGGL((k_render_mesh_queue<0, false, false, true>), grid, block, lds, stream, p, stride, min_traversing, leaf_thr);
        else if (p.dbg) hipLaunchKernelGGL((k_render_mesh_queue<0, true, false>), grid, block, lds, stream, p, stride, min_traversing, leaf_thr);
        else if (p.counters) hipLaunchKernelGGL((k_render_mesh_queue<0, false, true>), grid, block, lds, stream, p, stride, min_traversing, leaf_thr);
        else hipLaunchKernelGGL((k_render_mesh_queue<0, false, false>), grid, block, lds, stream, p, stride, min_traversing, leaf_thr);
    }
    return hipGetLastError();
}
