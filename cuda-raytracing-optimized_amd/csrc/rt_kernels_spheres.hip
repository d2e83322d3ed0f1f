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
//   * the closest-hit scan is split in two phases.  Phase 1 FINDS the spheres whose discriminant is positive: fused multiply-adds
//     and a proven slack (12 VALU ops/sphere, no branch, no compare: the sign bit is shifted into a per-lane mask with one
//     v_alignbit; every sphere the reference's own arithmetic accepts is flagged).  Phase 2 runs the literal sphereHit
//     (every product and sum rounded, IEEE sqrt + divide) only for those candidates.
//     Because a non-candidate returns FLT_MAX in the reference and never updates `closest`, and
//     candidates are visited in increasing index with the same strict `<`, the result is
//     bit-identical to the reference's linear scan;
//   * sphere-group culling + pair compaction (scan_pairs): the host splits the small spheres at medians
//     into compact groups of 16 with a tight AABB each; a ray only visits the groups whose box it can reach before its
//     current closest hit (slab test widened by a per-ray margin that covers the reference's own discriminant rounding:
//     exact for any ray origin), and the (ray, group) pairs of a wave are compacted so that all 64 lanes always work.
//     A skipped sphere could only have produced t > closest (rejected by the reference) or FLT_MAX, so the result
//     is unchanged; the explicit (t, original index) tie rule keeps the reference's first-index-wins order;
//   * the group boxes are stored per AXIS as (lo, hi, lo): a ray reads (near, far) at an offset given by the sign of its direction; an axis on
//     which all boxes agree (spheres on a plane) is evaluated once per pass; results are accumulated as sign bits (box_gap, group_needs);
//   * scenes whose small spheres rest on a horizontal plane (the benchmark) run the box tests behind a cell-table prefilter (group_needs_cells): the part of the ray
//     inside the spheres' slab has a bounding rectangle, per-cell sets of "boxes that begin below / end above" leave the 2-3 boxes it can overlap, and only those
//     take the slab test - conservative on its own terms, so the pair set and the frame are unchanged;
//   * framebuffer: the persistent kernels (default) store a finished pixel with ONE 12-byte store from the lane that owns it (pixels finish one by
//     one, in cost order) into the compact DEVICE framebuffer, and the two-dispatch frame sees to it that whole lines leave the L2: the cost lists and
//     queue counters exist once per XCD, a pixel belongs to the XCD rt_xcd_of_pixel() names - the 32 pixels of three adjacent 128-byte lines to the same
//     one - and a wave serves the queue of the XCD it runs on (then the others'), so the write-back L2 of that XCD merges the 12-byte stores; the copy
//     engine delivers the rows to the host framebuffer (RT_FB_DIRECT=1: straight into the pinned host framebuffer instead, RtSphereParams::fb_global_rows -
//     2 % less wall time on a 11 ms frame, every store its own bus write).  The tile kernel (variant 1) transposes its 8x8 tile through LDS so that a
//     wave writes row-contiguous dwords;
//   * parameters needed once per sample (camera, image size) are re-read from a device copy of the parameter block (RtSphereParams::self);
//   * the scene copy of a workgroup comes in three forms (stage_scene, template parameter SCENE): everything in the LDS; the test data in the LDS and the
//     hit data in global memory (two workgroups per CU up to ~1500 spheres, LDS-resident up to ~5000); everything read from global memory beyond that.
#include "rt_device.h"
#include "rt_params.h"

#include <float.h>
#include <cstdio>
#include <cstdlib>

using namespace rtd;

#if defined(RT_MODE_PARITY)
#define RT_LAUNCH_NAME rt_launch_spheres_parity
#elif defined(RT_MODE_FAST)
#define RT_LAUNCH_NAME rt_launch_spheres_fast
#else
#error "define RT_MODE_PARITY or RT_MODE_FAST"
#endif

namespace {

#ifndef RT_MID_WAVES
#define RT_MID_WAVES 0      // middle tier: waves per workgroup (0 = off) and pixels per such wave
#define RT_MID_CAP 16
#endif
#ifndef RT_BOX_CUT
#define RT_BOX_CUT 12       // group_needs_cells: the per-lane box loop stops when fewer lanes than this still have candidates (A/B on C5 at 256 spp, profiles/r04_ab_basic_c5.txt:
                            // 1 (never): 10870, 8: 10930, 12: 10945, 20: 10945 Msamples/s)
#endif
constexpr int kWavesPerWg = 16;             // 16 waves - a whole CU at 4 waves per SIMD - share ONE LDS copy of the scene (tile kernel: 16 tiles side by side)
constexpr int kThreads = 64 * kWavesPerWg;

constexpr int kPassGroups = 16;
constexpr int kCandCap = 192;
__host__ __device__ constexpr int ws_cand_cap(bool onepass) { return onepass ? 128 : kCandCap; }
// Per-wave scratch: ray table (64 x 32 B), best-hit keys (64 x 8 B), the (ray, group) pair list (2 B per entry), the candidate list.  The pair list holds one pass
// of kPassGroups groups for every lane + the carried remainder; the one-pass kernels (scan_pairs<ONEPASS>: 32 groups behind the prefilter, ~70 pairs per ray batch)
// keep a list of 8 groups per lane + remainder - a batch that would overflow it is cut into quarters (wave-uniform, not seen on the benchmark) - which is
// what lets two 10-wave workgroups with a scene copy each share a CU's LDS (5 waves per SIMD).
__host__ __device__ constexpr int ws_list_cap(bool onepass) { return onepass ? 64 * 8 + 64 : 64 * kPassGroups + 64; }
__host__ __device__ constexpr int ws_pairs(bool onepass) { return 64 * 32 + 64 * 8 + ws_list_cap(onepass) * 2; }
__host__ __device__ constexpr int ws_total(bool onepass) { return ws_pairs(onepass) + ws_cand_cap(onepass) * 4 + 16; }
constexpr int kWaveScratch = ws_total(false);

__device__ __forceinline__ int global_row(const RtPartition& pt, int lr) {
    const int stripe = lr / pt.stripe_rows;
    return (stripe * pt.world + pt.rank) * pt.stripe_rows + (lr - stripe * pt.stripe_rows);
}

// All-ones (a NaN in every float) into the rows of the HOST framebuffer this partition member owns (P.fb with fb_global_rows: the whole image;
// the member's local row lr is global row global_row(lr)).  Called by every thread of a grid with its flat index `tid` of `nthreads`: a pixel the work
// distribution loses then shows as NaN instead of as the previous frame's value.  The frame's first dispatch does this beside its compute (it stores no
// pixel: 12 bus writes of 4 bytes per thread on the benchmark frame); a single-dispatch frame runs k_poison_fb in front of its kernel.
__device__ __attribute__((noinline)) void poison_words(uint32_t* fbw, int nx, RtPartition part, size_t tid, size_t nthreads) {
    const size_t row_words = (size_t)nx * 3, total = (size_t)part.local_rows * row_words;
    if (part.world == 1) {                                           // the member owns every row: one contiguous range
        for (size_t k = tid; k < total; k += nthreads) fbw[k] = 0xFFFFFFFFu;
        return;
    }
    for (size_t k = tid; k < total; k += nthreads) {
        const int lr = (int)(k / row_words);
        fbw[(size_t)global_row(part, lr) * row_words + (k - (size_t)lr * row_words)] = 0xFFFFFFFFu;
    }
}
// (a real function taking values: inlined into the persistent kernel's loop it cost that kernel 13 SGPR and 3 VGPR spills)
__device__ __forceinline__ void poison_rows(const RtSphereParams& P, size_t tid, size_t nthreads) {
    poison_words(reinterpret_cast<uint32_t*>(P.fb), P.nx, P.part, tid, nthreads);
}
__global__ void __launch_bounds__(256) k_poison_fb(const RtSphereParams P) {
    poison_rows(P, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}

// sphereHit, intersections.h:85-104, on a pre-normalised direction `dn` with a = dot(dn,dn) hoisted (same bits every
// call) and r2 = radius*radius precomputed (same bits).  `t_max` is INCLUSIVE here: the caller applies the reference's
// strict `t < closest` itself, extended by the first-index-wins tie rule (see accept()).
__device__ __forceinline__ float sphere_hit_exact(float4 s, f3 org, f3 dn, float a, float t_min, float t_max) {
    const f3 oc = org - F3(s.x, s.y, s.z);
    const float b = dot(oc, dn);
    const float c = dot(oc, oc) - s.w;
    const float discriminant = b * b - a * c;
    if (discriminant > 0) {
        const float sq = rt_sqrt(discriminant);
        float temp = (-b - sq) / a;
        if (temp <= t_max && temp > t_min) return temp;
        temp = (-b + sq) / a;
        if (temp <= t_max && temp > t_min) return temp;
    }
    return FLT_MAX;
}

struct Hit { float closest; int sid; int orig; };    // sid = slot in the (re-ordered) scene arrays, orig = caller's index

// The reference scans spheres in the caller's index order with a strict `t < closest`, i.e. it returns the
// lexicographic minimum of (t_k, k).  We scan in a different (spatially sorted) order, so the tie is explicit.
__device__ __forceinline__ void accept(Hit& h, float t, int slot, int orig) {
    if (t < h.closest || (t == h.closest && orig < h.orig)) { h.closest = t; h.sid = slot; h.orig = orig; }
}

// LDS image of the scene, staged once per workgroup (README.md:93-103 used __constant__).
__device__ __forceinline__ int sidx(int slot) { return slot + (slot >> kSphereGroupShift); }

struct SceneLds {
    const float4* sph;      // (cx, cy, cz, r*r) of slot k at index sidx(k) = k + k/16: 17 float4 per group of 16, so that
                            // lanes reading DIFFERENT groups in one ds_read_b128 fall on different banks (pair scan)
    const float4* grp;      // 3 x n_groups: per group and AXIS (lo, hi, lo, -) of the AABB of its 16 slots (see box_reach)
    const float4* mat;      // n_padded x (r, g, b, param)
    const int*    typ;      // n_padded
    const int*    orig;     // n_padded: the caller's sphere index of each slot (INT_MAX for pad slots)
    const int*    slot_of;  // n: slot of the caller's sphere index
    const float*  rad;      // n_padded: radius (the hit normal divides by it: intersections.h:95)
    unsigned char* scratch; // kWavesPerWg x kWaveScratch bytes of per-wave work space (pair scan)
};

// SCENE: 0 = the whole scene copy in the LDS; 1 = nothing staged, every array read from global memory (L2-resident; scenes beyond the LDS);
//        2 = what every sphere TEST reads (centres, radii^2, group boxes, original indices: 21 bytes per sphere) in the LDS, what only a HIT
//            reads (colour, type, radius, slot index: 32 bytes per sphere) left in global memory - scenes of 500..1700 spheres keep two
//            workgroups per CU this way, scenes up to ~5500 spheres stay out of the all-global form.
template <bool WITH_FB, int SCENE = 0>
__device__ __forceinline__ SceneLds stage_scene(const RtSphereParams& P, unsigned char* smem, float** after) {
    if (SCENE == 1) {                                                // the scene does not fit the LDS: read it where it lies (L2-resident)
        *after = nullptr;
        return { P.spheres, P.groups, P.mat_color, P.mat_type, P.orig, P.slot_of, P.rad, smem };
    }
    float4* s_sph = reinterpret_cast<float4*>(smem);
    float4* s_grp = s_sph + P.n_padded + P.n_groups;
    if (SCENE == 2) {
        int* s_org = reinterpret_cast<int*>(s_grp + 3 * P.n_groups + rt_cell_f4(P.n_groups));
        for (int k = threadIdx.x; k < P.n_padded + P.n_groups; k += (int)blockDim.x) s_sph[k] = P.spheres[k];
        for (int k = threadIdx.x; k < P.n_padded; k += (int)blockDim.x) s_org[k] = P.orig[k];
        for (int k = threadIdx.x; k < 3 * P.n_groups + rt_cell_f4(P.n_groups); k += (int)blockDim.x) s_grp[k] = P.groups[k];     // (boxes + their cell tables)
        *after = nullptr;
        unsigned char* scratch = reinterpret_cast<unsigned char*>(s_org + P.n_padded);
        __syncthreads();
        return { s_sph, s_grp, P.mat_color, P.mat_type, s_org, P.slot_of, P.rad, scratch };
    }
    float4* s_mat = s_grp + 3 * P.n_groups + rt_cell_f4(P.n_groups);
    int*    s_typ = reinterpret_cast<int*>(s_mat + P.n_padded);
    int*    s_org = s_typ + P.n_padded;
    float*  s_rad = reinterpret_cast<float*>(s_org + P.n_padded);
    int*    s_sof = reinterpret_cast<int*>(s_rad + P.n_padded);
    for (int k = threadIdx.x; k < P.n_padded + P.n_groups; k += (int)blockDim.x) s_sph[k] = P.spheres[k];   // the image is laid out for the LDS on the host
    for (int k = threadIdx.x; k < P.n_padded; k += (int)blockDim.x) {
        s_rad[k] = P.rad[k];
        s_mat[k] = P.mat_color[k];
        s_typ[k] = P.mat_type[k];
        s_org[k] = P.orig[k];
    }
    for (int k = threadIdx.x; k < 3 * P.n_groups + rt_cell_f4(P.n_groups); k += (int)blockDim.x) s_grp[k] = P.groups[k];
    for (int k = threadIdx.x; k < P.n; k += (int)blockDim.x) s_sof[k] = P.slot_of[k];
    float* s_fb = reinterpret_cast<float*>(s_sof + ((P.n + 3) & ~3));
    *after = s_fb;                                                   // WITH_FB (tile kernel): kThreads x 3 floats of framebuffer staging
    unsigned char* scratch = reinterpret_cast<unsigned char*>(s_fb + (WITH_FB ? kThreads * 3 : 0));
    __syncthreads();
    return { s_sph, s_grp, s_mat, s_typ, s_org, s_sof, s_rad, scratch };
}

// Per-lane path state (path, helper_structs.h:48-71, minus what sphere scenes never use).
struct Lane {
    uint32_t rng;
    f3 col;                 // pixel accumulator (kernels.cu:547)
    f3 org, dir, atten;     // (the path's colour, path::color of helper_structs.h:57, is not carried: a sphere-scene path collects light once, when it ends in the
                            //  sky - shade() hands that sample colour to the caller, 0 + attenuation * sky as kernels.cu:397,419-421 compute it)
    int bounce;
    bool inside;
    int s;                  // sample index within the pixel
    int i, j;               // global pixel coordinates
    uint32_t pixelId;
};

// kernels.cu:549-555 + the head of color() :397-398: starts sample L.s of the lane's pixel
struct SampleParams { rt_camera cam; int32_t nx, ny, rng_mode; };     // what start_sample reads of the kernel parameters
// COUNTER = false: the caller knows the stream is the reference's (the two dispatches of the cost-ordered frame exist in that mode only), and the
// lane's pixel id is then not read at all
template <bool COUNTER = true, typename PP>
__device__ __forceinline__ void start_sample(const PP& P, Lane& L) {
    if (COUNTER && P.rng_mode == RT_RNG_COUNTER) L.rng = sample_seed(L.pixelId, (uint32_t)L.s);
    const float u = ((float)L.i + rnd(L.rng)) / (float)P.nx;
    const float v = ((float)L.j + rnd(L.rng)) / (float)P.ny;
    f3 d;
    get_ray(P.cam, u, v, L.rng, L.org, d);
    L.dir = unit(d);                                                 // ray.h:9 (get_ray returns a ray)
    L.atten = F3(1.0f, 1.0f, 1.0f);
    L.bounce = 0;
    L.inside = false;
}

// pixel state without its first sample (the caller starts it: start_sample)
__device__ __forceinline__ void init_pixel(const RtSphereParams& P, Lane& L, int i, int j, int first_sample = 0) {
    L.i = i; L.j = j;
    L.pixelId = (uint32_t)(j * P.nx + i);                            // kernels.cu:541 (global id -> seed)
    L.rng = pixel_seed(L.pixelId);
    L.col = F3(0, 0, 0);
    L.s = first_sample;
}

__device__ __forceinline__ void start_pixel(const RtSphereParams& P, Lane& L, int i, int j, int first_sample = 0) {
    init_pixel(P, L, i, j, first_sample);
    start_sample(P, L);
}

// One group of 16 consecutive slots, scanned by every live lane for its own ray (sphere data broadcast from LDS).
// Phase 1: 16 VALU ops per sphere in exactly the reference's rounding order + one v_alignbit that shifts the sign of
// -(discriminant) into a mask.  Phase 2: the literal sphereHit tail for the set bits only.
__device__ __forceinline__ void scan_group_broadcast(const RtSphereParams& P, const SceneLds& S, int g, f3 org, f3 dn, float a, Hit& h) {
    const int base = g << kSphereGroupShift;
    const float4* sp = S.sph + sidx(base);
    uint32_t mask = 0;
#pragma unroll
    for (int kk = 0; kk < kSphereGroup; kk++) {
        const float4 sph = sp[kk];                                   // wave-uniform address: LDS broadcast
        const float ocx = org.x - sph.x;
        const float ocy = org.y - sph.y;
        const float ocz = org.z - sph.z;
        const float b = ocx * dn.x + ocy * dn.y + ocz * dn.z;
        const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - sph.w;
        const float nd = a * c - b * b;                              // == -(b*b - a*c) bit for bit
        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(nd), 31);   // mask = mask<<1 | sign(nd)
    }
    mask <<= (32 - kSphereGroup);                                    // slot `base` at bit 31
    while (mask) {                                                   // candidates
        const int lz = __clz((int)mask);
        mask &= ~(0x80000000u >> lz);
        const int k = base + lz;
        const float t = sphere_hit_exact(S.sph[sidx(k)], org, dn, a, P.t_min, h.closest);
        const int o = S.orig[k];
        if (o != 0x7fffffff && t < FLT_MAX) accept(h, t, k, o);
    }
}

// ---- closest hit, LANE-PARALLEL brute force (kept for A/B): every lane scans every group ---------------------------
// `dn` is the renormalised direction (hit() rebuilds the ray: kernels.cu:326, ray.h:9), a = dot(dn,dn).
__device__ __forceinline__ Hit scan_lane_parallel(const RtSphereParams& P, const SceneLds& S, f3 org, f3 dn, float a, uint32_t& groups_done) {
    Hit h = { FLT_MAX, -1, 0x7fffffff };
    for (int g = 0; g < P.n_groups; g++) {
        groups_done += 4u;
        scan_group_broadcast(P, S, g, org, dn, a, h);
    }
    return h;
}

// ---- closest hit, WAVE-COOPERATIVE brute force (kept for A/B): the 64 lanes share ONE ray (that of lane q) ----------
// Lane l tests slot 64*r + l in round r.  The per-lane results are merged with the reference's tie rule.  Why this is
// still the reference's answer: sphereHit's result for sphere k does not depend on the running `closest` except for
// acceptance (the far root is never below the near root), so the linear scan computes the lexicographic minimum of
// (t_k, k) over all spheres — which can be evaluated in any order.  WAVE-LEVEL; q is wave-uniform.
__device__ __forceinline__ Hit scan_cooperative(const RtSphereParams& P, const SceneLds& S, int q, f3 org, f3 dn, float a) {
    const int lane = threadIdx.x & 63;
    const f3 O = F3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(org.x), q)),
                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(org.y), q)),
                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(org.z), q)));
    const f3 D = F3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(dn.x), q)),
                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dn.y), q)),
                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dn.z), q)));
    const float A = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), q));
    const int rounds = P.n_padded >> 6;
    Hit h = { FLT_MAX, -1, 0x7fffffff };
    for (int r = 0; r < rounds; r++) {
        const int k = (r << 6) + lane;
        const float4 sph = S.sph[sidx(k)];
        const f3 oc = O - F3(sph.x, sph.y, sph.z);
        const float b = oc.x * D.x + oc.y * D.y + oc.z * D.z;
        const float c = (oc.x * oc.x + oc.y * oc.y + oc.z * oc.z) - sph.w;
        const float nd = A * c - b * b;
        if (nd < 0.0f) {
            const float t = sphere_hit_exact(sph, O, D, A, P.t_min, h.closest);
            const int o = S.orig[k];
            if (o != 0x7fffffff && t < FLT_MAX) accept(h, t, k, o);
        }
    }
    // merge: lexicographic minimum of (t, orig) over the lanes that found something.  t > t_min >= 0, so the
    // IEEE bit patterns order like the values and the merge runs on the scalar unit.
    unsigned long long found = __ballot(h.closest < FLT_MAX);
    uint32_t st = __float_as_uint(FLT_MAX);
    int sk = -1, so = 0x7fffffff;
    while (found) {
        const int b = __builtin_ctzll(found);
        found &= found - 1;
        const uint32_t tb = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(h.closest), b);
        const int ob = __builtin_amdgcn_readlane(h.orig, b);
        if (tb < st || (tb == st && ob < so)) { st = tb; so = ob; sk = __builtin_amdgcn_readlane(h.sid, b); }
    }
    Hit out = { __uint_as_float(st), sk, so };
    return out;
}

// ---- exact culling: the ray side of the box tests ----------------------------------------------------------------------
// A group (or hierarchy node) may be skipped for a ray iff no sphere in it can give the reference's scan a hit with
// t <= the ray's current closest hit.  The test is a slab test of the ray against the node's AABB, made CONSERVATIVE:
//  * the box is widened, per ray, by a margin m that covers (a) the rounding of the reference's own fp32 discriminant
//    b*b - a*c (intersections.h:87-91): its error is <= 21 eps |oc|^2, so the reference reports hits for rays whose line
//    passes a sphere of radius r at a distance up to r_eff = sqrt(r^2 + K eps |oc|^2) - far from the scene that is far
//    outside any fixed inflation (camera at 100x the scene extent).  m = r_eff - r <= min(K eps D^2 / (2 r_min),
//    sqrt(K eps) D) with D >= |oc| for every small sphere (distance to the centre cloud's bounding sphere + its radius),
//    K = 96 (21 for the discriminant, the rest for the position of the phantom root along the ray); and (b) the rounding of
//    this slab test itself, <= 8 eps (|origin| + |box|) in position: k3 x (max |org| + coord_max);
//  * the entry distance is compared with cb = (closest + 1e-4) * 1.00002 >= closest;
//  * a NaN (0 * inf, inf - inf on a slab plane / axis-parallel ray) is dropped by min / max, i.e. ignores that axis: keeps the node.
// Not being part of the reference's arithmetic, the test may use what is fastest: v_rcp_f32 and fused multiply-adds
// (lo * inv - (org + m) * inv: six v_fma_f32 per box instead of twelve sub / mul).
// The two planes of an axis are not ordered by min / max after the fact: the box stores (lo, hi, lo) per axis and the ray reads two consecutive
// floats at offset 0 (direction >= 0: near = lo, far = hi) or 1 (near = hi, far = lo) - six VALU instructions fewer per box.
struct BoxRay { f3 inv, cn, cf; float cb, m; uint32_t sx, sy, sz; };   // cn / cf: (org +- m) * inv of the near / far plane; m: the margin; s*: 0 or 1

__device__ __forceinline__ BoxRay make_box_ray(const RtSphereParams& P, f3 org, f3 dn, float closest) {
    BoxRay r;
    r.inv = F3(__builtin_amdgcn_rcpf(dn.x), __builtin_amdgcn_rcpf(dn.y), __builtin_amdgcn_rcpf(dn.z));
    const f3 dc = org - F3(P.cull_cx, P.cull_cy, P.cull_cz);
    const float D = __builtin_amdgcn_sqrtf(dot(dc, dc)) * 1.000001f + P.cull_radius;
    const float m = fminf(P.cull_k1 * D * D, P.cull_k2 * D) + P.cull_k3 * (fmaxf(fmaxf(fabsf(org.x), fabsf(org.y)), fabsf(org.z)) + P.cull_coord_max);
    // plane lo - m: t = lo * inv - (org + m) * inv; plane hi + m: t = hi * inv - (org - m) * inv
    const f3 clo = F3((org.x + m) * r.inv.x, (org.y + m) * r.inv.y, (org.z + m) * r.inv.z);
    const f3 chi = F3((org.x - m) * r.inv.x, (org.y - m) * r.inv.y, (org.z - m) * r.inv.z);
    r.sx = __float_as_uint(r.inv.x) >> 31; r.sy = __float_as_uint(r.inv.y) >> 31; r.sz = __float_as_uint(r.inv.z) >> 31;
    r.cn = F3(r.sx ? chi.x : clo.x, r.sy ? chi.y : clo.y, r.sz ? chi.z : clo.z);
    r.cf = F3(r.sx ? clo.x : chi.x, r.sy ? clo.y : chi.y, r.sz ? clo.z : chi.z);
    r.cb = (closest + 1.0e-4f) * 1.00002f;
    r.m = m;
    return r;
}

// >= 0 (sign bit clear) = the ray may reach the box [lo - m, hi + m] at a distance <= its current closest hit.  `g3` = the group's three
// float4.  Both terms are NaN-free (v_max3 / v_min3 drop NaN operands and each has a finite one) and never both +inf, so the difference is no NaN.
__device__ __forceinline__ float box_gap(const float4* g3, const BoxRay& r) {
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    const float* f = reinterpret_cast<const float*>(g3);
    const f2u px = *reinterpret_cast<const f2u*>(f + r.sx), py = *reinterpret_cast<const f2u*>(f + 4 + r.sy), pz = *reinterpret_cast<const f2u*>(f + 8 + r.sz);
    const float t_in = fmaxf(fmaxf(__builtin_fmaf(px.x, r.inv.x, -r.cn.x), __builtin_fmaf(py.x, r.inv.y, -r.cn.y)), __builtin_fmaf(pz.x, r.inv.z, -r.cn.z));
    const float t_out = fminf(fminf(__builtin_fmaf(px.y, r.inv.x, -r.cf.x), __builtin_fmaf(py.y, r.inv.y, -r.cf.y)), __builtin_fmaf(pz.y, r.inv.z, -r.cf.z));
    // skip iff (t_in > t_out) || (t_out < 0) || (t_in > cb), cb >= 0
    return fminf(t_out, r.cb) - fmaxf(t_in, 0.0f);
}
__device__ __forceinline__ bool box_reach(const float4* g3, const BoxRay& r) { return !(box_gap(g3, r) < 0.0f); }

template <int I> __device__ __forceinline__ float comp(f3 v) { return I == 0 ? v.x : (I == 1 ? v.y : v.z); }
template <int I> __device__ __forceinline__ uint32_t bsign(const BoxRay& r) { return I == 0 ? r.sx : (I == 1 ? r.sy : r.sz); }

// The boxes of spheres that rest on a plane (the random-spheres scene: every small sphere has the same height and radius) have the same extent on
// one axis (found on the host: RtSphereParams::box_shared_axis).  That axis' slab is then the same for every box: evaluated once per pass and
// folded into the two clamps the test has anyway (t_in against 0, t_out against the closest hit) - 8 instead of 12 instructions per box.
template <int AX>
__device__ __forceinline__ uint32_t group_needs_shared(const SceneLds& S, int g0, int ng, const BoxRay& r, float shared_lo, float shared_hi) {
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    constexpr int A1 = (AX + 1) % 3, A2 = (AX + 2) % 3;
    const float* f = reinterpret_cast<const float*>(S.grp + 3 * g0);
    const float s_near = bsign<AX>(r) ? shared_hi : shared_lo, s_far = bsign<AX>(r) ? shared_lo : shared_hi;
    const float in0 = fmaxf(__builtin_fmaf(s_near, comp<AX>(r.inv), -comp<AX>(r.cn)), 0.0f);
    const float out0 = fminf(__builtin_fmaf(s_far, comp<AX>(r.inv), -comp<AX>(r.cf)), r.cb);
    const float* f1 = f + 4 * A1 + bsign<A1>(r);
    const float* f2 = f + 4 * A2 + bsign<A2>(r);
    uint32_t skip = 0;
    auto gap = [&](f2u p1, f2u p2) {
        const float t_in = fmaxf(fmaxf(__builtin_fmaf(p1.x, comp<A1>(r.inv), -comp<A1>(r.cn)), __builtin_fmaf(p2.x, comp<A2>(r.inv), -comp<A2>(r.cn))), in0);
        const float t_out = fminf(fminf(__builtin_fmaf(p1.y, comp<A1>(r.inv), -comp<A1>(r.cf)), __builtin_fmaf(p2.y, comp<A2>(r.inv), -comp<A2>(r.cf))), out0);
        return __float_as_uint(t_out - t_in);
    };
    // four boxes at a time, their eight LDS reads issued before the first test: one round trip per four boxes instead of one per box
    int g = 0;
    for (; g + 4 <= ng; g += 4) {
        f2u a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { a[k] = *reinterpret_cast<const f2u*>(f1 + 12 * (g + k)); b[k] = *reinterpret_cast<const f2u*>(f2 + 12 * (g + k)); }
        __builtin_amdgcn_sched_barrier(0);                           // keep the reads together: the scheduler otherwise re-serialises read -> test
#pragma unroll
        for (int k = 0; k < 4; k++) skip = __builtin_amdgcn_alignbit(skip, gap(a[k], b[k]), 31);
    }
    for (; g < ng; g++) skip = __builtin_amdgcn_alignbit(skip, gap(*reinterpret_cast<const f2u*>(f1 + 12 * g), *reinterpret_cast<const f2u*>(f2 + 12 * g)), 31);
    return __brev(~skip << (32 - ng));
}

// The prefilter for scenes whose group boxes SHARE an axis (template AX; spheres resting on a plane: the benchmark) - group_needs_cells below with the shared
// slab in place of the union box and a two-axis box test (8 instead of 12 instructions per box; the union clip and the third table cost the benchmark frame
// 5 %, profiles/r04_ab_cells3_c5.txt): the part of the ray inside the shared
// slab - t in [in0, out0], already clipped to [0, closest hit] - has a bounding rectangle on the two other axes, and the host's cell tables give, per cell,
// the boxes that begin at or below it and those that end at or above it: four LDS words and three ANDs leave the boxes whose rectangle overlaps the
// ray's (on the benchmark scene 2-3 of the 31), and only those take the box test, two per step of a per-lane loop: the wave runs as many steps as
// its lane with the most candidates needs (a grazing ray that lies long in the slab), which measured better than falling back to the uniform loop
// from any threshold (A/B on C5 at 256 spp, profiles/r03_ab_cells.txt: uniform loop 9765, prefilter 10280 Msamples/s; 64 cells per axis; 32: -0.2 %,
// 128: -0.3 %; fall-back above 6 candidates -1.8 %, above 12 -0.3 %).
// Conservative by itself: a box the ray enters at t* in [0, closest] contains the point org + t* d (up to the margin m, by which the ray's rectangle is
// widened here as the boxes are in the test); t* lies in the computed [in0, out0] (that is the slab test's own guarantee, the shared axis being one of
// its three), so the point's coordinates lie between the segment's end points - computed with two roundings each, covered by the 2^-20 relative pad - and
// the cell index is off by less than the tables' slack.  An infinite or NaN end point (no hit yet and a ray parallel to the slab) clamps to
// the first / last cell, whose words reject nothing.
template <int AX>
__device__ __forceinline__ uint32_t group_needs_cells_shared(const RtSphereParams& P, const SceneLds& S, int g0, int ng, const BoxRay& r, f3 org, f3 dn,
                                                             float shared_lo, float shared_hi, uint32_t& boxes_done, int word) {
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    constexpr int A1 = (AX + 1) % 3, A2 = (AX + 2) % 3;
    const float* f = reinterpret_cast<const float*>(S.grp + 3 * g0);
    const uint32_t* tab = reinterpret_cast<const uint32_t*>(S.grp + 3 * P.n_groups) + word;      // this pass' 32 groups: word `word` of every set
    const int W = rt_cell_words(P.n_groups);
    const float s_near = bsign<AX>(r) ? shared_hi : shared_lo, s_far = bsign<AX>(r) ? shared_lo : shared_hi;
    const float in0 = fmaxf(__builtin_fmaf(s_near, comp<AX>(r.inv), -comp<AX>(r.cn)), 0.0f);
    const float out0 = fminf(__builtin_fmaf(s_far, comp<AX>(r.inv), -comp<AX>(r.cf)), r.cb);
    auto overlap = [&](float o, float d, int q) -> uint32_t {
        const float x0 = __builtin_fmaf(in0, d, o), x1 = __builtin_fmaf(out0, d, o);
        const float pad = r.m + 9.5367431640625e-7f * (fabsf(o) + fabsf(x0) + fabsf(x1));
        const float lo = fminf(x0, x1) - pad, hi = fmaxf(x0, x1) + pad;
        // (a NaN goes to the permissive end: fmaxf / fminf return their other operand)
        const float c_lo = fminf(fmaxf(__builtin_fmaf(lo, P.cell_scale[q], P.cell_off[q]), 0.0f), (float)(kCellCount - 1));
        const float c_hi = fmaxf(fminf(__builtin_fmaf(hi, P.cell_scale[q], P.cell_off[q]), (float)(kCellCount - 1)), 0.0f);
        return tab[__umul24((uint32_t)(2 * kCellCount * q) + (uint32_t)c_hi, (uint32_t)W)] & tab[__umul24((uint32_t)(2 * kCellCount * q + kCellCount) + (uint32_t)c_lo, (uint32_t)W)];     // (24-bit multiply: full rate)
    };
    uint32_t cand = overlap(comp<A1>(org), comp<A1>(dn), A1) & overlap(comp<A2>(org), comp<A2>(dn), A2);      // (tables are indexed by axis)
    if (!(in0 <= out0)) cand = 0u;                                   // the ray misses the slab (or leaves it behind its closest hit): the box test would skip every box
    boxes_done += (uint32_t)__popc(cand);
    const float* f1 = f + 4 * A1 + bsign<A1>(r);
    const float* f2 = f + 4 * A2 + bsign<A2>(r);
    auto gap = [&](f2u p1, f2u p2) {
        const float t_in = fmaxf(fmaxf(__builtin_fmaf(p1.x, comp<A1>(r.inv), -comp<A1>(r.cn)), __builtin_fmaf(p2.x, comp<A2>(r.inv), -comp<A2>(r.cn))), in0);
        const float t_out = fminf(fminf(__builtin_fmaf(p1.y, comp<A1>(r.inv), -comp<A1>(r.cf)), __builtin_fmaf(p2.y, comp<A2>(r.inv), -comp<A2>(r.cf))), out0);
        return __float_as_uint(t_out - t_in);
    };
    uint32_t need = 0;
    // The loop below is per lane: the wave runs it as long as its lane with the most candidates needs (a grazing ray that lies long in the slab), two box
    // tests per step.  A step that serves only a few lanes costs the wave more than it can save: an untested candidate simply becomes a (ray, group) pair -
    // 16 sphere pre-tests on ONE lane of a pair round, the culling being an optimisation, never part of the result.  So the loop ends when fewer than
    // RT_BOX_CUT lanes still have candidates, and what they have left is taken untested.
    while (__popcll(__ballot(cand != 0u)) >= RT_BOX_CUT) {
        if (cand != 0u) {
            const int ga = __builtin_ctz(cand);
            cand &= cand - 1u;
            const int gb = cand != 0u ? __builtin_ctz(cand) : ga;
            cand &= cand - 1u;
            const f2u a1 = *reinterpret_cast<const f2u*>(f1 + 12 * ga), a2 = *reinterpret_cast<const f2u*>(f2 + 12 * ga);
            const f2u b1 = *reinterpret_cast<const f2u*>(f1 + 12 * gb), b2 = *reinterpret_cast<const f2u*>(f2 + 12 * gb);
            need |= ((~gap(a1, a2)) >> 31) << ga;
            need |= ((~gap(b1, b2)) >> 31) << gb;
        }
    }
    return need | cand;
}

// The box tests behind a PREFILTER (RtSphereParams::cell_on: scenes of up to 256 small groups).  The ray is first clipped to the UNION of the group boxes (one slab
// test, the margin folded in as for every box): what is left, t in [in0, out0] inside [0, closest hit], has a bounding box, and the host's cell tables give, per
// axis and cell, the boxes that begin at or below the cell and those that end at or above it: per axis four LDS words (two table entries for the two ends of
// the segment's extent, `word` of each) and an AND leave the boxes whose extent overlaps the segment's on that axis; the AND over the axes whose table can reject
// anything (cell_axes: spheres resting on a plane skip the vertical axis) leaves the boxes whose box overlaps the segment's bounding box - on the benchmark scene
// 2-3 of the 31 - and only those take the box test, two per step of a per-lane loop: the wave runs as many steps as its lane with the most candidates needs
// (a grazing ray that lies long in the slab), until few lanes are left (RT_BOX_CUT).  (A/B on C5 at 256 spp, round 3, profiles/r03_ab_cells.txt: uniform loop
// 9765, prefilter 10280 Msamples/s; 64 cells per axis; 32: -0.2 %, 128: -0.3 %.)
// Conservative by itself: a box the ray enters at t* in [0, closest] lies inside the union box, so t* is inside the union's own slab interval [in0, out0]
// (the slab test's guarantee, with the same margin m); the entry point org + t* d (up to m, by which the segment's extent is widened here as the boxes are in
// the test) then lies between the segment's end points on every axis - computed with two roundings each, covered by the 2^-20 relative pad - and the cell index
// is off by less than the tables' slack.  An infinite or NaN end point (no hit yet, a ray parallel to a slab) clamps to the first / last cell, whose words
// reject nothing.
__device__ __forceinline__ uint32_t group_needs_cells(const RtSphereParams& P, const SceneLds& S, int g0, int ng, const BoxRay& r, f3 org, f3 dn,
                                                      uint32_t& boxes_done, int word) {
    const uint32_t* tab = reinterpret_cast<const uint32_t*>(S.grp + 3 * P.n_groups) + word;      // this pass' 32 groups: word `word` of every set
    const int W = rt_cell_words(P.n_groups);
    // the union box: near / far plane per axis by the sign of the direction (BoxRay convention)
    const float nx_ = r.sx ? P.ubox[3] : P.ubox[0], fx_ = r.sx ? P.ubox[0] : P.ubox[3];
    const float ny_ = r.sy ? P.ubox[4] : P.ubox[1], fy_ = r.sy ? P.ubox[1] : P.ubox[4];
    const float nz_ = r.sz ? P.ubox[5] : P.ubox[2], fz_ = r.sz ? P.ubox[2] : P.ubox[5];
    const float t_in = fmaxf(fmaxf(__builtin_fmaf(nx_, r.inv.x, -r.cn.x), __builtin_fmaf(ny_, r.inv.y, -r.cn.y)), __builtin_fmaf(nz_, r.inv.z, -r.cn.z));
    const float t_out = fminf(fminf(__builtin_fmaf(fx_, r.inv.x, -r.cf.x), __builtin_fmaf(fy_, r.inv.y, -r.cf.y)), __builtin_fmaf(fz_, r.inv.z, -r.cf.z));
    const float in0 = fmaxf(t_in, 0.0f), out0 = fminf(t_out, r.cb);
    auto overlap = [&](float o, float d, int q) -> uint32_t {
        const float x0 = __builtin_fmaf(in0, d, o), x1 = __builtin_fmaf(out0, d, o);
        const float pad = r.m + 9.5367431640625e-7f * (fabsf(o) + fabsf(x0) + fabsf(x1));
        const float lo = fminf(x0, x1) - pad, hi = fmaxf(x0, x1) + pad;
        // (a NaN goes to the permissive end: fmaxf / fminf return their other operand)
        const float c_lo = fminf(fmaxf(__builtin_fmaf(lo, P.cell_scale[q], P.cell_off[q]), 0.0f), (float)(kCellCount - 1));
        const float c_hi = fmaxf(fminf(__builtin_fmaf(hi, P.cell_scale[q], P.cell_off[q]), (float)(kCellCount - 1)), 0.0f);
        return tab[__umul24((uint32_t)(2 * kCellCount * q) + (uint32_t)c_hi, (uint32_t)W)] & tab[__umul24((uint32_t)(2 * kCellCount * q + kCellCount) + (uint32_t)c_lo, (uint32_t)W)];     // (24-bit multiply: full rate)
    };
    uint32_t cand = ng >= 32 ? 0xFFFFFFFFu : ((1u << ng) - 1u);
    if (P.cell_axes & 1) cand &= overlap(org.x, dn.x, 0);
    if (P.cell_axes & 2) cand &= overlap(org.y, dn.y, 1);
    if (P.cell_axes & 4) cand &= overlap(org.z, dn.z, 2);
    if (!(in0 <= out0)) cand = 0u;                                   // the ray misses the union box (or leaves it behind its closest hit): the box test would skip every box
    boxes_done += (uint32_t)__popc(cand);
    const float4* g3 = S.grp + 3 * g0;
    uint32_t need = 0;
    // The loop below is per lane: the wave runs it as long as its lane with the most candidates needs, two box tests per step.  A step that serves only a few
    // lanes costs the wave more than it can save: an untested candidate simply becomes a (ray, group) pair - 16 sphere pre-tests on ONE lane of a pair round,
    // the culling being an optimisation, never part of the result.  So the loop ends when fewer than RT_BOX_CUT lanes still have candidates, and what they
    // have left is taken untested.
    while (__popcll(__ballot(cand != 0u)) >= RT_BOX_CUT) {
        if (cand != 0u) {
            const int ga = __builtin_ctz(cand);
            cand &= cand - 1u;
            const int gb = cand != 0u ? __builtin_ctz(cand) : ga;    // (a lone last candidate is tested twice: the same bit)
            cand &= cand - 1u;
            const float gap_a = box_gap(g3 + 3 * ga, r), gap_b = box_gap(g3 + 3 * gb, r);
            need |= ((~__float_as_uint(gap_a)) >> 31) << ga;
            need |= ((~__float_as_uint(gap_b)) >> 31) << gb;
        }
    }
    return need | cand;
}

__device__ __forceinline__ uint32_t group_needs(const RtSphereParams& P, const SceneLds& S, int g0, int ng, const BoxRay& br, bool cull) {
    if (!cull) return (ng >= 32) ? 0xFFFFFFFFu : ((1u << ng) - 1u);
    if (P.box_shared_axis == 2) return group_needs_shared<1>(S, g0, ng, br, P.box_shared_lo, P.box_shared_hi);
    if (P.box_shared_axis == 1) return group_needs_shared<0>(S, g0, ng, br, P.box_shared_lo, P.box_shared_hi);
    if (P.box_shared_axis == 3) return group_needs_shared<2>(S, g0, ng, br, P.box_shared_lo, P.box_shared_hi);
    // the skip flags are the SIGN BITS of the gaps, shifted in one v_alignbit per box (compare + select + shift + or otherwise);
    // box g of the pass ends up at bit ng - 1 - g: one bit reversal per pass puts it back at bit g
    uint32_t skip = 0;
#pragma unroll 4
    for (int g = 0; g < ng; g++) skip = __builtin_amdgcn_alignbit(skip, __float_as_uint(box_gap(S.grp + 3 * (g0 + g), br)), 31);
    return __brev(~skip << (32 - ng));                               // 1 <= ng <= 32
}

// Inclusive prefix sum over the 64 lanes with DPP moves (row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast:15
// and row_bcast:31 carry the row totals across): six VALU instructions and no LDS round trip, where six __shfl_up steps are
// six dependent ds_bpermute latencies.  WAVE-LEVEL: all 64 lanes must be active.
__device__ __forceinline__ int wave_inclusive_scan(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true);   // row_bcast:15 -> rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true);   // row_bcast:31 -> rows 2 and 3
    return x;
}

// ---- closest hit, PAIR-COMPACTED form (default) ---------------------------------------------------------------------
// 1. Every live lane scans the BIG spheres (ground, unit spheres) for its own ray: a first `closest`.
// 2. Every live lane finds the small-sphere groups its ray can still reach before that hit (a few of the ~31).
// 3. The (ray, group) PAIRS of the whole wave are written densely into an LDS list (wave64 prefix sum of the per-lane
//    pair counts) and processed 64 at a time: pair-lane j fetches ray + group of pair j, runs the 16 sphere tests,
//    resolves its candidates exactly and folds its best (t, original index) into the owner's slot with one 64-bit LDS
//    atomic min (t > 0, so the IEEE bits order like the values: the minimum of (t_bits << 32 | orig) IS the
//    reference's first-index-wins closest hit).
// The work a wave does is proportional to the pairs that exist, not to 64 x (union of groups): incoherent waves do not
// pay for each other's groups, and a wave with few live rays uses all 64 lanes on them.  WAVE-LEVEL: all 64 lanes call it.
// One-list form (the launcher's promise: culling on, cell tables on, at most 32 x OP small groups - the benchmark scene and everything of its size): the pass
// loop below runs exactly once with every pass-level decision known at compile time (no windows, no carry between passes, no choice of the box test).
//   OP   words of the one-list form: 0 = the general pass loop; 1 / 2 = the scene has at most 32 / 64 small groups, which take ONE list per ray batch
//   C3   (with OP > 0) the prefilter is the general 3-axis one (group_needs_cells) instead of the shared-vertical-axis one (group_needs_cells_shared<1>)
template <int OP = 0, bool C3 = false>
__device__ __forceinline__ Hit scan_pairs(const RtSphereParams& P, const SceneLds& S, f3 org, f3 dn, float a, bool has_ray, bool cull,
                                          uint32_t& groups_done, uint32_t& boxes_done, unsigned long long* tm = nullptr) {
    // tm (diagnostic instantiation only): cycles in [1] big spheres, [2] group boxes + pair list, [3] pair rounds, [4] candidates
    unsigned long long tc = tm ? __builtin_amdgcn_s_memtime() : 0ull;
    auto lap = [&](int k) { if (tm) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tm[k] += n_ - tc; tc = n_; } };
    const int lane = threadIdx.x & 63;
    constexpr bool ONEPASS = OP > 0;
    unsigned char* W = S.scratch + (threadIdx.x >> 6) * ws_total(ONEPASS);
    float4* w_ray = reinterpret_cast<float4*>(W);                               // ray r: (origin, a) at w_ray[r], (direction, -) at w_ray[64 + r] - two arrays of 16-byte
                                                                                // entries: the pair lanes of a round fetch rays of (mostly) consecutive owners, and 16
                                                                                // consecutive 16-byte entries cover all 64 banks once (32-byte entries: twice)
    unsigned long long* w_best = reinterpret_cast<unsigned long long*>(W + 64 * 32);
    unsigned short* w_pair = reinterpret_cast<unsigned short*>(W + 64 * 32 + 64 * 8);
    uint32_t* w_cand = reinterpret_cast<uint32_t*>(W + ws_pairs(ONEPASS));
    const float t_min = P.t_min;
    uint32_t n_c = 0;                                                // wave-uniform: candidates in the list (a register: the list is this wave's own)

    // exact resolution of ONE candidate (owner ray, sphere slot) by this lane: the literal sphereHit tail, merged into the
    // owner's slot with the (t, original index) key.  t_max = FLT_MAX: a root beyond the owner's current best loses the min anyway.
    auto resolve = [&](uint32_t e) {
        const int owner = (int)(e >> 24), k = (int)(e & 0xFFFFFFu);
        const float4 ro = w_ray[owner], rd = w_ray[64 + owner];
        const float t = sphere_hit_exact(S.sph[sidx(k)], F3(ro.x, ro.y, ro.z), F3(rd.x, rd.y, rd.z), ro.w, t_min, FLT_MAX);
        const int o = S.orig[k];
        if (o != 0x7fffffff && t < FLT_MAX)
            atomicMin(&w_best[owner], ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(uint32_t)o);
    };

    w_ray[lane] = make_float4(org.x, org.y, org.z, a);
    w_ray[64 + lane] = make_float4(dn.x, dn.y, dn.z, 0.0f);
    w_best[lane] = ~0ull;                                            // (published, with the ray table, by the fence behind the first pair list: nothing reads them before)

    // Appends every lane's candidates (`bits`: one bit per candidate; `entry_of` pops the next one) to the wave's list.  The positions come from
    // ballots, level by level - level j = the lanes that have more than j candidates, a lane's j-th candidate lands at n_c + (lanes of level j below
    // it) - and the count lives in a register: no LDS counter, no atomic (round 2 reserved positions with one returning LDS atomic per lane on ONE
    // word: 30-40 lanes serialised on it, most of the kernel's LDS bank-conflict cycles, with the atomic's latency in front of every round).  The order of
    // the list is irrelevant (the merge is an atomic min).  A lane with more than kLevels candidates, or a full list, resolves in place.  WAVE-LEVEL.
    constexpr int kLevels = 3;
    auto append = [&](uint32_t bits, auto entry_of) {
#pragma unroll
        for (int lvl = 0; lvl < kLevels; lvl++) {
            const unsigned long long m = __ballot(bits != 0u);
            if (m == 0ull) break;
            if (bits != 0u) {
                const uint32_t pos = n_c + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (pos < (uint32_t)ws_cand_cap(ONEPASS)) w_cand[pos] = entry_of(bits);     // (a full list leaves the candidate in `bits`)
            }
            n_c = min(n_c + (uint32_t)__popcll(m), (uint32_t)ws_cand_cap(ONEPASS));
        }
        while (bits != 0u) resolve(entry_of(bits));                  // more than kLevels candidates in one lane, or the list is full (never seen on C2): in place
    };

    // 1. big spheres (ground, unit spheres; there are only a few): every lane runs phase 1 for its own ray.  The exact
    //    sphereHit tail (IEEE sqrt + divide, taken by an uneven half of the lanes) goes through the candidate list like any
    //    other; what the group culling below needs NOW is only an upper bound of the closest big-sphere hit.  `bound` is
    //    that: the roots from v_sqrt_f32 / v_rcp_f32 (1 ulp each) pushed up by 6x their worst-case error, and a root is only
    //    trusted to lie above t_min if it does so by that margin - so bound >= the exact closest hit of the reference
    //    (near root accepted: closest <= t1 <= bound; near root at or below t_min: the far root bounds whatever is accepted).
    float bound = FLT_MAX;
    if (has_ray) groups_done += (uint32_t)P.n_big_groups * 4u;      // in units of 4 sphere tests
    const float ra = __builtin_amdgcn_rcpf(a);
    for (int k0 = 0; k0 < P.n_big; k0 += 32) {
        uint32_t bm = 0;
        const int kn = min(32, P.n_big - k0);
        if (has_ray) {
#pragma unroll 4
            for (int k = 0; k < kn; k++) {
                const float4 sph = S.sph[sidx(k0 + k)];              // wave-uniform address: LDS broadcast
                const float ocx = org.x - sph.x, ocy = org.y - sph.y, ocz = org.z - sph.z;
                const float b = ocx * dn.x + ocy * dn.y + ocz * dn.z;
                const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - sph.w;
                const float nd = a * c - b * b;                      // == -(b*b - a*c) bit for bit
                const float sq = __builtin_amdgcn_sqrtf(fmaxf(-nd, 0.0f));
                const float err = (fabsf(b) + sq) * ra * 2.0e-6f;
                const float t1 = (-b - sq) * ra, t2 = (-b + sq) * ra;
                // a candidate needs the exact tail only if its FAR root can lie above t_min at all: a ray that leaves the ground sphere
                // (origin on the sphere, both roots <= ~0: every secondary ray from a ground hit) has a positive discriminant, and its
                // exact tail - IEEE sqrt + two IEEE divides - would only find FLT_MAX.  t2 + err bounds the reference's far root from above.
                const bool cand = nd < 0.0f && (t2 + err > t_min);
                const float lo = t_min + err;
                const float tb = t1 > lo ? t1 + err : (t2 > lo ? t2 + err : FLT_MAX);
                if (cand) bound = fminf(bound, tb);
                bm |= (cand ? 1u : 0u) << k;
            }
        }
        append(bm, [&](uint32_t& bits) {
            const int k = __builtin_ctz(bits);
            bits &= bits - 1u;
            return ((uint32_t)lane << 24) | (uint32_t)(k0 + k);
        });
    }
    lap(1);
    const BoxRay br = make_box_ray(P, org, dn, bound);

    // The pair list is filled pass by pass (kPassGroups groups at a time); only FULL rounds of 64 pairs are processed
    // inside a pass, the remainder is carried to the front of the next pass's list, so a partial round runs once per
    // ray batch instead of once per pass.  List entries hold the ABSOLUTE group index (8 bits) and the owner lane.
    int carry = 0;                                                   // wave-uniform: pairs already in the list
    int win_base = P.n_big_groups;                                   // entries hold lane << 10 | (group - win_base): a window of 1024 groups
    // A scene of up to 32 small groups (the benchmark: 31) takes ONE pass of 32: one prefix sum, one list write, one synchronisation per ray batch instead of
    // two.  A pass of 32 whose pairs would not fit the list (every ray reaching a good part of the scene: not seen) is split into quarters of 8 groups (wave-uniform).
    constexpr int NW = OP > 0 ? OP : 1;                             // need words per pass
    const int pass_w = ONEPASS ? 32 * OP : ((P.n_groups - P.n_big_groups <= 32 || (cull && P.cell_on != 0)) ? 32 : kPassGroups);
    for (int g0 = P.n_big_groups; g0 < P.n_groups; g0 += pass_w) {
        const int ng = ONEPASS ? P.n_groups - g0 : min(pass_w, P.n_groups - g0);
        // the last pass runs the partial round too; so does a pass at the end of a 1024-group window (scenes beyond 16 k spheres only)
        const bool flush = !ONEPASS && g0 + 2 * pass_w - win_base > 1024;
        const bool last_pass = ONEPASS || g0 + pass_w >= P.n_groups || flush;
        uint32_t need_w[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) need_w[w] = 0u;
        if (has_ray) {
            if (ONEPASS || (cull && P.cell_on != 0)) {               // (words of 32 groups: word k of the cell sets)
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    const int gw = g0 + 32 * w, nw = min(32, P.n_groups - gw), word = (gw - P.n_big_groups) >> 5;
                    if (w > 0 && nw <= 0) break;
                    if (ONEPASS ? C3 : (P.box_shared_axis != 2)) need_w[w] = group_needs_cells(P, S, gw, nw, br, org, dn, boxes_done, word);
                    else need_w[w] = group_needs_cells_shared<1>(P, S, gw, nw, br, org, dn, P.box_shared_lo, P.box_shared_hi, boxes_done, word);
                }
            } else {
                need_w[0] = group_needs(P, S, g0, ng, br, cull);
                boxes_done += (uint32_t)ng;
            }
        }
        // exclusive prefix sum of the pair counts over the wave
        int cnt_all = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) cnt_all += __popc(need_w[w]);
        const int incl_all = wave_inclusive_scan(cnt_all);
        // a pass whose pairs would not fit the list is cut into quarters of its words, 8 groups each (64 x 8 + the carried remainder always fit)
        const bool split = pass_w >= 32 && carry + __builtin_amdgcn_readlane(incl_all, 63) > ws_list_cap(ONEPASS);
      for (int half = 0; half < (split ? 4 * NW : 1); half++) {
        const int sw = half >> 2, sq = half & 3;                     // (split: word and quarter of this sub-pass)
        uint32_t need[NW];
        int cnt = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            need[w] = split ? (w == sw ? ((need_w[w] >> (8 * sq)) & 0xFFu) << (8 * sq) : 0u) : need_w[w];
            cnt += __popc(need[w]);
        }
        const bool last_sub = last_pass && (!split || half == 4 * NW - 1);
        const int incl = split ? wave_inclusive_scan(cnt) : incl_all;
        const int total = carry + __builtin_amdgcn_readlane(incl, 63);
        int at = carry + incl - cnt;
#pragma unroll
        for (int w = 0; w < NW; w++)
            for (uint32_t m = need[w]; m; m &= m - 1) w_pair[at++] = (unsigned short)((lane << 10) | (g0 + 32 * w - win_base + __builtin_ctz(m)));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        const int stop = last_sub ? total : (total & ~63);           // full rounds only, except in the last pass
        lap(2);
        for (int base = 0; base < stop; base += 64) {
            // A round takes 64 pairs, one per lane, 16 sphere tests each.  The LAST round of a ray batch is partial (a batch has ~68 pairs:
            // one full round and a handful): with <= 32 (<= 16) pairs left, every pair is shared by 2 (4) lanes, 8 (4) tests each - the
            // round issues a half (a quarter) of the instructions instead of running 16 tests on a few lanes.
            const int left = stop - base;                            // wave-uniform
            const int ls = left > 32 ? 0 : (left > 16 ? 1 : 2);
            const int spl = kSphereGroup >> ls;                      // sphere tests per lane
            const int j = base + (lane >> ls);
            uint32_t mask = 0;                                       // this lane's candidates: bit 31 - k = slot0 + k
            int owner = 0, slot0 = 0;
            if (j < stop) {
                const unsigned pr = w_pair[j];
                owner = (int)(pr >> 10);
                slot0 = ((win_base + (int)(pr & 0x3FFu)) << kSphereGroupShift) + (lane & ((1 << ls) - 1)) * spl;
                const int sbase = sidx(slot0);
                const float4 ro = w_ray[owner], rd = w_ray[64 + owner];
                const f3 O = F3(ro.x, ro.y, ro.z), D = F3(rd.x, rd.y, rd.z);
                // A pair round only has to FIND the spheres whose discriminant (intersections.h:90-93: b*b - a*c, every product and sum rounded)
                // is positive; their hits are then computed literally by resolve().  So the round evaluates -(b*b - a*c) with fused
                // multiply-adds (12 instead of 18 instructions per sphere) and flags what is below a slack that covers the difference between
                // the two evaluations: with S = |oc|^2 both differ from the real value by at most 13 eps (S + r^2) (three-term sums of
                // products, |b| <= sqrt(S a), a = 1), i.e. from each other by 26 eps (S + r^2), eps = 2^-24.  Flagged: nd_fma < kPairSlack
                // (S + r^2) with kPairSlack = 2^-18 = 64 eps; S + r^2 = c + 2 r^2 <= c + 2 r_max^2, folded into the two last operations:
                // nd_fma - slack = (a - kPairSlack) c - (b b + pair_k0).  A flagged sphere that the reference's arithmetic rejects costs one
                // exact test and changes nothing (resolve() returns FLT_MAX for it).
                const float Ak = ro.w - 3.814697265625e-6f;
                groups_done += (uint32_t)(spl >> 2);                 // in units of 4 sphere tests
                for (int k4 = 0; k4 < spl; k4 += 4) {
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) {
                        const float4 sph = S.sph[sbase + k4 + kk];
                        const float ocx = O.x - sph.x;
                        const float ocy = O.y - sph.y;
                        const float ocz = O.z - sph.z;
                        const float b = __builtin_fmaf(ocz, D.z, __builtin_fmaf(ocy, D.y, ocx * D.x));
                        const float c = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocx, ocx, -sph.w)));
                        const float v = __builtin_fmaf(Ak, c, -__builtin_fmaf(b, b, P.pair_k0));
                        mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(v), 31);
                    }
                }
                mask <<= (32 - spl);                                 // slot0 at bit 31
            }
            // The spheres whose discriminant is positive (0..3 of the 16, most often 0 or 1) still need the exact
            // sphereHit tail: IEEE sqrt + divide, 60 instructions.  Resolved in place, the wave would loop max-over-lanes
            // times with a quarter of its lanes busy; instead every lane appends its candidates to a wave-wide LDS list
            // and the list is resolved 64 candidates at a time, one per lane (order-free: the merge is an atomic min).
            append(mask, [&](uint32_t& bits) {
                const int lz = __clz((int)bits);
                bits &= ~(0x80000000u >> lz);
                return ((uint32_t)owner << 24) | (uint32_t)(slot0 + lz);
            });
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            lap(3);
            if (n_c >= 64u) {
                do {
                    n_c -= 64u;
                    resolve(w_cand[n_c + (uint32_t)lane]);
                } while (n_c >= 64u);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // (the next round's appends overwrite what was just read)
                __builtin_amdgcn_wave_barrier();
            }
            lap(4);
        }
        // carry the remainder to the front of the list (nothing to move when no round ran - the common first pass of a 31-group scene: its ~35 pairs
        // already sit at the front - or when nothing is left)
        carry = total - stop;
        if ((!ONEPASS || split) && stop > 0 && carry > 0) {
            unsigned short moved = 0;
            if (lane < carry) moved = w_pair[stop + lane];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < carry) w_pair[lane] = moved;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
      }
        if (flush) win_base = g0 + pass_w;                           // (the list is empty here)
        if (ONEPASS) break;
    }
    {                                                                // the candidates left over
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // (big-sphere candidates when no pair round ran)
        __builtin_amdgcn_wave_barrier();
        lap(2);
        while (n_c > 0u) {                                           // (more than 64 only if no pair round ran after the big spheres)
            const uint32_t take = min(n_c, 64u);
            n_c -= take;
            if ((uint32_t)lane < take) resolve(w_cand[n_c + (uint32_t)lane]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        lap(4);
    }
    const unsigned long long key = w_best[lane];
    Hit out = { FLT_MAX, -1, 0x7fffffff };
    if (key != ~0ull) {
        out.closest = __uint_as_float((uint32_t)(key >> 32));
        out.orig = (int)(uint32_t)key;
        out.sid = S.slot_of[out.orig];
    }
    return out;
}

// ---- closest hit, SPARSE form: at most kSparseRays live rays in the wave ----------------------------------------------
// The tail of the frame belongs to a handful of pixels whose paths bounce up to maxDepth times in the wedge where a sphere rests on the ground
// (thousands of rays, strictly sequential because of the per-pixel RNG stream).  For them the latency of ONE ray is what
// counts, so the wave spends all 64 lanes on each live ray in turn: 16 lanes per group, 4 groups per step; the group
// boxes are tested one group per lane.  Same exact tests, same (t, original index) merge as scan_pairs.  WAVE-LEVEL.
constexpr int kSparseRays = 4;

__device__ __forceinline__ void sparse_test_slot(const RtSphereParams& P, const SceneLds& S, int slot, f3 O, f3 D, float A,
                                                 unsigned long long* best) {
    const float4 sph = S.sph[sidx(slot)];
    // positive discriminants are FOUND with fused multiply-adds and the slack of the pair rounds (scan_pairs), here with the sphere's own
    // radius (big spheres come this way too): flagged iff nd_fma < 2^-18 (S + r^2) = 2^-18 (c + 2 r^2); the hit itself is literal
    const f3 oc = O - F3(sph.x, sph.y, sph.z);
    const float b = __builtin_fmaf(oc.z, D.z, __builtin_fmaf(oc.y, D.y, oc.x * D.x));
    const float c = __builtin_fmaf(oc.z, oc.z, __builtin_fmaf(oc.y, oc.y, __builtin_fmaf(oc.x, oc.x, -sph.w)));
    const float v = __builtin_fmaf(A - 3.814697265625e-6f, c, -__builtin_fmaf(b, b, 7.62939453125e-6f * sph.w));
    if (v < 0.0f) {
        const float t = sphere_hit_exact(sph, O, D, A, P.t_min, FLT_MAX);
        const int o = S.orig[slot];
        if (o != 0x7fffffff && t < FLT_MAX)
            atomicMin(best, ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(uint32_t)o);
    }
}

// All live rays (1..16 of them) are handled TOGETHER: the work items of a phase - (ray, big slot), (ray, group box),
// (reachable (ray, group) pair, sphere of the group) - are numbered across the rays and dealt to the 64 lanes, 64 at a time;
// a lane fetches the ray of its item from the wave's LDS ray table.  Two rays therefore cost three lane passes like one
// ray does (16 + 16 big slots, 31 + 31 boxes, ~2 x 16 spheres), where handling the rays one after the other cost six.
template <bool ONEPASS = false>
__device__ __forceinline__ Hit scan_sparse(const RtSphereParams& P, const SceneLds& S, f3 org, f3 dn, float a, unsigned long long live,
                                           bool cull, unsigned long long* tm = nullptr) {
    // tm (diagnostic instantiation only): cycles in [10] ray table + box set-up, [11] big spheres, [12] group boxes, [13] sphere tests + read-back
    unsigned long long tc = tm ? __builtin_amdgcn_s_memtime() : 0ull;
    auto lap = [&](int k) { if (tm) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tm[k] += n_ - tc; tc = n_; } };
    const int lane = threadIdx.x & 63;
    unsigned char* W = S.scratch + (threadIdx.x >> 6) * ws_total(ONEPASS);
    float4* w_ray = reinterpret_cast<float4*>(W);                               // ray r at w_ray[r], w_ray[64 + r] (as in scan_pairs)
    unsigned long long* w_best = reinterpret_cast<unsigned long long*>(W + 64 * 32);   // best key of ray r at w_best[r]
    unsigned short* w_pair = reinterpret_cast<unsigned short*>(W + 64 * 32 + 64 * 8);  // reachable (ray << 12 | group) pairs
    const bool mine = ((live >> lane) & 1ull) != 0ull;
    const int m = (int)__popcll(live);                               // rays (wave-uniform)
    const int my_r = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
    float4* w_box = w_ray + 16;                                                 // the ray side of the box tests, 3 float4 per ray: <= 16 rays use entries [0, 16) and [64, 80) of the
                                                                                // ray table, so its entries [16, 64) are free
    if (mine) {
        w_ray[my_r] = make_float4(org.x, org.y, org.z, a);
        w_ray[64 + my_r] = make_float4(dn.x, dn.y, dn.z, 0.0f);
        w_best[my_r] = ~0ull;
        const BoxRay b = make_box_ray(P, org, dn, 0.0f);             // once per ray (margin, reciprocals), not once per (ray, group) item
        w_box[3 * my_r] = make_float4(b.inv.x, b.inv.y, b.inv.z, 0.0f);
        w_box[3 * my_r + 1] = make_float4(b.cn.x, b.cn.y, b.cn.z, 0.0f);
        w_box[3 * my_r + 2] = make_float4(b.cf.x, b.cf.y, b.cf.z, 0.0f);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    lap(10);
    // (a) big spheres: items (ray, slot)
    const int nbs = P.n_big_groups << kSphereGroupShift;
    for (int base = 0; base < m * nbs; base += 64) {
        const int w = base + lane;
        if (w < m * nbs) {
            const int r = w / nbs, sl = w - r * nbs;
            const float4 ro = w_ray[r], rd = w_ray[64 + r];
            sparse_test_slot(P, S, sl, F3(ro.x, ro.y, ro.z), F3(rd.x, rd.y, rd.z), ro.w, &w_best[r]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    lap(11);
    // (b) group boxes: items (ray, group); the reachable ones are appended to the pair list
    const int ng = P.n_groups - P.n_big_groups;
    int np = 0;                                                      // pairs in the list (wave-uniform)
    for (int base = 0; base < m * ng; base += 64) {
        const int w = base + lane;
        bool reach = false;
        int r = 0, g = 0;
        if (w < m * ng) {
            r = w / ng;
            g = P.n_big_groups + (w - r * ng);
            if (!cull) {
                reach = true;
            } else {
                const float closest = __uint_as_float((uint32_t)(w_best[r] >> 32));   // FLT_MAX-or-larger bit pattern if none: keeps everything
                const float4 b0 = w_box[3 * r], b1 = w_box[3 * r + 1], b2 = w_box[3 * r + 2];
                BoxRay br;
                br.inv = F3(b0.x, b0.y, b0.z); br.cn = F3(b1.x, b1.y, b1.z); br.cf = F3(b2.x, b2.y, b2.z);
                br.sx = __float_as_uint(b0.x) >> 31; br.sy = __float_as_uint(b0.y) >> 31; br.sz = __float_as_uint(b0.z) >> 31;
                br.cb = (fminf(closest, FLT_MAX) + 1.0e-4f) * 1.00002f;
                reach = box_reach(S.grp + 3 * g, br);
            }
        }
        const unsigned long long rm = __ballot(reach);
        if (reach) {
            const int at = np + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(rm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rm, 0u));
            w_pair[at] = (unsigned short)((r << 12) | g);
        }
        np += (int)__popcll(rm);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    lap(12);
    // (c) the spheres of the reachable pairs: items (pair, sphere of the group)
    for (int base = 0; base < (np << kSphereGroupShift); base += 64) {
        const int w = base + lane;
        if (w < (np << kSphereGroupShift)) {
            const unsigned pr = w_pair[w >> kSphereGroupShift];
            const int r = (int)(pr >> 12), g = (int)(pr & 0xFFFu);
            const float4 ro = w_ray[r], rd = w_ray[64 + r];
            sparse_test_slot(P, S, (g << kSphereGroupShift) + (w & (kSphereGroup - 1)), F3(ro.x, ro.y, ro.z), F3(rd.x, rd.y, rd.z), ro.w, &w_best[r]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    Hit out = { FLT_MAX, -1, 0x7fffffff };
    if (mine) {
        const unsigned long long key = w_best[my_r];
        if (key != ~0ull) {
            out.closest = __uint_as_float((uint32_t)(key >> 32));
            out.orig = (int)(uint32_t)key;
            out.sid = S.slot_of[out.orig];
        }
    }
    lap(13);
    return out;
}

// ---- closest hit, SINGLE-RAY form: the wave has exactly one live ray ------------------------------------------------------
// The frame cannot end before its longest pixel has traced its rays one after the other (one RNG stream per pixel, kernels.cu:541-548; the
// longest pixels of the benchmark frame sit in the wedge where a sphere rests on the ground: ~3700 rays of 50 diffuse bounces each sample,
// three draws per bounce - nothing to overlap), so for such a pixel the LATENCY of one ray step is the whole cost.  The sparse form above
// walks three dependent phases through LDS lists (big spheres -> group boxes -> reachable spheres); with a single ray the shortest dependency
// chain is the brute-force one: the ray is broadcast through SGPRs, every lane runs the fused pre-test on slots lane, lane + 64, ... (8 rounds of
// independent LDS reads for the 488-sphere scene), resolves its own candidates with the literal sphereHit tail, and the per-lane (t, original
// index) keys are merged on the scalar unit.  Same pre-test and slack as sparse_test_slot, same exact tail, same tie rule.  WAVE-LEVEL; q is
// wave-uniform; at most 32 rounds (2048 slots).
__device__ __forceinline__ Hit scan_single(const RtSphereParams& P, const SceneLds& S, int q, f3 org, f3 dn, float a) {
    const int lane = threadIdx.x & 63;
    auto bc = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), q)); };
    const f3 O = F3(bc(org.x), bc(org.y), bc(org.z)), D = F3(bc(dn.x), bc(dn.y), bc(dn.z));
    const float A = bc(a);
    const float Ak = A - 3.814697265625e-6f;
    const int rounds = P.n_padded >> 6;
    uint32_t mask = 0;
#pragma unroll 4
    for (int r = 0; r < rounds; r++) {
        const float4 sph = S.sph[sidx((r << 6) + lane)];
        const f3 oc = O - F3(sph.x, sph.y, sph.z);
        const float b = __builtin_fmaf(oc.z, D.z, __builtin_fmaf(oc.y, D.y, oc.x * D.x));
        const float c = __builtin_fmaf(oc.z, oc.z, __builtin_fmaf(oc.y, oc.y, __builtin_fmaf(oc.x, oc.x, -sph.w)));
        const float v = __builtin_fmaf(Ak, c, -__builtin_fmaf(b, b, 7.62939453125e-6f * sph.w));
        mask |= (__float_as_uint(v) >> 31) << r;                     // flagged: see sparse_test_slot
    }
    Hit h = { FLT_MAX, -1, 0x7fffffff };
    while (mask) {
        const int r = __builtin_ctz(mask);
        mask &= mask - 1u;
        const int k = (r << 6) + lane;
        const float t = sphere_hit_exact(S.sph[sidx(k)], O, D, A, P.t_min, FLT_MAX);
        const int o = S.orig[k];
        if (o != 0x7fffffff && t < FLT_MAX) accept(h, t, k, o);
    }
    // merge: lexicographic minimum of (t, orig) over the lanes that found something (t > t_min >= 0: the bit patterns order like the values)
    unsigned long long found = __ballot(h.closest < FLT_MAX);
    uint32_t st = __float_as_uint(FLT_MAX);
    int sk = -1, so = 0x7fffffff;
    while (found) {
        const int bq = __builtin_ctzll(found);
        found &= found - 1;
        const uint32_t tb = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(h.closest), bq);
        const int ob = __builtin_amdgcn_readlane(h.orig, bq);
        if (tb < st || (tb == st && ob < so)) { st = tb; so = ob; sk = __builtin_amdgcn_readlane(h.sid, bq); }
    }
    Hit out = { __uint_as_float(st), sk, so };
    return out;
}

// ---- shading of one hit / miss: the rest of color()'s loop body (kernels.cu:415-531) -------------------------------
// Returns true when the path ended; `sample` is then the path's colour (the caller adds it to the pixel's sum and starts the next sample).
// The reference's `#ifdef STATS` ray statistics (kernels.cu:47-67,399-432,514-531) for sphere scenes, counted when P.counters is set (an untimed run): a
// sphere scene's spheres are its "mesh" (the oracle counts them the same way), so SECONDARY_MESH = SECONDARY and SECONDARY_NOHIT = 0; there are no
// scene-bounds, shadow or BVH statistics.  One device atomic per wave and event (hipcc folds the per-lane adds of a wave into one).  Compiled into the
// counting / diagnostic instantiation of the persistent kernel only (STATS; the launcher takes it when counters are asked for): in the production one the
// sites would cost registers.
template <bool STATS>
__device__ __forceinline__ void ray_stat(const RtSphereParams& P, int k) { if (STATS && P.counters) atomicAdd(&P.counters->ref_stats[k], 1ull); }

template <bool STATS, bool BASIC = false>
__device__ __forceinline__ bool shade(const RtSphereParams& P, const SceneLds& S, Lane& L, f3 dn, Hit h, f3& sample) {
    sample = F3(0, 0, 0);                                            // kernels.cu:397
    const bool primary = L.bounce == 0;
    if (STATS && P.counters) {                                       // kernels.cu:403-408
        ray_stat<STATS>(P, primary ? RT_STAT_PRIMARY : RT_STAT_SECONDARY);
        if (!primary) ray_stat<STATS>(P, RT_STAT_SECONDARY_MESH);
        if (len(L.atten) < 0.01f) ray_stat<STATS>(P, RT_STAT_LOW_POWER);
    }
    if (h.sid < 0) {
        ray_stat<STATS>(P, primary ? RT_STAT_PRIMARY_NOHITS : RT_STAT_SECONDARY_MESH_NOHIT);      // kernels.cu:414-417
        sample = sample + L.atten * sky_color(P.sky, L.dir);         // kernels.cu:419-425 (0 + x: kept as an addition, -0 would become +0)
        return true;
    }
    if (primary) ray_stat<STATS>(P, RT_STAT_PRIMARY_HIT_MESH);              // kernels.cu:428-432
    const float4 sc4 = S.sph[sidx(h.sid)];
    const float radius = S.rad[h.sid];                               // slot-indexed, like every scene array on the device
    const f3 hp = L.org + h.closest * dn;                            // ray.h:12 point_at_parameter
    f3 normal = (hp - F3(sc4.x, sc4.y, sc4.z)) / radius;             // intersections.h:95
    if (dot(dn, normal) > 0.0f) normal = -normal;                    // kernels.cu:354-355
    const float4 m = S.mat[h.sid];
    Scatter sc;
    material_scatter<BASIC>(sc, h.closest, hp, normal, L.inside, L.dir, S.typ[h.sid], F3(m.x, m.y, m.z), m.w, L.rng);
    L.org = L.org + sc.t * L.dir;                                    // kernels.cu:485-489
    L.dir = sc.wi;
    L.atten = L.atten * sc.throughput;
    L.inside = sc.refracted ? !L.inside : L.inside;
    bool path_done = false;
    if (P.rr && L.bounce > 3) {                                      // kernels.cu:512-527
        const float mx = max3(L.atten);
        if (rnd(L.rng) > mx) {
            ray_stat<STATS>(P, RT_STAT_RUSSIAN_KILL);
            return true;                                             // kernels.cu:517-520 (the bounce counter no longer matters: the path is over)
        } else {
            const float kk = 1.0f / mx;
            L.atten = F3(L.atten.x * kk, L.atten.y * kk, L.atten.z * kk);
        }
    }
    L.bounce++;
    if (L.bounce >= P.max_depth) { ray_stat<STATS>(P, RT_STAT_EXCEED_MAX_BOUNCE); path_done = true; }     // loop bound, kernels.cu:402,529-531
    return path_done;
}

// One iteration of color()'s bounce loop for every lane of the wave that has a ray (`has_ray`).  WAVE-LEVEL: all 64
// lanes must call it together.  With many live lanes each lane scans the sphere list for its own ray; with few
// (the tail of a tile / of the frame, where a handful of pixels in sphere / ground wedges need thousands of rays each) the
// whole wave works on one ray at a time, which cuts the latency of a ray ~30x and with it the critical path.
template <bool LEGACY, bool STATS = false, bool BASIC = false, int OP = 0, bool C3 = false>
__device__ __forceinline__ bool trace_rays(const RtSphereParams& P, const SceneLds& S, Lane& L, bool has_ray, int coop_below, bool cull,
                                           uint32_t& groups_done, uint32_t& boxes_done, f3& sample, int sparse_max = kSparseRays, unsigned long long* tm = nullptr, bool single = false) {
    // tm (diagnostic instantiation only): cycles in [0] ray set-up, [1..4] scan_pairs, [5] shade, [6] sparse scan
    unsigned long long tc = tm ? __builtin_amdgcn_s_memtime() : 0ull;
    auto lap = [&](int k) { if (tm) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tm[k] += n_ - tc; tc = n_; } };
    // ---- hit(), kernels.cu:325-360: the ray is rebuilt from the path, which renormalises the direction
    const f3 dn = unit(L.dir);
    const float a = dot(dn, dn);
    Hit h = { FLT_MAX, -1, 0x7fffffff };
    const unsigned long long live = __ballot(has_ray);
    lap(0);
    if (!LEGACY) {                                                   // default: pair-compacted scan, sparse form for the tail
        // (the sparse form lists its reachable (ray, group) pairs in the wave's pair list: rays x groups must fit it)
        if (single && __popcll(live) == 1 && P.n_padded <= 2048 && sparse_max > 0) {     // one live ray: the shortest dependency chain (scan_single)
            const Hit hq = scan_single(P, S, (int)__builtin_ctzll(live), L.org, dn, a);
            if (has_ray) h = hq;
            if (tm) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tm[6] += n_ - tc; tc = n_; }
        } else
        // (one list of 32 groups: at most 16 rays x 32 groups = 512 entries always fit)
        if (__popcll(live) <= min(sparse_max, 16) && coop_below == -1 && (OP == 1 || (P.n_groups <= 4096 &&
            (int)__popcll(live) * (P.n_groups - P.n_big_groups) <= ws_list_cap(OP > 0)))) { h = scan_sparse<(OP > 0)>(P, S, L.org, dn, a, live, cull, tm); if (tm) tc = __builtin_amdgcn_s_memtime(); }
        else { h = scan_pairs<OP, C3>(P, S, L.org, dn, a, has_ray, cull, groups_done, boxes_done, tm); if (tm) tc = __builtin_amdgcn_s_memtime(); }
    } else if (__popcll(live) >= coop_below) {
        if (has_ray) h = scan_lane_parallel(P, S, L.org, dn, a, groups_done);
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
    if (has_ray) done = shade<STATS, BASIC>(P, S, L, dn, h, sample);
    lap(5);
    return done;
}

// ---- variant 1: one 8x8 pixel tile per wave, the lane keeps its pixel for the whole kernel ------------------------
template <bool LEGACY>
__global__ void __launch_bounds__(kThreads) k_render_spheres_tiles(const RtSphereParams P, int coop_below, int cull) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* s_fb;
    const SceneLds S = stage_scene<true>(P, smem, &s_fb);            // s_fb: kThreads x 3 floats

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
    uint32_t nrays = 0, groups_done = 0, boxes_done = 0;
    bool active = valid && (P.ns > 0);
    if (active) start_pixel(P, L, i, global_row(P.part, lr));

    while (__ballot(active) != 0ull) {                               // wave-uniform loop: idle lanes stay to help
        if (active) nrays++;
        f3 sample;
        const bool done = trace_rays<LEGACY>(P, S, L, active, coop_below, cull != 0, groups_done, boxes_done, sample);
        if (active && done) {
            L.col = L.col + sample;                                  // kernels.cu:558
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
        atomicAdd(&P.counters->exec_tests, (unsigned long long)groups_done * 4ull);     // lane-parallel phase-1 tests executed
        atomicAdd(&P.counters->box_tests, (unsigned long long)boxes_done);
    }
}

// ---- work-order pre-pass ------------------------------------------------------------------------------------------
// The persistent kernel ends with a tail in which every lane finishes the pixel it happens to hold.  That tail is short
// when the LAST pixels handed out are cheap and alike.  A pixel whose centre ray (no lens offset, no jitter) misses every
// sphere is almost surely a sky pixel: one ray per sample, the cheapest and most uniform work there is.  This kernel
// sorts the tile-major pixel indices into three lists — class 0 "the centre ray crosses a glass sphere" (the pixels that
// can need thousands of sequential rays: handed out first, scattered, so they run alongside everything else instead of
// after it), class 1 "hits something else" (scattered), class 2 "sky" (last) — with one atomic per wave and list.  The
// lists only change WHO renders a pixel and WHEN, never the result (the seed depends on the pixel id alone).
// P.queue[4..6] = lengths of the lists.
template <int SCENE>
__global__ void __launch_bounds__(kThreads) k_classify_spheres(const RtSphereParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* unused;
    const SceneLds S = stage_scene<false, SCENE>(P, smem, &unused);
    const int tiles_x = (P.nx + 7) >> 3;
    const int tiles_y = (P.part.local_rows + 7) >> 3;
    const uint32_t total = (uint32_t)tiles_x * (uint32_t)tiles_y * 64u;
    const uint32_t p = blockIdx.x * kThreads + threadIdx.x;
    const uint32_t tile = p >> 6, within = p & 63u;
    const int ty = (int)(tile / (uint32_t)tiles_x), tx = (int)(tile - (uint32_t)ty * (uint32_t)tiles_x);
    const int i = tx * 8 + (int)(within & 7u);
    const int lr = ty * 8 + (int)(within >> 3);
    const bool valid = p < total && i < P.nx && lr < P.part.local_rows;
    bool hits = false, glass = false;
    if (valid) {
        const int j = global_row(P.part, lr);
        const float u = ((float)i + 0.5f) / (float)P.nx, v = ((float)j + 0.5f) / (float)P.ny;
        const f3 org = ld3(P.cam.origin);
        const f3 dn = unit(ld3(P.cam.lower_left_corner) + u * ld3(P.cam.horizontal) + v * ld3(P.cam.vertical) - org);
        for (int k = 0; k < P.n_padded; k++) {
            const float4 sph = S.sph[sidx(k)];
            const f3 oc = org - F3(sph.x, sph.y, sph.z);
            const float b = dot(oc, dn);
            const float c = dot(oc, oc) - sph.w;
            const bool h = (S.orig[k] != 0x7fffffff) && (b * b - c > 0.0f) && (b < 0.0f || c < 0.0f);
            hits = hits || h;
            glass = glass || (h && S.typ[k] == RT_GLASS);
        }
    }
    // class 0: the centre ray crosses a glass sphere (candidates for very long paths) - handed out FIRST
    // class 1: hits something else;  class 2: sky - handed out LAST
    const int cls = glass ? 0 : (hits ? 1 : 2);
    const unsigned long long m0 = __ballot(valid && cls == 0), m1 = __ballot(valid && cls == 1), m2 = __ballot(valid && cls == 2);
    uint32_t b0 = 0, b1 = 0, b2 = 0;
    if ((threadIdx.x & 63) == 0) {
        if (m0) b0 = atomicAdd(P.queue + 4, (uint32_t)__popcll(m0));
        if (m1) b1 = atomicAdd(P.queue + 5, (uint32_t)__popcll(m1));
        if (m2) b2 = atomicAdd(P.queue + 6, (uint32_t)__popcll(m2));
    }
    b0 = __builtin_amdgcn_readfirstlane(b0);
    b1 = __builtin_amdgcn_readfirstlane(b1);
    b2 = __builtin_amdgcn_readfirstlane(b2);
    if (valid) {
        const unsigned long long m = cls == 0 ? m0 : (cls == 1 ? m1 : m2);
        const uint32_t base = cls == 0 ? b0 : (cls == 1 ? b1 : b2);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        P.order[(uint32_t)cls * total + base + rank] = p;
    }
}

// ---- cost-ordered second phase ---------------------------------------------------------------------------------------
// In the reference-stream mode the frame is rendered in two launches of the persistent kernel.  Phase 1 traces the first
// `s_split` samples of every pixel and parks the pixel: RNG state, running colour sum, rays used.  k_order_by_cost then
// sorts the pixels by that measured cost into kCostClasses lists, and phase 2 resumes every pixel's stream exactly where
// it stopped, longest jobs first.  No sample is traced twice and none is traced differently: the stream, the order of the
// samples and the order of the additions into `col` are those of the single-launch kernel.
//
// Cost estimate of a pixel = rays per sample so far, averaged over the pixel and its (up to 8) neighbours: the
// neighbours see almost the same scene, so this is an 18-sample estimate instead of a 2-sample one.  A lane that is
// not on a boosted chain advances one ray per wave iteration, so a pixel handed out at time T ends near
// T + rays x iteration time: the classes are narrow (x1.25 steps) so that the pixels handed out last are the cheapest.
// class 0: >= 6 rays per sample (candidates for the very long chains); class kCostClasses-1: one ray per sample in the
// whole window (sky).  Two passes (PASS 0 counts, PASS 1 fills) make the lists compact: P.order holds each pixel once.
// P.queue[4 + c] = length of list c;  P.queue[4 + kCostClasses + c] = fill cursor of list c.
constexpr int kCostClasses = 18;
constexpr int kChainClasses = 4;      // lists 0..3: >= 12 rays per sample, the chains -> chain waves (see k_render_spheres_queue); list 0 (>= P.chain_top_thr / 16
                                      // rays per sample: ~100 of the 960 k pixels of the benchmark frame, among them the 45 that need > 3000 rays) has waves
                                      // that hold ONE pixel (kernel parameter `caps`) and trace it in the single-ray form (scan_single)
constexpr int kHeavyClasses = 3;      // lists 4..6: 6..12 rays per sample -> spread over the first fill of the normal waves

__device__ __forceinline__ int cost_class(const RtSphereParams& P, int i, int lr) {
    uint32_t sum = 0, cnt = 0;
    for (int dy = -1; dy <= 1; dy++) {
        const int y = lr + dy;
        if (y < 0 || y >= P.part.local_rows) continue;
        for (int dx = -1; dx <= 1; dx++) {
            const int x = i + dx;
            if (x < 0 || x >= P.nx) continue;
            sum += P.px_rays[(size_t)y * P.nx + x];
            cnt++;
        }
    }
    // e = 16 x (rays per sample); >= 16 always (every sample starts with one ray)
    // the window mean under-rates an isolated long pixel (its neighbours miss the glass): never below 3/4 of the pixel's own rate
    const uint32_t own = (P.px_rays[(size_t)lr * P.nx + i] * 12u) / (uint32_t)P.s_split;
    const uint32_t e = max((sum * 16u) / (cnt * (uint32_t)P.s_split), own);
    const uint32_t lim[kCostClasses - 1] = { (uint32_t)P.chain_top_thr, 320u, 240u, 192u, 160u, 128u, 96u, 72u, 56u, 44u, 36u, 30u, 26u, 22u, 19u, 18u, 17u };
    int cls = kCostClasses - 1;
#pragma unroll
    for (int c = kCostClasses - 2; c >= 0; c--) if (e >= lim[c]) cls = c;
    return cls;
}

__device__ __forceinline__ uint32_t coprime_stride_of(uint32_t n) {   // ~0.618 n, coprime with n
    if (n <= 64u) return 1u;
    uint32_t c = ((uint32_t)((unsigned long long)n * 2654435769ull >> 32)) | 1u;
    for (;;) {
        uint32_t a = c, b = n;
        while (b) { const uint32_t t = a % b; a = b; b = t; }
        if (a == 1u) break;
        c += 2u;
    }
    return c % n;
}

// Each workgroup owns a contiguous range of pixel slots, counts its classes in LDS and touches the global counters once
// per class (a global atomic per wave and class would serialise on 12 addresses: 1.7 ms instead of 40 us).
constexpr int kOrderBlocks = 1024;

template <int PASS>
__global__ void __launch_bounds__(kThreads) k_order_by_cost(const RtSphereParams P) {
    // One set of kCostClasses lists for the machine, or (P.xcd_queues == kXcdQueues) one set per XCD: list (x, c) = cost class c of the pixels that belong to XCD x
    // (rt_xcd_of_pixel), counted into P.queue + x * kXcdQueueWords.  In P.order the sets lie one behind the other (XCD 0's lists, XCD 1's ...); word [3] of a
    // queue block holds the first position of its set.
    constexpr int kLists = kCostClasses * kXcdQueues;
    __shared__ uint32_t s_cnt[kLists], s_base[kLists], s_start[kLists], s_n[kLists], s_stride[kLists];
    const int xq = P.xcd_queues == kXcdQueues ? kXcdQueues : 1;
    const int tiles_x = (P.nx + 7) >> 3;
    const int tiles_y = (P.part.local_rows + 7) >> 3;
    const uint32_t total = (uint32_t)tiles_x * (uint32_t)tiles_y * 64u;
    const uint32_t per = ((total + gridDim.x - 1u) / gridDim.x + (uint32_t)kThreads - 1u) / (uint32_t)kThreads * (uint32_t)kThreads;
    const uint32_t first = blockIdx.x * per, last = min(first + per, total);
    uint32_t packed = 0;                                             // (local row << 16) | column of the pixel class_of() looked at
    auto class_of = [&](uint32_t p) {                                // list index x * kCostClasses + class, or -1
        const uint32_t tile = p >> 6, within = p & 63u;
        const int ty = (int)(tile / (uint32_t)tiles_x), tx = (int)(tile - (uint32_t)ty * (uint32_t)tiles_x);
        const int i = tx * 8 + (int)(within & 7u);
        const int lr = ty * 8 + (int)(within >> 3);
        packed = ((uint32_t)lr << 16) | (uint32_t)i;
        if (!(p < last && i < P.nx && lr < P.part.local_rows)) return -1;
        const int x = xq > 1 ? (int)rt_xcd_of_pixel((uint32_t)lr, (uint32_t)P.nx, (uint32_t)i) : 0;
        return x * kCostClasses + cost_class(P, i, lr);
    };
    if (threadIdx.x < kLists) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t p = first + threadIdx.x; p < last; p += kThreads) {
        const int cls = class_of(p);
        if (cls >= 0) atomicAdd(&s_cnt[cls], 1u);
    }
    __syncthreads();
    const int lx = (int)threadIdx.x / kCostClasses, lc = (int)threadIdx.x - lx * kCostClasses;      // the list this thread looks after
    uint32_t* const Q = P.queue + (size_t)lx * kXcdQueueWords;
    if (PASS == 0) {
        if ((int)threadIdx.x < xq * kCostClasses && s_cnt[threadIdx.x]) atomicAdd(Q + 4 + lc, s_cnt[threadIdx.x]);
        return;
    }
    // The list entries are written where the render kernel will read them: position = start + (rank x stride) mod n, the
    // multiplicative permutation that scatters neighbouring pixels over different waves (stride ~ 0.618 n, coprime with n;
    // 1 for the last, sky list), and hold the pixel as (local row << 16 | column) - so that fetching a pixel costs the render
    // kernel one load, not a 64-bit modulo and a division by the tile count.
    if ((int)threadIdx.x < xq * kCostClasses) {
        uint32_t start = 0;                                          // first position of this list in P.order
        for (int x = 0; x < lx; x++)
            for (int c = 0; c < kCostClasses; c++) start += P.queue[(size_t)x * kXcdQueueWords + 4 + c];
        if (lc == 0) Q[3] = start;                                   // (every workgroup writes the same value)
        for (int c = 0; c < lc; c++) start += Q[4 + c];
        const uint32_t n = Q[4 + lc];
        s_start[threadIdx.x] = start;
        s_n[threadIdx.x] = n;
        s_stride[threadIdx.x] = (lc == kCostClasses - 1) ? 1u : coprime_stride_of(n);
        s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(Q + 4 + kCostClasses + lc, s_cnt[threadIdx.x]) : 0u;   // first rank of this workgroup
        s_cnt[threadIdx.x] = 0u;
    }
    __syncthreads();
    for (uint32_t p = first + threadIdx.x; p < last; p += kThreads) {
        const int cls = class_of(p);
        if (cls >= 0) {
            const uint32_t rank = s_base[cls] + atomicAdd(&s_cnt[cls], 1u);
            const uint32_t at = s_start[cls] + (uint32_t)(((unsigned long long)rank * s_stride[cls]) % s_n[cls]);
            const size_t px = (size_t)(packed >> 16) * P.nx + (packed & 0xFFFFu);
            if (P.ord_rec) {                                         // one 32-byte record per queue position (RtSphereParams::ord_rec)
                P.ord_rec[2 * (size_t)at] = P.px_state[px];
                P.ord_rec[2 * (size_t)at + 1] = make_float4(__uint_as_float(packed), __uint_as_float(P.px_rays[px]), 0.0f, 0.0f);
            } else {
                P.order[at] = packed;
                P.ord_state[at] = P.px_state[px];                    // the parked state travels with the list entry (see RtSphereParams::ord_state)
                P.ord_rays[at] = P.px_rays[px];
            }
        }
    }
}

// ---- variant 0 (default): persistent waves + pixel queue ----------------------------------------------------------
// A pixel's samples are sequential (one RNG stream per pixel), so the pixel is the atom of work.  Pixels are
// numbered tile-major (8x8 tiles, row-major tiles, row-major inside a tile) and handed out from ONE global
// counter; in a wave, the lanes whose pixel is finished are counted with a wave64 ballot, ONE atomic reserves that
// many queue positions, and an mbcnt prefix sum gives each idle lane its own position (dense allocation).  Every lane
// therefore traces a ray in (almost) every iteration until the queue is empty; which lane renders which pixel is
// irrelevant to the result because the seed is a function of the global pixel id only.

// Template parameters (one lean instantiation per job instead of one kernel that carries every mode as run-time state:
// the all-in-one version kept 98 SGPRs spilled to VGPR lanes):
//   PHASE    0 = the whole pixel (or sample chunk) in one launch; 1 = first P.s_split samples, then park the pixel;
//            2 = resume every parked pixel from its saved stream position (cost-ordered lists)
//   CLS      work order: 0 = queue position -> pixel by a multiplicative permutation (stride 1 = tile-major);
//            1 = the three lists of k_classify_spheres; 2 = the kCostClasses compact lists of k_order_by_cost, tiered
//            (chain lists -> chain waves, heavy lists spread over the first fill, the rest in descending cost)
//   CHUNKED  RT_RNG_COUNTER: a pixel's samples are independent, P.chunks work items per pixel
//   DBG      diagnostics (RT_WAVE_DEBUG): time stamps and section timers
// cfg: bit 0 cull; bits 8..15 extra sparse-form rays per iteration for lanes on a long chain (boost); bits 16..23 a wave takes the
// sparse form at <= this many live rays.  chain_cfg: bits 0..7 chain waves live in every N-th workgroup; 8..11 chain waves per such
// workgroup; 12..15 pixels a chain wave holds; 16..23 boost threshold (rays per sample); 24..27 number of chain lists; 28..31 pixels a
// chain wave holds while one of them comes from list 0 (the longest chains).
//   SCENE    where the scene is read from (stage_scene): 0 = an LDS copy, 1 = global memory, 2 = test data in the LDS, hit data in global memory
//   LEAN     bit 0: the scene's materials are the three basic ones: lean shading (material_scatter<BASIC>); bit 1: its small groups take ONE list per ray batch
//            behind the cell-table prefilter (scan_pairs<OP>: at most 32 groups, 64 with bit 4, 128 with bit 5); bit 3: that prefilter is the general 3-axis one (no shared
//            vertical extent: spheres scattered in space); only with SCENE = 0 (a few kinds also with SCENE = 2) and without DBG.  Bits 0 + 1 need ~100 VGPRs instead of 128; bit 2 (with
//            both): compiled for SIX waves per SIMD (80 VGPRs, ~20 of them spilled) and launched as two 12-wave workgroups per CU - see launch_spheres for when
template <int PHASE, int CLS, bool CHUNKED, bool DBG, int SCENE = 0, int LEAN = 0>
__global__ void __launch_bounds__(kThreads, (LEAN & 4) ? 6 : 4) k_render_spheres_queue(const RtSphereParams P, uint32_t stride, int cfg, int chain_cfg, int caps) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* unused;
    const SceneLds S = stage_scene<false, SCENE>(P, smem, &unused);

    const bool cull = (cfg & 1) != 0;
    const int boost = (cfg >> 8) & 0xFF;
    const int sparse_max = (cfg >> 16) & 0xFF;
    const int tiles_x = (P.nx + 7) >> 3;
    const int tiles_y = (P.part.local_rows + 7) >> 3;
    const uint32_t padded = (uint32_t)tiles_x * (uint32_t)tiles_y * 64u;
    // classified order (CLS = 1: the three lists of k_classify_spheres, each `padded` apart in P.order;
    // CLS = 2: the kCostClasses compact lists of k_order_by_cost).  Queue positions walk list 0 (the long chains)
    // first, then the other lists in order; inside a list the order is scattered, except the last (sky) list.
    // Everything the refill needs about the lists lives in LDS (s_q), not in SGPRs: it is read once per refill.
    constexpr int n_cls = CLS == 2 ? kCostClasses : (CLS == 1 ? 3 : 0);
    __shared__ uint32_t s_cls_base[kCostClasses], s_cls_pos[kCostClasses + 1], s_cls_stride[kCostClasses];
    // s_q: 8 words per queue - [0] nA (chain-list pixels)  [1] n0 (heavy-list pixels)  [2] total items of the general queue  [3] spread  [4] spread_ok  [5] middle tier on
    // [6] first position of the queue's lists in P.order.  One queue for the machine, or (CLS = 2 with P.xcd_queues: RtSphereParams::xcd_queues) one per XCD: a wave
    // serves the queue of the XCD it runs on (XCC_ID) and, when that is empty, the general queues of the others in turn (`stolen`); the list tables
    // (s_cls_*) are those of the wave's own XCD - only its chain waves look at them.
    __shared__ uint32_t s_q[8 * kXcdQueues];
    const uint32_t xq = (CLS == 2 && P.xcd_queues == kXcdQueues) ? (uint32_t)kXcdQueues : 1u;
    const uint32_t myx = xq > 1u ? ((uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) & (uint32_t)(kXcdQueues - 1)) : 0u;       // XCC_ID
    if (threadIdx.x < xq) {
        const uint32_t x = threadIdx.x;
        const uint32_t* const Q = P.queue + (size_t)x * kXcdQueueWords;
        const uint32_t seg = CLS == 2 ? Q[3] : 0u;
        // CLS = 2 (tiered): lists [0, n_chain) are the chain lists, the lists up to kChainClasses + kHeavyClasses the
        // heavy lists, the remaining ones the rest.  CLS = 1: no chain lists; list 0 is the heavy list.
        const int n_chain = CLS == 2 ? ((chain_cfg >> 24) & 0xF) : 0;
        const int n_heavy_end = CLS == 2 ? kChainClasses + kHeavyClasses : 1;
        uint32_t pos = 0, nA = 0, n0 = 0;                            // nA: chain-list pixels; n0: heavy pixels
        for (int c = 0; c < n_cls; c++) {
            const uint32_t n = Q[4 + c];
            if (x == myx) {
                s_cls_base[c] = CLS == 2 ? seg + pos : (uint32_t)c * padded;
                s_cls_pos[c] = pos;
                s_cls_stride[c] = (CLS != 1 || c == n_cls - 1) ? 1u : coprime_stride_of(n);       // (CLS 2: the lists are stored permuted)
            }
            if (c < n_chain) nA += n; else if (c < n_heavy_end) n0 += n;
            pos += n;
        }
        if (x == myx) for (int c = n_cls; c <= kCostClasses; c++) s_cls_pos[c] = pos;
        const uint32_t total_px = n_cls ? pos - nA : padded;         // pixels in the general queue
        const uint32_t n_rest = total_px - n0;
        // lanes that draw from the general queue at t = 0: all (of this XCD), minus the chain waves (which start on the chain lists)
        const uint32_t wgs = xq > 1u ? max(gridDim.x / xq, 1u) : gridDim.x;
        uint32_t spread = wgs * blockDim.x;
        if (nA > 0u) spread -= min(spread - 64u, ((wgs + (uint32_t)(chain_cfg & 0xFF) - 1u) / (uint32_t)(chain_cfg & 0xFF)) * (uint32_t)((chain_cfg >> 8) & 0xF) * 64u);
        uint32_t* const sq = s_q + 8 * x;
        sq[0] = nA; sq[1] = n0; sq[2] = total_px * (CHUNKED ? (uint32_t)P.chunks : 1u); sq[3] = spread;
        sq[4] = (n0 > 0u && n0 <= spread && (spread - n0) <= n_rest) ? 1u : 0u;
        // Middle tier (caps bits 16..19 = waves per workgroup, 20..27 = pixels such a wave holds; CLS = 2 only): the heavy lists are not spread over the normal
        // waves but served, from their own counter (queue word [2]), by "middle" waves that hold only a few pixels - see the role comment below.  The general
        // queue is then the rest lists alone.
        sq[5] = 0u;
        if (CLS == 2 && ((caps >> 16) & 0xF) != 0 && n0 > 0u) { sq[5] = 1u; sq[2] = n_rest; sq[4] = 0u; }
        sq[6] = seg;
    }
    __syncthreads();
    // Chain waves.  A lane that is not boosted advances ONE ray per wave iteration, and an iteration takes 3-4 us for <= 4
    // live lanes (sparse form) but 15-25 us for 64: a pixel handed out at time T ends near T + rays x iteration time, and
    // the longest pixels (3700 rays: 50 bounces inside glass, sample after sample) would end the frame at 60 ms.
    // So some waves (role 0: the first waves of every N-th workgroup) serve the chain lists from their own counter
    // (P.queue[1]), longest first, and hold at most kSparseRays pixels: they always run the sparse form, at raised
    // priority, and the longest chain is over after 15 ms.  When the chain lists are empty the wave becomes a normal wave
    // (role 2: general queue, all 64 lanes), but keeps the cap for as long as it still holds a chain pixel.
    // (Tried and dropped: half-occupied "medium" waves for the heavy lists - what they gain in the tail they lose in throughput.)
    // Middle waves (role 1; round 4): pixels of 6-12 rays per sample (the heavy lists) are too many for the chain waves and too long for a dense wave - a lane
    // of a dense wave advances one ray per ~13 us iteration, a 100-spp pixel of 800 rays ends after 10 ms whenever it starts, and the frame with it.  A wave that
    // holds only `mid_cap` pixels runs the same dense form at about half the iteration time (the per-lane phases cost what they cost, the pair rounds shrink
    // with the rays): such pixels end in half the time for ~2.5x the cost per ray, on a few per cent of the frame's rays.
    int role = 2;
    if (CLS == 2 && s_q[8 * myx] > 0u && (int)(threadIdx.x >> 6) < ((chain_cfg >> 8) & 0xF) && (blockIdx.x % (uint32_t)(chain_cfg & 0xFF)) == 0u) role = 0;
    else if (CLS == 2 && s_q[8 * myx + 5] != 0u && (int)(threadIdx.x >> 6) < ((chain_cfg >> 8) & 0xF) + ((caps >> 16) & 0xF)) role = 1;
    const int mid_cap = (caps >> 20) & 0xFF;
    // wave-uniform (a queue per XCD only): how far this wave has moved on from its own XCD's queue.  Even: it draws from the general queue of XCD
    // (myx + stolen / 2) mod 8; odd: that general queue is empty and the wave (role 2) takes what is LEFT ON THAT QUEUE'S CHAIN LISTS.  Chain lists are served
    // by the chain waves of their own XCD; but nothing promises that every XCD got a workgroup of this grid (a grid of six workgroups leaves two without), and
    // at the end of a frame the other XCDs' waves have nothing else to do: they take the leftovers one pixel per grab, as pixels of the last chain class (the
    // wave then holds at most that class's number of pixels, like a chain wave).
    // (Kept as three plain values - xq, myx, stolen.  Packed into one word of bit fields the same logic cost 4.5 % of the frame in EVERY mode, with the same
    // register counts: profiles/r04_ab_qstate.txt.)
    uint32_t stolen = 0;
    int ccls = 7;                       // chain list (0 .. kChainClasses - 1) the lane's pixel came from (fetched while the wave had role 0), 7 = none.  A wave that
                                        // holds a pixel of list c holds at most caps[c] pixels (4 bits each)

    Lane L;
    L.col = F3(0, 0, 0);
    L.org = F3(0, 0, 0); L.dir = F3(0, 0, 1);
    int lr = 0;                         // local row of the lane's pixel (framebuffer row)
    int chunk = 0, s_end_lane = 0;      // the lane's work item: samples [chunk * spw, end) of its pixel
    // (only sample chunks end at a per-lane sample: every other form of the kernel ends all its pixels at the same one - a scalar, not a register per lane)
    auto s_end_of = [&]() -> int { return CHUNKED ? s_end_lane : (PHASE == 1 ? P.s_split : (PHASE == 2 ? P.ns : min(P.ns, P.spw))); };
    uint32_t nrays = 0, groups_done = 0, boxes_done = 0;
    bool have_pixel = false;            // lane owns an unfinished pixel
    uint32_t pix_rays = 0;              // rays traced so far for the lane's current pixel
    bool exhausted = false;             // wave-uniform: the global queue is empty
    uint32_t pool_next = 0, pool_end = 0;   // wave-uniform: the wave's reserved queue positions [pool_next, pool_end)
    float* fbf = reinterpret_cast<float*>(P.fb);
    // Parameters read once per sample are NOT kept in SGPRs for the life of the wave (the camera alone is 22 of them, spilled to VGPR lanes):
    // they are re-read from the device copy of the parameter block through a pointer the optimiser cannot see through (scalar loads, cached).
    auto sample_params = [&]() {
        typedef const __attribute__((address_space(4))) float* ColdF;
        typedef const __attribute__((address_space(4))) int32_t* ColdI;
        unsigned long long base = (unsigned long long)P.self;
        asm volatile("" : "+s"(base));
        SampleParams sp;
        ColdF cf = (ColdF)(base + offsetof(RtSphereParams, cam));
        float* dst = reinterpret_cast<float*>(&sp.cam);
#pragma unroll
        for (int k = 0; k < (int)(sizeof(rt_camera) / 4); k++) dst[k] = cf[k];
        sp.nx = *(ColdI)(base + offsetof(RtSphereParams, nx)); sp.ny = *(ColdI)(base + offsetof(RtSphereParams, ny));
        sp.rng_mode = *(ColdI)(base + offsetof(RtSphereParams, rng_mode));
        return sp;
    };
    // diagnostics (only when wdbg, and only in the DBG instantiation: in the production one they compile away, with
    // their SGPR pressure): 100 MHz time stamps and iteration counts of this wave
    unsigned long long* const wdbg = DBG ? P.wave_dbg : nullptr;
    const unsigned long long dbg_t0 = wdbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const unsigned long long dbg_t0c = wdbg ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long dbg_tex = 0ull;
    uint32_t dbg_iters = 0, dbg_coop_iters = 0, dbg_coop_rays = 0, dbg_maxpix = 0;
    unsigned long long dbg_tm[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };   // section cycles (see trace_rays), [7] refill, [8]/[9] main / boost steps
    float dbg_grab = 0.0f, dbg_p1 = 0.0f;                           // this lane's pixel: time it was grabbed, rays of phase 1

    // end of a path for the lanes in `fin`: accumulate, start the next sample, or store the finished pixel
    // A lane starts a sample in three situations: its previous path ended (finish), it fetched a new pixel, it resumed a
    // parked one.  As three call sites a wave ran start_sample (lens sampling, renormalisation: ~180 instructions) up to three
    // times per iteration; the lanes are flagged instead and ONE site after the refill starts them all.  Only a boosted
    // wave starts the sample at once (`now`), because its heavy lanes trace again within the same iteration.
    bool need_sample = false;
    auto finish = [&](bool fin, bool now, f3 sample) {
        if (fin) {
            if (DBG && P.counters && (isnan(sample.x) || isnan(sample.y) || isnan(sample.z))) ray_stat<DBG>(P, RT_STAT_NAN);      // kernels.cu:559-561
            L.col = L.col + sample;                                  // kernels.cu:558
            L.s++;
            if (L.s < s_end_of()) {
                if (now) start_sample<PHASE == 0>(sample_params(), L); else need_sample = true;
            } else {
                if (PHASE == 1) {                                    // first samples done: park the pixel (RNG state, running sum, cost)
                    const size_t px = (size_t)lr * P.nx + L.i;
                    P.px_state[px] = make_float4(L.col.x, L.col.y, L.col.z, __uint_as_float(L.rng));
                    P.px_rays[px] = pix_rays;
                } else if (!CHUNKED) {
                    const f3 out = L.col / (float)P.ns;              // kernels.cu:568
                    // one 12-byte store (global_store_dwordx3): a lane finishes its pixel on its own, so three dword
                    // stores would be three partial-sector writes
                    *reinterpret_cast<float3*>(fbf + ((size_t)(P.fb_global_rows ? L.j : lr) * P.nx + L.i) * 3) = make_float3(out.x, out.y, out.z);
                } else {                                             // partial sum of this chunk; k_sum_chunks adds them in order
                    float* dst = reinterpret_cast<float*>(P.partial) + (((size_t)lr * P.nx + L.i) * (uint32_t)P.chunks + (uint32_t)chunk) * 3;
                    dst[0] = L.col.x; dst[1] = L.col.y; dst[2] = L.col.z;
                }
                if (wdbg) {
                    dbg_maxpix = max(dbg_maxpix, pix_rays);
                    if (PHASE == 2)                                  // per-pixel time line: (grabbed, finished) in ms, rays in total, rays in phase 1
                        P.px_state[(size_t)lr * P.nx + L.i] = make_float4(dbg_grab, (float)(__builtin_amdgcn_s_memrealtime() - dbg_t0) * 1e-5f, (float)pix_rays, dbg_p1);
                }
                have_pixel = false;
            }
        }
    };

    // PHASE 1 poisons the host framebuffer beside its compute (poison_rows): 11.5 MB over the bus on the benchmark frame.  Not at the head of the kernel - every
    // wave would sit behind its own stores at the first `s_waitcnt vmcnt(0)` (the scene staging), all 4096 at once: 0.11 ms - but from inside the loop,
    // the waves taking turns over the first ~50 iterations: a wave's 2.8 KB are acknowledged long before its next queue grab waits for them.
    uint32_t poison_at = (PHASE == 1 && P.poison_fb) ? 2u + 2u * ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % 24u) : ~0u;
    uint32_t iter_no = 0;
    while (true) {
        if (PHASE == 1 && iter_no++ == poison_at) poison_rows(P, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
        // ---- refill idle lanes --------------------------------------------------------------------------------
        const bool dbg_timers = DBG && wdbg && (cfg & (1 << 29)) == 0;          // RT_WAVE_DEBUG_LIGHT=1: time stamps of waves and pixels only (the section timers
                                                                                 // stretch a sparse step by 40 %: the light form keeps the frame's real proportions)
        const unsigned long long t_refill = dbg_timers ? __builtin_amdgcn_s_memtime() : 0ull;
        while (!exhausted) {
            // live-lane cap of this wave: by its role and by the tiers of the pixels it still holds
            const unsigned long long live_m = __ballot(have_pixel);
            int cap = 64;
            if (CLS == 2 && (role == 1 || __ballot(have_pixel && ccls == 6) != 0ull)) cap = mid_cap;        // a middle wave, or a wave that still holds a middle-tier pixel
            if (CLS == 2 && (role == 0 || __ballot(have_pixel && ccls < 6) != 0ull)) {
                cap = (chain_cfg >> 12) & 0xF;
                if (__ballot(have_pixel && ccls == 2) != 0ull) cap = min(cap, (caps >> 8) & 0xF);
                if (__ballot(have_pixel && ccls == 1) != 0ull) cap = min(cap, (caps >> 4) & 0xF);
                if (__ballot(have_pixel && ccls == 0) != 0ull) cap = min(cap, caps & 0xF);
            }
            const int allowed = cap - (int)__popcll(live_m);
            if (allowed <= 0) break;
            const unsigned long long idle_m = ~live_m;
            const uint32_t idle_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
            const unsigned long long need = __ballot(!have_pixel && idle_rank < (uint32_t)allowed);
            if (need == 0ull) break;
            // Whole pixels: reserve exactly as many as there are idle lanes, so nothing is hoarded in a wave while
            // other waves idle.  Sample chunks (counter RNG) and the 960 k two-sample items of phase 1 are small and plentiful,
            // one atomic per idle lane-group would serialise on the counter, so a wave reserves 128 at a time into a wave-local pool.
            // chain waves take ONE pixel per grab (cfg bit 1): all of them start together, so the grabs interleave and the head of the chain lists - the
            // longest estimates - is dealt one pixel to a wave instead of four neighbours of the list to the first wave that arrives
            const uint32_t cnt = (role == 0 && (cfg & 2)) ? 1u : (uint32_t)__popcll(need);
            const uint32_t qx = xq > 1u ? ((myx + (stolen >> 1)) & (uint32_t)(kXcdQueues - 1)) : 0u;      // the queue this wave draws from now
            const uint32_t* const sq = s_q + 8 * qx;
            const uint32_t total = sq[2];
            if (pool_next >= pool_end) {
                // (PHASE 2, experiments: cfg bits 3..7 x 4 = positions a normal wave reserves at least per grab)
                const bool leftovers = (stolen & 1u) != 0u;             // (role 2 only)
                const uint32_t grab = leftovers ? 1u : ((CHUNKED || PHASE == 1) ? max(cnt, 128u) : ((PHASE == 2 && role == 2) ? max(cnt, (uint32_t)((cfg >> 3) & 0x1F) * 4u) : cnt));
                const uint32_t limit = (role == 0 || leftovers) ? sq[0] : (role == 1 ? sq[1] : total);
                uint32_t b = 0;
                if ((threadIdx.x & 63) == 0) b = atomicAdd(P.queue + (size_t)qx * kXcdQueueWords + ((role == 0 || leftovers) ? 1 : (role == 1 ? 2 : 0)), grab);
                b = __builtin_amdgcn_readfirstlane(b);
                if (b >= limit) {
                    if (role < 2) { role = 2; continue; }            // this wave's lists are empty: a normal wave from now on
                    // a queue per XCD: this queue's general part is empty - what is left on its chain lists; that too - on to the next XCD's queue
                    if (xq > 1u && stolen + 1u < 2u * xq) { stolen++; continue; }
                    exhausted = true;
                    break;
                }
                pool_next = b;
                pool_end = min(b + grab, limit);
            }
            const uint32_t base = pool_next;
            const uint32_t take = min(cnt, pool_end - pool_next);
            pool_next += take;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            if (role == 2 && pool_next >= total && xq == 1u) exhausted = true;   // this grab took the last items (one queue for the machine; with a queue per XCD the failing grabs say so)
            if (!have_pixel && rank < take) {
                const uint32_t item = base + rank;
                uint32_t pos = item;                                 // position in the pixel order; the chunks of a pixel are adjacent items
                if (CHUNKED) { pos = item / (uint32_t)P.chunks; chunk = (int)(item - pos * (uint32_t)P.chunks); }
                uint32_t p, opos = 0;                                // opos: position in P.order (CLS 2): where the pixel's parked state lies as well
                float4 rec0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // (P.ord_rec: the parked state and ray count came with the list entry)
                uint32_t rec_rays = 0;
                if (CLS == 0) {
                    // queue position -> pixel: a multiplicative permutation (stride coprime with total) scatters
                    // neighbouring pixels over different waves, so the few very long pixels (50-bounce paths in the wedge between a sphere and
                    // the ground, clustered along the contact line) never share a wave; stride 1 = tile-major order
                    // (first dispatch, cfg bit 30: the permutation moves ROW SEGMENTS of a tile - 8 adjacent pixels stay together in 8 adjacent lanes, whose parked
                    // states fill one 128-byte line of px_state; `stride` is then coprime with padded / 8)
                    if (PHASE == 1 && (cfg & (1 << 30)) != 0) p = ((uint32_t)(((unsigned long long)(pos >> 3) * stride) % (padded >> 3)) << 3) | (pos & 7u);
                    else p = (uint32_t)(((unsigned long long)pos * stride) % padded);
                } else {
                    // the n0 pixels of the heavy lists are spread evenly over the first `spread` queue positions (= the lanes in
                    // flight at t = 0), so every wave starts with a few of them instead of a few waves with nothing
                    // else; position p is a heavy-list position iff floor((p+1) n0 / spread) > floor(p n0 / spread)
                    uint32_t q;                                      // position in the concatenation of all lists
                    if (role == 0 || (stolen & 1u) != 0u) q = pos;                           // a chain-list position
                    else if (role == 1) q = sq[0] + pos;                                    // the heavy lists, in order (longest estimates first)
                    else if (CLS == 2 && sq[5] != 0u) q = sq[0] + sq[1] + pos;           // middle tier on: the general queue is the rest lists
                    else {
                        const uint32_t nA = sq[0], n0 = sq[1], spread = sq[3];
                        const bool spread_ok = sq[4] != 0u;
                        uint32_t i0, i1;
                        bool is0;
                        if (spread_ok && pos < spread) {
                            const uint32_t before = (uint32_t)(((unsigned long long)pos * n0) / spread);
                            const uint32_t after = (uint32_t)(((unsigned long long)(pos + 1u) * n0) / spread);
                            is0 = after > before; i0 = before; i1 = pos - before;
                        } else if (spread_ok) {
                            is0 = false; i0 = 0; i1 = pos - n0;
                        } else {
                            is0 = pos < n0; i0 = pos; i1 = pos - n0;
                        }
                        q = nA + (is0 ? i0 : n0 + i1);
                    }
                    if (CLS == 2) {
                        // the lists of a queue lie one behind the other in P.order: position = first position of the queue's lists + q, whichever list q falls into
                        opos = sq[6] + q;
                        if (P.ord_rec) {                             // one 32-byte record: parked state | pixel, rays (k_order_by_cost)
                            rec0 = P.ord_rec[2 * (size_t)opos];
                            const float4 rec1 = P.ord_rec[2 * (size_t)opos + 1];
                            p = __float_as_uint(rec1.x);
                            rec_rays = __float_as_uint(rec1.y);
                        } else {
                            p = P.order[opos];                       // already permuted, already (row << 16 | column): k_order_by_cost
                        }
                    } else {
                        int c = 0;
                        for (int k = 1; k < n_cls; k++) if (q >= s_cls_pos[k]) c = k;
                        const uint32_t j = q - s_cls_pos[c];
                        const uint32_t nc = s_cls_pos[c + 1] - s_cls_pos[c];
                        p = P.order[s_cls_base[c] + (uint32_t)(((unsigned long long)j * s_cls_stride[c]) % nc)];
                    }
                }
                int i;
                if (CLS == 2) {
                    i = (int)(p & 0xFFFFu);
                    lr = (int)(p >> 16);
                } else {
                    const uint32_t tile = p >> 6, within = p & 63u;
                    const int ty = (int)(tile / (uint32_t)tiles_x), tx = (int)(tile - (uint32_t)ty * (uint32_t)tiles_x);
                    i = tx * 8 + (int)(within & 7u);
                    lr = ty * 8 + (int)(within >> 3);
                }
                if (i < P.nx && lr < P.part.local_rows) {            // pixels of partial edge tiles are skipped
                    if (PHASE == 0) {
                        s_end_lane = min(P.ns, (chunk + 1) * P.spw);
                        init_pixel(P, L, i, global_row(P.part, lr), chunk * P.spw);
                        pix_rays = 0;
                    } else if (PHASE == 1) {
                        init_pixel(P, L, i, global_row(P.part, lr), 0);
                        pix_rays = 0;
                    } else {                                         // resume: the pixel's stream continues where phase 1 left it
                        const float4 st4 = (CLS == 2 && P.ord_rec) ? rec0 : P.ord_state[opos];
                        L.i = i; L.j = global_row(P.part, lr);
                        L.pixelId = (uint32_t)(L.j * P.nx + i);
                        L.rng = __float_as_uint(st4.w);
                        L.col = F3(st4.x, st4.y, st4.z);
                        L.s = P.s_split;
                        pix_rays = (CLS == 2 && P.ord_rec) ? rec_rays : P.ord_rays[opos];
                        if (wdbg) { dbg_grab = (float)(__builtin_amdgcn_s_memrealtime() - dbg_t0) * 1e-5f; dbg_p1 = (float)pix_rays; }
                    }
                    need_sample = true;
                    have_pixel = true;
                    ccls = 7;
                    if (CLS == 2 && role == 0) { ccls = 0; for (int k = 1; k < kChainClasses; k++) if (pos >= s_cls_pos[k]) ccls = k; }
                    if (CLS == 2 && role == 1) ccls = 6;
                    if (CLS == 2 && role == 2 && (stolen & 1u) != 0u) ccls = kChainClasses - 1;        // a chain pixel nobody had taken
                }
            }
        }
        if (need_sample) { start_sample<PHASE == 0>(sample_params(), L); need_sample = false; }
        const unsigned long long live_now = __ballot(have_pixel);
        if (dbg_timers) dbg_tm[7] += __builtin_amdgcn_s_memtime() - t_refill;
        if (live_now == 0ull) break;                                 // wave-uniform exit: idle lanes stay to help
        if (wdbg) {
            if (exhausted && dbg_tex == 0ull) dbg_tex = __builtin_amdgcn_s_memrealtime();
            dbg_iters++;
            if (__popcll(live_now) <= sparse_max) { dbg_coop_iters++; dbg_coop_rays += (uint32_t)__popcll(live_now); }   // sparse-form iterations / rays
        }

        // ---- one ray per live lane, then `boost` extra rays for the lanes on a long chain ---------------------------------
        // A pixel that keeps needing >= 10 rays per sample is one of the strictly sequential chains (paths caught in
        // a sphere / ground wedge) that decide when the frame ends.  While the wave is busy with 64 rays such a lane would advance one ray
        // per full iteration; it gets `boost` extra rays per iteration, traced in the cheap sparse form (all 64 lanes on
        // one ray).  Scheduling only: the lane consumes its own RNG stream in order, so results do not change.
        // (One call site for both: the scan is large and must not be inlined twice.)
        // (a wave of the middle tier is fast already: no boost steps while it holds such a pixel)
        const int steps = (boost > 0 && __popcll(live_now) > sparse_max && !(CLS == 2 && s_q[8 * myx + 5] != 0u && __ballot(have_pixel && ccls == 6) != 0ull)) ? 1 + boost : 1;
        for (int x = 0; x < steps; x++) {
            bool sel = have_pixel;
            if (x > 0) {
                const uint32_t heavy_thr = (uint32_t)((chain_cfg >> 16) & 0xFF);
                const bool heavy = have_pixel && pix_rays > heavy_thr * (uint32_t)(L.s - chunk * P.spw + 2);
                const unsigned long long hm = __ballot(heavy);
                if (hm == 0ull) break;
                // at most kSparseRays of them per extra step, taking turns: step x serves the heavy lanes of rank
                // [4 (x-1), 4 x) mod their number (all of them when there are <= 4)
                const int h = (int)__popcll(hm);
                const int hr = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                const int start = h > kSparseRays ? ((x - 1) * kSparseRays) % h : 0;
                int d = hr - start;
                if (d < 0) d += h;
                sel = heavy && d < kSparseRays;
            }
            if (sel) { nrays++; pix_rays++; }
            if (x > 0 || steps == 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
            f3 sample;
            const bool done = trace_rays<false, DBG, (LEAN & 1) != 0, (LEAN & 2) ? ((LEAN & 32) ? 4 : ((LEAN & 16) ? 2 : 1)) : 0, (LEAN & 8) != 0>(P, S, L, sel, -1, cull, groups_done, boxes_done, sample, sparse_max, dbg_timers ? dbg_tm : nullptr, (cfg & 4) != 0);
            if (dbg_timers) { dbg_tm[x > 0 ? 9 : 8] += 1ull; }
            finish(done && sel, steps > 1, sample);
        }
    }

    if (PHASE == 1 && poison_at != ~0u && iter_no <= poison_at) poison_rows(P, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);   // (left the loop before its turn)
    // (the launcher runs the DBG instantiation whenever counters are asked for: in the production one the three per-lane counters and their increments are dead code)
    if (DBG && P.counters) {
        atomicAdd(&P.counters->rays, (unsigned long long)nrays);
        atomicAdd(&P.counters->prim_tests, (unsigned long long)nrays * (unsigned long long)P.n);
        atomicAdd(&P.counters->exec_tests, (unsigned long long)groups_done * 4ull);     // lane-parallel phase-1 tests executed
        atomicAdd(&P.counters->box_tests, (unsigned long long)boxes_done);
    }
    if (wdbg) atomicMax(wdbg + 65536ull * 8 - 1, (unsigned long long)dbg_maxpix);      // longest pixel chain of the frame
    if (wdbg && (threadIdx.x & 63) == 0) {
        unsigned long long* w = wdbg + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8;
        w[0] = dbg_t0; w[1] = dbg_tex; w[2] = __builtin_amdgcn_s_memrealtime();
        w[3] = dbg_iters; w[4] = dbg_coop_iters; w[5] = dbg_coop_rays;
        w[6] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID, HW_ID
        w[7] = 1;
        for (int k = 0; k < 8; k++) atomicAdd(wdbg + 65534ull * 8 + k, dbg_tm[k]);      // section cycles, summed over the waves
        if (dbg_coop_iters >= 2000ull) {                                                // the same for the waves that carry the longest chains, + the sparse form's phases
            for (int k = 0; k < 8; k++) atomicAdd(wdbg + 65532ull * 8 + k, dbg_tm[k]);
            for (int k = 0; k < 4; k++) atomicAdd(wdbg + 65531ull * 8 + k, dbg_tm[10 + k]);
            atomicAdd(wdbg + 65531ull * 8 + 4, dbg_coop_iters); atomicAdd(wdbg + 65531ull * 8 + 5, 1ull);
            atomicAdd(wdbg + 65531ull * 8 + 6, __builtin_amdgcn_s_memtime() - dbg_t0c);
        }
        atomicAdd(wdbg + 65533ull * 8 + 0, dbg_tm[8]); atomicAdd(wdbg + 65533ull * 8 + 1, dbg_tm[9]);
    }
}

}  // namespace

// RT_RNG_COUNTER with chunks > 1: fb[pixel] = (chunk_0 + chunk_1 + ...) / ns, chunks added in index order (deterministic)
namespace {
__global__ void __launch_bounds__(256) k_sum_chunks(const RtSphereParams P) {
    const size_t px = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (px >= (size_t)P.part.local_rows * P.nx) return;
    const float* src = reinterpret_cast<const float*>(P.partial) + px * P.chunks * 3;
    f3 col = F3(src[0], src[1], src[2]);
    for (int c = 1; c < P.chunks; c++) col = col + F3(src[3 * c], src[3 * c + 1], src[3 * c + 2]);
    const f3 out = col / (float)P.ns;
    float* dst = reinterpret_cast<float*>(P.fb) + px * 3;
    dst[0] = out.x; dst[1] = out.y; dst[2] = out.z;
}
}  // namespace

constexpr size_t kStaticLds = 1024;          // what the kernels declare statically beside the dynamic allocation (the queue words), rounded up
static size_t lds_bytes(int n_padded, int n, bool with_fb, int scene = 0, int waves = kWavesPerWg, bool onepass = false) {
    // spheres + group bounds (+ material colour + type / original index / radius per slot + slot_of: scene 0; + original index: scene 2),
    // + fb staging (tile kernel only) + the per-wave scratch
    const size_t test_data = (size_t)(n_padded + n_padded / kSphereGroup) * 16 + (size_t)(n_padded / kSphereGroup) * 48 + (size_t)rt_cell_f4(n_padded / kSphereGroup) * 16;
    const size_t scratch = (size_t)waves * ws_total(onepass);
    if (scene == 1) return scratch;
    if (scene == 2) return test_data + (size_t)n_padded * 4 + scratch;
    return test_data + (size_t)n_padded * 16 + (size_t)n_padded * 12 +
           (size_t)((n + 3) & ~3) * 4 + (with_fb ? (size_t)kThreads * 3 * 4 : 0) + scratch;
}

#if defined(RT_MODE_PARITY)
hipError_t rt_order_pixels_by_cost(const RtSphereParams& q, hipStream_t stream) {
    hipLaunchKernelGGL(k_order_by_cost<0>, dim3(kOrderBlocks), dim3(kThreads), 0, stream, q);
    hipLaunchKernelGGL(k_order_by_cost<1>, dim3(kOrderBlocks), dim3(kThreads), 0, stream, q);
    return hipGetLastError();
}
size_t rt_sphere_kernel_lds_bytes(int n_padded, int n) {                       // of the smallest LDS-resident form: beyond it the scene is read from global memory
    return lds_bytes(n_padded, n, false, 2, 8) + kStaticLds;
}
#endif

// variant: bits 0..7   kernel: 0 = persistent waves + pixel queue (default), 1 = one tile per wave;
//          bits 8..15  workgroups per CU of the persistent kernel (0 = default 2);
//          bits 16..23 0 = pair-compacted scan + sparse form (default); 255 = pair-compacted scan only.  Otherwise the earlier hybrid (A/B, tile
//                      kernel only): lane-parallel scan, switching to the wave-cooperative scan when fewer than this many lanes of a wave have a ray
//                      (1 = never cooperative, 65 = always cooperative);
//          bit  26     disable sphere-group culling (every group is scanned: the plain brute-force scan);
//          bits 24..25 work order of the persistent kernel: 0 = two-phase, cost-ordered (reference stream; otherwise as 3),
//                      1 = tile-major, 2 = scattered only, 3 = one launch ordered by the centre-ray pre-pass
//                      (glass-crossing pixels first, sky last).
template <bool CHUNKED>
static hipError_t launch_queue_kernel_global(const RtSphereParams& q, unsigned blocks, hipStream_t stream, uint32_t stride, int cfg, int chain_cfg) {
    const size_t lds = (size_t)kWavesPerWg * kWaveScratch;          // only the per-wave scratch: the scene stays in global memory
    // (counters - and with them the reference's ray statistics - live in the diagnostic instantiation only)
    if (q.counters != nullptr || q.wave_dbg != nullptr) hipLaunchKernelGGL((k_render_spheres_queue<0, 0, CHUNKED, true, 1>), dim3(blocks), dim3(kThreads), lds, stream, q, stride, cfg, chain_cfg, 0x4444);
    else hipLaunchKernelGGL((k_render_spheres_queue<0, 0, CHUNKED, false, 1>), dim3(blocks), dim3(kThreads), lds, stream, q, stride, cfg, chain_cfg, 0x4444);
    return hipGetLastError();
}

static int g_queue_threads = kThreads;      // workgroup size of the persistent kernel for the scene being launched (launch_spheres: 16 waves, or 8 when only that fits)

static int g_lean_dbg = 0;                  // the kind the scene would take without the diagnostics (time lines of the production kernel)
static int g_lean = 0;                      // LEAN bits of the instantiation launch_spheres chose for the scene being launched (0 = the general kernel)

template <int PHASE, int CLS, bool CHUNKED, int SCENE>
static hipError_t launch_queue_kernel_scene(const RtSphereParams& q, unsigned blocks, size_t lds, hipStream_t stream, uint32_t stride, int cfg, int chain_cfg, int caps) {
    // (the diagnostic instantiation - counters, the reference's ray statistics, time stamps - is the general kernel: launch_spheres leaves g_lean at 0 for it)
    const bool counting = q.wave_dbg != nullptr || q.counters != nullptr;
    auto go = [&](auto kern) -> hipError_t {
        // the attribute goes on the function that is launched
        if (lds > 64 * 1024) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(g_queue_threads), lds, stream, q, stride, cfg, chain_cfg, caps);
        return hipGetLastError();
    };
    // (time lines of the PRODUCTION kernel of the benchmark scene - RT_WAVE_DEBUG with RT_WAVE_DEBUG_LIGHT=1, no counters: the lean kind 3 with the stamps)
    if (counting && q.counters == nullptr && SCENE == 0 && !CHUNKED && PHASE != 0 && g_lean_dbg == 3) return go(k_render_spheres_queue<PHASE, CLS, false, true, 0, 3>);
    if (counting) return go(k_render_spheres_queue<PHASE, CLS, CHUNKED, true, SCENE>);
    if (SCENE == 0 && CHUNKED) {                                     // the sample chunks of the counter stream: the kinds of the benchmark's shape
        switch (g_lean) {
        case 1:  return go(k_render_spheres_queue<PHASE, CLS, true, false, 0, 1>);
        case 3:  return go(k_render_spheres_queue<PHASE, CLS, true, false, 0, 3>);
        case 7:  return go(k_render_spheres_queue<PHASE, CLS, true, false, 0, 7>);
        default: break;
        }
    }
    if (SCENE == 0 && !CHUNKED) {
        switch (g_lean) {
        case 1:  return go(k_render_spheres_queue<PHASE, CLS, false, false, 0, 1>);
        case 3:  return go(k_render_spheres_queue<PHASE, CLS, false, false, 0, 3>);
        case 7:  return go(k_render_spheres_queue<PHASE, CLS, false, false, 0, 7>);
        case 11: return go(k_render_spheres_queue<PHASE, CLS, false, false, 0, 11>);
        case 15: return go(k_render_spheres_queue<PHASE, CLS, false, false, 0, 15>);
        case 19: return go(k_render_spheres_queue<PHASE, CLS, false, false, 0, 19>);
        case 27: return go(k_render_spheres_queue<PHASE, CLS, false, false, 0, 27>);
        case 35: return go(k_render_spheres_queue<PHASE, CLS, false, false, 0, 35>);
        case 43: return go(k_render_spheres_queue<PHASE, CLS, false, false, 0, 43>);
        default: break;
        }
    }
    if (SCENE == 2 && !CHUNKED) {                                    // the hybrid scene copy (hit data in global memory: ~1500-3400 spheres)
        switch (g_lean) {
        case 1:  return go(k_render_spheres_queue<PHASE, CLS, false, false, 2, 1>);
        case 35: return go(k_render_spheres_queue<PHASE, CLS, false, false, 2, 35>);
        case 43: return go(k_render_spheres_queue<PHASE, CLS, false, false, 2, 43>);
        default: break;
        }
    }
    return go(k_render_spheres_queue<PHASE, CLS, CHUNKED, false, SCENE>);
}

// `hybrid`: stage_scene's form 2 (test data in the LDS, hit data in global memory)
template <int PHASE, int CLS, bool CHUNKED>
static hipError_t launch_queue_kernel(const RtSphereParams& q, unsigned blocks, size_t lds, bool hybrid, hipStream_t stream, uint32_t stride, int cfg, int chain_cfg, int caps = 0x4444) {
    return hybrid ? launch_queue_kernel_scene<PHASE, CLS, CHUNKED, 2>(q, blocks, lds, stream, stride, cfg, chain_cfg, caps)
                  : launch_queue_kernel_scene<PHASE, CLS, CHUNKED, 0>(q, blocks, lds, stream, stride, cfg, chain_cfg, caps);
}

static hipError_t launch_spheres(const RtSphereParams& p, int variant, hipStream_t stream);

hipError_t RT_LAUNCH_NAME(const RtSphereParams& p, int variant, hipStream_t stream) {
    // p.self: the device copy of the parameter block, owned and refreshed by the renderer (one per device state and frame).  Only fields that are
    // the same for the whole frame of that device state (camera, image size, RNG mode) are read through it.  (Reading the per-pixel fields -
    // framebuffer, parked state, partition - this way too took the kernel from 61 to 43 spilled SGPRs and gained nothing more: 7030 against 7025 Msamples/s.)
    if (!p.self) return hipErrorInvalidValue;
    return launch_spheres(p, variant, stream);
}

static hipError_t launch_spheres(const RtSphereParams& p, int variant, hipStream_t stream) {
    // p.poison_fb: a single-dispatch frame is preceded by k_poison_fb on the same stream; the two-dispatch frame lets its first dispatch do it
    auto wait_fb = [&]() -> hipError_t {
        if (!p.poison_fb) return hipSuccess;
        hipLaunchKernelGGL(k_poison_fb, dim3(512), dim3(256), 0, stream, p);
        return hipGetLastError();
    };
    int kind = variant & 0xFF;
    const int cb_bits = (variant >> 16) & 0xFF;
    const bool legacy = cb_bits != 0 && cb_bits != 255;
    if (legacy) kind = 1;                       // the brute-force A/B scans live in the tile kernel only
    if (p.global_scene) kind = 0;               // scenes beyond the LDS: the persistent kernel only
    const size_t kLdsPerCu = 160 * 1024 - kStaticLds;
    if (lds_bytes(p.n_padded, p.n, true, 0, kWavesPerWg) > kLdsPerCu) kind = 0;        // the tile kernel only knows the full copy
    const int cull = ((variant >> 26) & 1) ? 0 : 1;
    // Lean instantiations of the persistent kernel (template parameter LEAN), chosen by what the scene is: bit 0 = its materials are the three basic ones,
    // bit 1 = its small groups (at most 128) take one list per ray batch behind the cell-table prefilter (bits 4 / 5: two / four words of 32 groups; bit 3:
    // the general 3-axis prefilter, no shared vertical extent).  Bits 0 + 1 need ~100 VGPRs where the general kernel fills its 128 (C5 at 256 spp:
    // 10530 -> 11515 Msamples/s, C2 7800 -> 8580; profiles/r04_ab_lean_c5.txt), and compiled for 80 (bit 2) they run SIX waves per SIMD as two
    // 12-wave workgroups per CU with a scene copy each (the one-list scratch is 1.3 KB per wave smaller: ws_total; workgroups must be a multiple of four
    // waves to pack - a workgroup's waves go round the SIMDs from SIMD 0, two 10-wave workgroups do not fit five per SIMD: tools/mb_occupancy.hip).
    // Six waves buy throughput with latency: +11 % on a 3840x2160 frame, -10 % on a 1200x800 one at 100 AND at 1000 spp - a frame is as long as its
    // throughput or its slowest pixels allow, whichever is longer, and the slowest pixels (6-12 rays per sample in dense waves, one ray per iteration) get
    // slower with every wave that shares the SIMD; both scale with spp, so the pixel count decides: 1920x1080 -4 %, 2560x1440 +9 %, 3200x1800 +10 %
    // (profiles/r04_ab_lean6_sizes.txt).  RT_BASIC=0 / RT_ONEPASS=0 / RT_LEAN6_PIXELS=<n> (0 = never): A/B.
    static const bool basic_env = !(getenv("RT_BASIC") && getenv("RT_BASIC")[0] == '0');
    static const bool onepass_env = !(getenv("RT_ONEPASS") && getenv("RT_ONEPASS")[0] == '0');
    static const long long lean6_pixels = getenv("RT_LEAN6_PIXELS") ? atoll(getenv("RT_LEAN6_PIXELS")) : 2600000ll;
    const bool counting = p.wave_dbg != nullptr || p.counters != nullptr;
    g_lean = 0;
    int lean_wgs = 1;
    g_lean_dbg = 0;
    if (kind == 0 && !p.global_scene && p.basic_materials && basic_env) {
        const int n_small_groups = p.n_groups - p.n_big_groups;
        g_lean = 1;
        if (cull && p.cell_on != 0 && n_small_groups >= 1 && n_small_groups <= 128 && onepass_env) {
            g_lean |= 2;                                                         // one list per ray batch
            if (n_small_groups > 64) g_lean |= 32;                               // ... of four words
            else if (n_small_groups > 32) g_lean |= 16;                          // ... of two
            if (p.box_shared_axis != 2) g_lean |= 8;                             // no shared vertical extent: the 3-axis prefilter
        }
    }
    if (p.chunks > 1 && g_lean != 1 && g_lean != 3) g_lean &= 1;             // (sample chunks: kinds 1, 3 and 7 are built)
    static const bool dbg_light_env = getenv("RT_WAVE_DEBUG_LIGHT") && getenv("RT_WAVE_DEBUG_LIGHT")[0] == '1';
    // (only a frame that takes the two cost-ordered dispatches: its two kernels are the ones instantiated with the stamps)
    const bool two_phase_frame = ((variant >> 24) & 3) == 0 && p.order && p.px_state && p.px_rays && p.chunks == 1 && p.rng_mode == RT_RNG_REFERENCE_STREAM && p.ns >= 8 &&
                                 p.nx <= 65535 && p.part.local_rows <= 65535;
    if (counting) { g_lean_dbg = (g_lean == 3 && p.counters == nullptr && dbg_light_env && two_phase_frame) ? 3 : 0; g_lean = g_lean_dbg == 3 ? 3 : 0; }
    const bool lean_list = (g_lean & 2) != 0;                                    // (the smaller per-wave scratch)
    // The persistent kernel's workgroup is a whole CU's worth of waves (16: the launch bound's 4 per SIMD) around ONE scene copy - 88 KB of per-wave scratch (68
    // in the one-list kernels) leave 72 (92) KB for the scene: the full copy up to ~1200 (1500) spheres (60 bytes per sphere), the hybrid one (what a sphere TEST
    // reads in the LDS, what only a HIT reads in global memory: 21 bytes per sphere) up to ~3400; an 8-wave workgroup (44 KB of scratch, 2 waves per SIMD) keeps
    // the hybrid copy resident up to ~5500 spheres; beyond that the same kernel reads the scene from global memory (p.global_scene, decided by the renderer).
    int waves = kWavesPerWg;
    bool hybrid = false;
    if (kind == 0 && !p.global_scene) {
        if (lds_bytes(p.n_padded, p.n, false, 0, 16, lean_list) <= kLdsPerCu) { hybrid = false; waves = 16; }
        else {                                                                   // hybrid: the lean kinds built for it are 1 and the four-word lists
            hybrid = true;
            if (g_lean != 1 && (g_lean & 32) == 0) g_lean &= 1;
            if (lds_bytes(p.n_padded, p.n, false, 2, 16, (g_lean & 2) != 0) <= kLdsPerCu) waves = 16;
            else { waves = 8; }
        }
    }
    if ((g_lean & 0x32) == 2 && lean6_pixels > 0 && (long long)p.nx * p.part.local_rows >= lean6_pixels &&
        2 * (lds_bytes(p.n_padded, p.n, false, 0, 12, true) + kStaticLds) <= (size_t)160 * 1024) { g_lean |= 4; waves = 12; lean_wgs = 2; }
    g_queue_threads = 64 * waves;
    const size_t lds = kind == 1 ? lds_bytes(p.n_padded, p.n, true, 0, kWavesPerWg) : lds_bytes(p.n_padded, p.n, false, hybrid ? 2 : 0, waves, (g_lean & 2) != 0);
    // bits 27..29: extra sparse-form rays per iteration for lanes on a long chain (0 = default 2, 7 = off)
    const int pb = (variant >> 27) & 7;
    // bits 30..31: a wave switches to the sparse form at <= 4 / 8 / 12 / 16 live rays (0 = default)
    int sparse_max = 4 + 4 * ((variant >> 30) & 3);
    if (cb_bits == 255) sparse_max = 0;         // pair-compacted scan only (A/B): never the sparse form
    const int boost = (pb == 7 || cb_bits == 255) ? 0 : (pb == 0 ? 1 : pb);      // measured on C2: round 2 threshold 8, 2 extra steps: 6300 against 6190 with 10 / 4; round 3, after the dense
                                                                                 // iteration had gained 10 %: 1 extra step 7850, 2: 7650 Msamples/s (profiles/r03_sweep_tail9.txt)
    if (kind == 1) {
        int coop_below = cb_bits;
        if (coop_below == 0) coop_below = -1;      // pair-compacted scan (+ sparse form)
        if (coop_below == 255) coop_below = -2;    // pair-compacted scan only (A/B)
        const void* kern = legacy ? reinterpret_cast<const void*>(k_render_spheres_tiles<true>) : reinterpret_cast<const void*>(k_render_spheres_tiles<false>);
        if (lds > 64 * 1024) {
            const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        const dim3 grid((p.nx + 8 * kWavesPerWg - 1) / (8 * kWavesPerWg), (p.part.local_rows + 7) / 8);
        { const hipError_t ew = wait_fb(); if (ew != hipSuccess) return ew; }
        if (legacy) hipLaunchKernelGGL(k_render_spheres_tiles<true>, grid, dim3(kThreads), lds, stream, p, coop_below, cull);
        else hipLaunchKernelGGL(k_render_spheres_tiles<false>, grid, dim3(kThreads), lds, stream, p, coop_below, cull);
        return hipGetLastError();
    }
    if (!p.queue) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(p.queue, 0, 256, stream);
    if (e != hipSuccess) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int wg_per_cu = (variant >> 8) & 0xFF;
    if (wg_per_cu == 0) wg_per_cu = lean_wgs;   // one workgroup = the residency the kernel's launch bound (4 waves/SIMD, 128 VGPRs) and the LDS allow; the lean kernel: two of 10 waves
    const long long total_px = (long long)((p.nx + 7) / 8) * ((p.part.local_rows + 7) / 8) * 64;
    long long blocks = (long long)cus * wg_per_cu;
    const long long useful = (total_px + g_queue_threads - 1) / g_queue_threads;      // never more lanes than pixels
    if (blocks > useful) blocks = useful;
    if (blocks < 1) blocks = 1;
    // scattered order: stride ~ 0.618 * total, coprime with total
    uint32_t stride = 1;
    int order_mode = (variant >> 24) & 3;
    // Scenes whose hit data (hybrid copy) or whole scene (global) is read from global memory: the scattered single dispatch beats the cost-ordered two
    // dispatches (tools/sweep_scene_sizes.py, 1200x800x50, 16-wave workgroups: 1500 spheres 2819 against 2751, 2000: 2113 / 1696, 2600: 1812 / 1149).
    static const bool hybrid_two = getenv("RT_HYBRID_TWO") && getenv("RT_HYBRID_TWO")[0] == '1';      // A/B: the cost-ordered two dispatches for the hybrid copy too
    if (order_mode == 0 && ((hybrid && !hybrid_two) || p.global_scene)) order_mode = 2;
    if (order_mode != 1 && total_px > 64) {
        auto gcd = [](unsigned long long a, unsigned long long b) { while (b) { const unsigned long long t = a % b; a = b; b = t; } return a; };
        unsigned long long cand = (unsigned long long)((double)total_px * 0.6180339887) | 1ull;
        while (gcd(cand, (unsigned long long)total_px) != 1ull) cand += 2;
        stride = (uint32_t)(cand % (unsigned long long)total_px);
    }
    // chain waves: wave 0 of every workgroup (512 waves) serves the chain lists, kSparseRays pixels to a wave; lanes of
    // normal waves above 10 rays per sample are boosted.  Measured on C2 (flat basin) with the multi-ray sparse form:
    // 512 waves x 4 pixels 5780, x 3: 5740, x 2: 5720; 1024 waves x 2: 5610 Msamples/s (before it: 256 x 4: 5040, 512 x 2: 5540).
    // chain waves: three of the 16 waves of every workgroup (768) serve the chain lists.  They take ONE pixel per grab (the grabs of the waves interleave: the
    // head of the lists - the longest estimates - is dealt one pixel to a wave), a wave that holds a pixel of list 0 (>= 24 rays per sample estimated: ~250 pixels,
    // among them every pixel above 3000 rays) holds nothing else and traces it in the single-ray form, the others hold up to kSparseRays.  Round 3, after
    // the dense iteration had gained 4 %: 7300 against 7117 Msamples/s with 512 waves x 4 pixels (profiles/r03_sweep_tail5.txt, r03_sweep_tail6.txt; a flat
    // basin: 768 waves, list 0 from 20..26 rays per sample, 2..4 pixels for the other lists all within 0.5 %; 1024 waves -2 %, list 0 from 30: -6 %).
    int chain_cfg = 1 | ((waves == 16 ? 3 : (waves >= 10 ? 2 : 1)) << 8) | (kSparseRays << 12) | (8 << 16) | (kChainClasses << 24);
    int caps = 1 | (4 << 4) | (4 << 8) | (4 << 12);     // pixels a chain wave holds while one of them comes from chain list 0 / 1 / 2 / 3
    static const char* mid_env = getenv("RT_MID");       // "waves per workgroup,pixels per wave" of the middle tier (k_render_spheres_queue, role 1); 0 = off
    int mid_waves = RT_MID_WAVES, mid_cap = RT_MID_CAP;
    if (mid_env) sscanf(mid_env, "%d,%d", &mid_waves, &mid_cap);
    if (mid_waves < 0 || mid_waves > 12 || mid_cap < 1 || mid_cap > 64) return hipErrorInvalidValue;
    const int caps_mid = (mid_waves << 16) | (mid_cap << 20);
    caps |= caps_mid;
    int cfg = cull | (boost << 8) | (sparse_max << 16);
    static const bool chain_single = !(getenv("RT_CHAIN_SINGLE") && getenv("RT_CHAIN_SINGLE")[0] == '0');  // chain waves grab one pixel at a time
    static const bool single_ray = !(getenv("RT_SINGLE_RAY") && getenv("RT_SINGLE_RAY")[0] == '0');          // scan_single for waves with one live ray
    // phase 2: a normal wave reserves at least 8 queue positions per grab (one atomic round trip per ~6 finished pixels instead of per ~1: +1.3 % on C2;
    // 16 and more hoard pixels at the end of the frame and lose: 8 -> 7144, 16 -> 6616, 32 -> 6019 Msamples/s, profiles/r03_sweep_pool.txt)
    static const int pool_env = getenv("RT_POOL") ? atoi(getenv("RT_POOL")) : 4;      // (round 4, lean kernel: 4 -> 8517, 8 -> 8420, 16 -> 8110 Msamples/s, profiles/r04_sweep_tune_c2.txt)
    if (chain_single) cfg |= 2;
    if (single_ray) cfg |= 4;
    cfg |= ((pool_env / 4) & 0x1F) << 3;
    static const bool dbg_light = getenv("RT_WAVE_DEBUG_LIGHT") && getenv("RT_WAVE_DEBUG_LIGHT")[0] == '1';
    if (dbg_light) cfg |= 1 << 29;
    if (const char* t = getenv("RT_TUNE")) {        // experiments: "chain_every,chain_waves,heavy_thr,n_chain,boost,chain_pixels,chain_pixels of list 0,1,2"
        int a = 1, b = waves == 16 ? 3 : 1, c = 8, d = kChainClasses, e2 = boost, f = kSparseRays, g = 1, g1 = 4, g2 = 4;
        sscanf(t, "%d,%d,%d,%d,%d,%d,%d,%d,%d", &a, &b, &c, &d, &e2, &f, &g, &g1, &g2);
        // every field is a bit-field of chain_cfg / cfg and some are divisors or loop bounds in the kernel: refuse what does not fit
        if (a < 1 || a > 255 || b < 0 || b > waves || c < 1 || c > 255 || d < 0 || d > 15 || e2 < 0 || e2 > 255 ||
            f < 1 || f > 15 || g < 1 || g > 15 || g1 < 1 || g1 > 15 || g2 < 1 || g2 > 15) {
            fprintf(stderr, "rt error: RT_TUNE=%s out of range (chain_every 1..255, chain_waves 0..%d, heavy_thr 1..255, n_chain 0..15, boost 0..255, "
                            "chain_pixels 1..15, chain_pixels of list 0 / 1 / 2 1..15)\n", t, kWavesPerWg);
            return hipErrorInvalidValue;
        }
        chain_cfg = a | (b << 8) | (f << 12) | (c << 16) | (d << 24);
        caps = g | (g1 << 4) | (g2 << 8) | (f << 12) | caps_mid;
        cfg = cull | (e2 << 8) | (sparse_max << 16) | (chain_single ? 2 : 0) | (single_ray ? 4 : 0) | (((pool_env / 4) & 0x1F) << 3) | (dbg_light ? (1 << 29) : 0);
    }
    const unsigned nb = (unsigned)blocks;
    const unsigned cls_blocks = (unsigned)((total_px + kThreads - 1) / kThreads);
    if (p.global_scene) {                       // single dispatch, scattered order
        e = wait_fb();
        if (e != hipSuccess) return e;
        e = p.chunks > 1 ? launch_queue_kernel_global<true>(p, nb, stream, stride, cfg, chain_cfg) : launch_queue_kernel_global<false>(p, nb, stream, stride, cfg, chain_cfg);
        if (e != hipSuccess) return e;
        if (p.chunks > 1) {
            const size_t npx = (size_t)p.part.local_rows * p.nx;
            hipLaunchKernelGGL(k_sum_chunks, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, stream, p);
        }
        return hipGetLastError();
    }

    // order_mode 0 (default), reference stream, enough samples: two phases — measure the cost of every pixel on its first
    // samples, then resume all pixels longest-first (see k_order_by_cost).  Otherwise: one launch, optionally ordered
    // by the centre-ray pre-pass (k_classify_spheres: order_mode 3) or plainly scattered (2) / tile-major (1).
    const int split = 2;                                             // measured: 1 -> 15.5 ms, 2 -> 15.0, 3 -> 15.1 (round 2); 2 -> 24.5, 4 -> 24.8, 6 -> 25.4 (round 1)
    if (order_mode == 0 && p.order && p.px_state && p.px_rays && p.chunks == 1 && p.rng_mode == RT_RNG_REFERENCE_STREAM && p.ns >= 8 &&
        p.nx <= 65535 && p.part.local_rows <= 65535) {                                   // list entries pack (row << 16 | column)
        RtSphereParams q = p;
        q.phase = 1; q.s_split = split;
        // (q.p1_tile_major 1: the two-sample items in tile-major order - a wave parks two adjacent 8x8 tiles, whole lines of px_state; 2: scattered as row
        // segments of 8 pixels - a line of px_state per 8 lanes, the scattering kept)
        uint32_t stride1 = stride;
        int cfg1 = cfg;
        if (q.p1_tile_major == 1) stride1 = 1u;
        if (q.p1_tile_major == 2 && total_px > 512) {
            auto gcd = [](unsigned long long a, unsigned long long b) { while (b) { const unsigned long long t = a % b; a = b; b = t; } return a; };
            const unsigned long long segs = (unsigned long long)total_px >> 3;
            unsigned long long cand = (unsigned long long)((double)segs * 0.6180339887) | 1ull;
            while (gcd(cand, segs) != 1ull) cand += 2;
            stride1 = (uint32_t)(cand % segs);
            cfg1 |= 1 << 30;
        }
        e = launch_queue_kernel<1, 0, false>(q, nb, lds, hybrid, stream, stride1, cfg1, chain_cfg);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(p.queue, 0, sizeof(uint32_t) * kXcdQueues * kXcdQueueWords, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_order_by_cost<0>, dim3(kOrderBlocks), dim3(kThreads), 0, stream, q);
        hipLaunchKernelGGL(k_order_by_cost<1>, dim3(kOrderBlocks), dim3(kThreads), 0, stream, q);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        q.phase = 2;
        static const bool dbg_phase1 = getenv("RT_WAVE_DEBUG_PHASE") && getenv("RT_WAVE_DEBUG_PHASE")[0] == '1';      // diagnostics: the time line of the FIRST dispatch
        if (dbg_phase1) q.wave_dbg = nullptr;
        return launch_queue_kernel<2, 2, false>(q, nb, lds, hybrid, stream, stride, cfg, chain_cfg, caps);
    }
    bool classified = false;
    if ((order_mode == 0 || order_mode == 3) && p.order != nullptr) {
        const void* cls_kern = hybrid ? reinterpret_cast<const void*>(k_classify_spheres<2>) : reinterpret_cast<const void*>(k_classify_spheres<0>);
        if (lds > 64 * 1024) {
            e = hipFuncSetAttribute(cls_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        if (hybrid) hipLaunchKernelGGL(k_classify_spheres<2>, dim3(cls_blocks), dim3(kThreads), lds, stream, p);
        else hipLaunchKernelGGL(k_classify_spheres<0>, dim3(cls_blocks), dim3(kThreads), lds, stream, p);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        classified = true;
    }
    const bool chunked = p.chunks > 1;
    e = wait_fb();
    if (e != hipSuccess) return e;
    if (classified) e = chunked ? launch_queue_kernel<0, 1, true>(p, nb, lds, hybrid, stream, stride, cfg, chain_cfg) : launch_queue_kernel<0, 1, false>(p, nb, lds, hybrid, stream, stride, cfg, chain_cfg);
    else e = chunked ? launch_queue_kernel<0, 0, true>(p, nb, lds, hybrid, stream, stride, cfg, chain_cfg) : launch_queue_kernel<0, 0, false>(p, nb, lds, hybrid, stream, stride, cfg, chain_cfg);
    if (e != hipSuccess) return e;
    if (chunked) {
        const size_t npx = (size_t)p.part.local_rows * p.nx;
        hipLaunchKernelGGL(k_sum_chunks, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, stream, p);
    }
    return hipGetLastError();
}
