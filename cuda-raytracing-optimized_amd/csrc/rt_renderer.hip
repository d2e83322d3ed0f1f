// rt_renderer.hip — host runtime behind the C-ABI of include/rt_api.h (librt_mi355x.so).
//
// Replaces the host half of /root/reference/kernels.cu: the global RenderContext (:69-145),
// initRenderer (:571-650), runRenderer (:652-664), cleanupRenderer (:666-680) and check_cuda
// (:27-38).  Same contract: one global context, synchronous render, void returns, exit(99) on any
// runtime failure.  MI355X differences:
//   * the framebuffer handed to the caller is pinned host memory; each device renders into a compact
//     device buffer and its stripes are gathered with plain hipMemcpy2DAsync (no managed memory
//     page migration, no RCCL);
//   * the image can be split into interleaved row stripes over several devices of this process
//     (options.devices) and/or over several processes (options.part_rank/part_world);
//   * the BVH is read with plain 16-byte global loads (no texture object);
//   * the kernel is timed with HIP events on the stream it is launched on (getRenderStats).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/rt_api.h"
#include "rt_params.h"

namespace {

#define HIP_CHECK(expr) hip_check((expr), #expr, __FILE__, __LINE__)

void hip_check(hipError_t result, const char* func, const char* file, int line) {   // kernels.cu:27-38
    if (result != hipSuccess) {
        fprintf(stderr, "HIP error = %s at %s:%d '%s' \n", hipGetErrorString(result), file, line, func);
        (void)hipDeviceReset();                                     // kernels.cu:34-35: reset before exiting
        exit(99);
    }
}

[[noreturn]] void rt_fail(const char* msg) {
    fprintf(stderr, "rt error: %s\n", msg);
    exit(99);
}

// Defaults of the traffic forms of the sphere kernel's two-dispatch frame (see runRenderer; each has an environment switch of the same meaning for A/B runs)
constexpr bool kDefaultOrdPacked = true;
constexpr bool kDefaultXcdQueues = true;
constexpr int kDefaultP1Tile = 2;
constexpr bool kDefaultFbDirect = false;
static bool env_flag(const char* name, bool dflt) {     // "0" = off, any other value = on, unset = dflt
    const char* v = getenv(name);
    return v ? v[0] != '0' : dflt;
}

struct DeviceState {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    RtSphereParams* d_params = nullptr; // device copy of the sphere kernel's parameter block (RtSphereParams::self), one per DeviceState
    RtSphereParams* h_params = nullptr; // its pinned staging copy (the source of the asynchronous upload must outlive the call)
    // sphere scene
    float4* d_spheres = nullptr;
    float* d_rad = nullptr;
    float4* d_mat_color = nullptr;
    int32_t* d_mat_type = nullptr;
    float4* d_groups = nullptr;
    int32_t* d_orig = nullptr;
    int32_t* d_slot_of = nullptr;
    // mesh scene
    rt_triangle* d_tris = nullptr;
    float4* d_bvh = nullptr;
    float* d_bvh_axis = nullptr;
    float4* d_leaf_tri = nullptr;
    uint32_t* d_leaf_ofs = nullptr;
    rt_material* d_materials = nullptr;
    std::vector<float*> d_tex;
    float** d_tex_data = nullptr;
    int32_t* d_tex_width = nullptr;
    int32_t* d_tex_height = nullptr;
    // output
    rt_vec3* d_fb = nullptr;
    size_t fb_rows = 0;
    RtCounters* d_counters = nullptr;
    uint32_t* d_queue = nullptr;
    unsigned long long* d_wave_dbg = nullptr;
    uint32_t* d_order = nullptr;
    rt_vec3* d_partial = nullptr;
    size_t partial_bytes = 0;
    float4* d_px_state = nullptr;       // two-phase rendering: per-pixel (col, rng) and rays after the first samples
    uint32_t* d_px_rays = nullptr;
    float4* d_ord_state = nullptr;      // ... and their copies in queue order (RtSphereParams::ord_state / ord_rays)
    uint32_t* d_ord_rays = nullptr;
    float4* d_ord_rec = nullptr;        // ... or as one 32-byte record per queue position (RtSphereParams::ord_rec)
};

struct RenderContext {
    bool initialised = false;
    bool is_spheres = false;
    int nx = 0, ny = 0, max_depth = 0;
    rt_camera cam;
    rt_render_options opt;
    rt_vec3* h_fb = nullptr;            // pinned, nx*ny, handed to the caller
    rt_vec3* h_ext = nullptr;           // caller-owned framebuffer (setExternalFramebuffer), or null
    bool ext_registered = false;        // h_ext is page-locked and device-mapped (hipHostRegister succeeded): the kernels may store into it directly
    // host copies of the scene (so devices can be (re)configured by setRenderOptions)
    std::vector<float4> h_spheres;      // the kernel's sphere image (rt_params.h): (n_padded + n_groups) x (cx, cy, cz, r*r)
    std::vector<float> h_rad;           // n_padded radii
    int global_scene = 0;
    int basic_materials = 0;            // every material is RT_DIFFUSE / RT_METAL / RT_GLASS (RtSphereParams::basic_materials)
    std::vector<float4> h_mat_color;
    std::vector<int32_t> h_mat_type;
    std::vector<float4> h_groups;       // three float4 per group of kSphereGroup slots: per axis (lo, hi, lo, -) of the tight AABB
    float cull_c[3] = { 0, 0, 0 }, cull_radius = 0, cull_k1 = 0, cull_k2 = 0, cull_k3 = 0, cull_coord_max = 0, pair_k0 = 0;
    int box_shared_axis = 0;
    int cell_on = 0;
    float cell_scale[3] = { 0, 0, 0 }, cell_off[3] = { 0, 0, 0 }, ubox[6] = { 0, 0, 0, 0, 0, 0 };
    int cell_axes = 0;
    float box_shared_lo = 0, box_shared_hi = 0;
    std::vector<int32_t> h_orig;        // slot -> caller's sphere index (INT_MAX = pad)
    std::vector<int32_t> h_slot_of;     // caller's sphere index -> slot
    int n_spheres = 0, n_padded = 0, n_groups = 0, n_big_groups = 0, n_big = 0;
    std::vector<rt_triangle> h_tris;
    std::vector<float4> h_bvh;          // numBvhNodes * 24 B viewed as float4 (padded)
    std::vector<float> h_bvh_axis;      // RtMeshParams::bvh_axis
    std::vector<float4> h_leaf_tri;     // RtMeshParams::leaf_tri (empty = not built: sentinels inside leaves, or more than 16 M triangles)
    std::vector<uint32_t> h_leaf_ofs;   // RtMeshParams::leaf_ofs
    int num_bvh_nodes = 0;
    int nppl = 0;
    int leaf_sentinels_trailing = 1;
    rt_bbox bounds;
    rt_plane floor;                     // kernel_scene.floor (helper_structs.h:219): used when rt_render_options.floor = 1
    std::vector<rt_material> h_materials;
    std::vector<std::vector<float>> h_tex;
    std::vector<int32_t> h_tex_w, h_tex_h;
    std::vector<DeviceState> devs;
    rt_render_stats stats;
};

RenderContext g_ctx;     // kernels.cu:145: one global context per process

void default_options(rt_render_options* o, int spheres) {
    memset(o, 0, sizeof *o);
    o->sky = spheres ? RT_SKY_GRADIENT : RT_SKY_CONST_GREY;     // kernels.cu:419-424
    o->nee = spheres ? 0 : 1;                                   // kernels.cu:16 SHADOW
    o->rr = spheres ? 0 : 1;                                    // kernels.cu:14 RUSSIAN_ROULETTE
    o->t_min = spheres ? 0.001f : 0.01f;                        // kernels.cu:19 EPSILON
    o->rng = RT_RNG_REFERENCE_STREAM;
    o->fp = RT_FP_PARITY;
    o->light.center.e[0] = (float)52.514355;                    // kernels.cu:93
    o->light.center.e[1] = (float)715.686951;
    o->light.center.e[2] = (float)-272.620972;
    o->light.radius = 50.0f;
    o->lightColor.e[0] = o->lightColor.e[1] = o->lightColor.e[2] = 20.0f;   // kernels.cu:94
    o->stripe_rows = 8;
    o->num_devices = 0;                                         // 0 = the process's current HIP device
    o->part_rank = 0;
    o->part_world = 1;
}

void free_device(DeviceState& d) {
    HIP_CHECK(hipSetDevice(d.device));
    if (d.stream) HIP_CHECK(hipStreamSynchronize(d.stream));
    auto fr = [](void* p) { if (p) HIP_CHECK(hipFree(p)); };
    fr(d.d_spheres); fr(d.d_rad); fr(d.d_mat_color); fr(d.d_mat_type); fr(d.d_groups); fr(d.d_orig); fr(d.d_slot_of);
    fr(d.d_tris); fr(d.d_bvh); fr(d.d_bvh_axis); fr(d.d_leaf_tri); fr(d.d_leaf_ofs); fr(d.d_materials);
    for (float* t : d.d_tex) fr(t);
    fr(d.d_tex_data); fr(d.d_tex_width); fr(d.d_tex_height);
    fr(d.d_fb); fr(d.d_counters); fr(d.d_queue); fr(d.d_wave_dbg); fr(d.d_order); fr(d.d_partial); fr(d.d_px_state); fr(d.d_px_rays); fr(d.d_ord_state); fr(d.d_ord_rays); fr(d.d_ord_rec);
    fr(d.d_params);
    if (d.h_params) HIP_CHECK(hipHostFree(d.h_params));
    if (d.ev_start) HIP_CHECK(hipEventDestroy(d.ev_start));
    if (d.ev_stop) HIP_CHECK(hipEventDestroy(d.ev_stop));
    if (d.stream) HIP_CHECK(hipStreamDestroy(d.stream));
    d = DeviceState();
}

template <typename T>
T* upload(const std::vector<T>& v) {
    if (v.empty()) return nullptr;
    T* p = nullptr;
    HIP_CHECK(hipMalloc((void**)&p, v.size() * sizeof(T)));
    HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

// Rows of the image owned by partition member `rank` of `world` with stripes of `sr` rows.
int local_rows_of(int ny, int sr, int rank, int world) {
    int rows = 0;
    const int nstripes = (ny + sr - 1) / sr;
    for (int k = rank; k < nstripes; k += world) rows += std::min(sr, ny - k * sr);
    return rows;
}

// (Re)creates the per-device state for the device list in g_ctx.opt.
void setup_devices() {
    RenderContext& c = g_ctx;
    for (DeviceState& d : c.devs) free_device(d);
    c.devs.clear();
    int count = 0;
    HIP_CHECK(hipGetDeviceCount(&count));
    int nd = c.opt.num_devices <= 0 ? 1 : c.opt.num_devices;
    if (nd > RT_MAX_DEVICES) rt_fail("too many devices");
    int current = 0;
    HIP_CHECK(hipGetDevice(&current));
    for (int k = 0; k < nd; k++) {
        DeviceState d;
        d.device = (c.opt.num_devices <= 0) ? current : c.opt.devices[k];
        if (d.device < 0 || d.device >= count) rt_fail("device index out of range");
        HIP_CHECK(hipSetDevice(d.device));
        HIP_CHECK(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreate(&d.ev_start));
        HIP_CHECK(hipEventCreate(&d.ev_stop));
        if (c.is_spheres) {
            HIP_CHECK(hipMalloc((void**)&d.d_params, sizeof(RtSphereParams)));
            HIP_CHECK(hipHostMalloc((void**)&d.h_params, sizeof(RtSphereParams), hipHostMallocDefault));
            d.d_spheres = upload(c.h_spheres);
            d.d_rad = upload(c.h_rad);
            d.d_mat_color = upload(c.h_mat_color);
            d.d_mat_type = upload(c.h_mat_type);
            d.d_groups = upload(c.h_groups);
            d.d_orig = upload(c.h_orig);
            d.d_slot_of = upload(c.h_slot_of);
        } else {
            d.d_tris = upload(c.h_tris);
            d.d_bvh = upload(c.h_bvh);
            d.d_bvh_axis = upload(c.h_bvh_axis);
            d.d_leaf_tri = upload(c.h_leaf_tri);
            d.d_leaf_ofs = upload(c.h_leaf_ofs);
            d.d_materials = upload(c.h_materials);
            const int nt = (int)c.h_tex.size();
            if (nt > 0) {
                for (int t = 0; t < nt; t++) d.d_tex.push_back(upload(c.h_tex[t]));
                std::vector<float*> ptrs(d.d_tex.begin(), d.d_tex.end());
                d.d_tex_data = upload(ptrs);
                d.d_tex_width = upload(c.h_tex_w);
                d.d_tex_height = upload(c.h_tex_h);
            }
        }
        const int world = c.opt.part_world * nd, rank = c.opt.part_rank * nd + k;
        d.fb_rows = (size_t)local_rows_of(c.ny, c.opt.stripe_rows, rank, world);
        if (d.fb_rows > 0) HIP_CHECK(hipMalloc((void**)&d.d_fb, d.fb_rows * c.nx * sizeof(rt_vec3)));
        if (d.fb_rows > 0) {        // work-order lists of the persistent kernels: 3 x (pixels padded to 8x8 tiles)
            const size_t padded = (size_t)((c.nx + 7) / 8) * ((d.fb_rows + 7) / 8) * 64;
            HIP_CHECK(hipMalloc((void**)&d.d_order, 3 * padded * sizeof(uint32_t)));
            {                                                       // the two-dispatch frames of both kernels: parked state + its copy in queue order
                HIP_CHECK(hipMalloc((void**)&d.d_px_state, d.fb_rows * c.nx * sizeof(float4)));
                HIP_CHECK(hipMalloc((void**)&d.d_px_rays, d.fb_rows * c.nx * sizeof(uint32_t)));
                HIP_CHECK(hipMalloc((void**)&d.d_ord_state, padded * sizeof(float4)));
                HIP_CHECK(hipMalloc((void**)&d.d_ord_rays, padded * sizeof(uint32_t)));
                HIP_CHECK(hipMalloc((void**)&d.d_ord_rec, padded * 2 * sizeof(float4)));
            }
        }
        HIP_CHECK(hipMalloc((void**)&d.d_counters, sizeof(RtCounters)));
        HIP_CHECK(hipMemset(d.d_counters, 0, sizeof(RtCounters)));
        HIP_CHECK(hipMalloc((void**)&d.d_queue, sizeof(uint32_t) * kXcdQueues * kXcdQueueWords));          // one block of queue words per XCD (rt_params.h)
        HIP_CHECK(hipMemset(d.d_queue, 0, sizeof(uint32_t) * kXcdQueues * kXcdQueueWords));
        c.devs.push_back(d);
    }
    HIP_CHECK(hipSetDevice(current));
}

void validate_options(const rt_render_options& o) {
    if (o.stripe_rows <= 0 || (o.stripe_rows % 8) != 0) rt_fail("stripe_rows must be a positive multiple of 8");
    if (o.part_world < 1 || o.part_rank < 0 || o.part_rank >= o.part_world) rt_fail("bad part_rank/part_world");
    if (o.num_devices < 0 || o.num_devices > RT_MAX_DEVICES) rt_fail("bad num_devices");
    if (!(o.t_min >= 0.0f)) rt_fail("t_min must be >= 0");
    if (o.sky != RT_SKY_CONST_GREY && o.sky != RT_SKY_GRADIENT) rt_fail("bad sky mode");
    if (o.rng != RT_RNG_REFERENCE_STREAM && o.rng != RT_RNG_COUNTER) rt_fail("bad rng mode");
    if (o.fp != RT_FP_PARITY && o.fp != RT_FP_FAST) rt_fail("bad fp mode");
}

void common_init(const rt_camera& cam, rt_vec3** fb, int nx, int ny, int maxDepth) {
    RenderContext& c = g_ctx;
    if (nx <= 0 || ny <= 0 || (long long)nx * ny > (1ll << 30)) rt_fail("bad image size");
    if (!fb) rt_fail("fb out-parameter is null");
    c.nx = nx; c.ny = ny;
    c.max_depth = maxDepth > 255 ? 255 : maxDepth;      // uint8_t bounce, helper_structs.h:58
    c.cam = cam;
    HIP_CHECK(hipHostMalloc((void**)&c.h_fb, (size_t)nx * ny * sizeof(rt_vec3), hipHostMallocDefault));   // kernels.cu:578-580
    memset(c.h_fb, 0, (size_t)nx * ny * sizeof(rt_vec3));
    *fb = c.h_fb;
    memset(&c.stats, 0, sizeof c.stats);
    setup_devices();
    c.initialised = true;
}


// Device layout of a sphere scene.  The spheres are re-ordered into SLOTS, kSphereGroup (G) slots per group:
//   * "big" spheres (radius > 4 x the median radius: the ground and the three unit spheres of the benchmark scene)
//     come first; their groups are always scanned, by every lane, and give each ray a first `closest`;
//   * "small" spheres are split recursively at medians so that the G slots of a group are neighbours in space; each
//     group of G gets an axis-aligned bounding box, inflated well beyond fp32 rounding (1 % + 1e-4 of the scene
//     extent), which the kernel uses to skip the group for rays that cannot reach it before their current hit;
//   * pad slots fill the last group of each class and the tail up to a multiple of 64 slots; they carry
//     orig = INT_MAX and are never accepted.
// Scanning in slot order instead of the caller's order cannot change the result: the kernel resolves equal-t ties
// by the caller's index (h_orig), which is exactly the reference's first-index-wins rule.
void build_sphere_groups(const rt_sphere* spheres, const rt_material* materials, int n) {
    RenderContext& c = g_ctx;
    std::vector<float> radii(n);
    for (int k = 0; k < n; k++) radii[k] = fabsf(spheres[k].radius);
    std::vector<float> sorted = radii;
    std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
    const float big_above = 4.0f * sorted[n / 2];
    std::vector<int> small, big;
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    for (int k = 0; k < n; k++) {
        if (radii[k] > big_above || !std::isfinite(radii[k])) { big.push_back(k); continue; }
        small.push_back(k);
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], (double)spheres[k].center.e[a]);
            hi[a] = std::max(hi[a], (double)spheres[k].center.e[a]);
        }
    }
    // Groups of G = kSphereGroup: recursive median split of the small spheres along the axis of largest centre extent, the left part
    // rounded to a multiple of G, until a part fits one group.  Compact parts = small boxes = few (ray, group) pairs; a
    // 3D Morton sort (the first version) makes strips and L-shapes when the spheres lie on a plane.  ceil(n / G) groups.
    constexpr size_t G = (size_t)kSphereGroup;
    std::vector<int> ordered;
    std::function<void(std::vector<int>&, size_t, size_t)> split = [&](std::vector<int>& v, size_t b0, size_t e0) {
        const size_t cnt = e0 - b0;
        if (cnt <= (size_t)G) {
            for (size_t q = b0; q < e0; q++) ordered.push_back(v[q]);
            while (ordered.size() % G) ordered.push_back(-1);      // pad this group
            return;
        }
        double l3[3] = { 1e300, 1e300, 1e300 }, h3[3] = { -1e300, -1e300, -1e300 };
        for (size_t q = b0; q < e0; q++)
            for (int a = 0; a < 3; a++) {
                l3[a] = std::min(l3[a], (double)spheres[v[q]].center.e[a]);
                h3[a] = std::max(h3[a], (double)spheres[v[q]].center.e[a]);
            }
        int axis = 0;
        for (int a = 1; a < 3; a++) if (h3[a] - l3[a] > h3[axis] - l3[axis]) axis = a;
        std::stable_sort(v.begin() + b0, v.begin() + e0, [&](int x, int y) { return spheres[x].center.e[axis] < spheres[y].center.e[axis]; });
        const size_t groups = (cnt + G - 1) / G;
        const size_t left = std::min(cnt - 1, (groups / 2) * G);     // a multiple of G: only the last group of the scene is padded
        split(v, b0, b0 + left);
        split(v, b0 + left, e0);
    };
    std::vector<int> slots;                                     // slot -> caller index, -1 = pad
    for (int k : big) slots.push_back(k);                       // big spheres first: groups [0, n_big_groups)
    while (slots.size() % G) slots.push_back(-1);
    const int n_big_groups = (int)slots.size() / G;
    if (!small.empty()) split(small, 0, small.size());
    for (int k : ordered) slots.push_back(k);
    while (slots.size() % 64) slots.push_back(-1);

    double extent = 1.0;
    for (int a = 0; a < 3; a++) if (hi[a] > lo[a]) extent = std::max(extent, hi[a] - lo[a]);
    c.n_spheres = n;
    c.n_padded = (int)slots.size();
    c.n_groups = c.n_padded / G;
    c.n_big_groups = n_big_groups;
    c.n_big = (int)big.size();
    const auto sidx = [](int slot) { return slot + slot / kSphereGroup; };
    c.h_spheres.assign(c.n_padded + c.n_groups, make_float4(0.0f, 3.0e18f, 0.0f, 0.0f));      // pad: radius 0, far away
    c.h_rad.assign(c.n_padded, 0.0f);
    c.h_mat_color.assign(c.n_padded, make_float4(0, 0, 0, 0));
    c.h_mat_type.assign(c.n_padded, RT_DIFFUSE);
    c.h_orig.assign(c.n_padded, INT_MAX);
    c.h_slot_of.assign(n, 0);
    // bounds: 3 float4 per group, one per AXIS: (lo, hi, lo, -) - a ray reads two consecutive floats, at 0 or at 1 by the sign of its direction,
    // and has (near plane, far plane).  Empty group: lo > hi on every axis (never reachable).
    c.h_groups.assign((size_t)c.n_groups * 3, make_float4(3.0e38f, -3.0e38f, 3.0e38f, 0.0f));
    c.basic_materials = 1;
    for (int k = 0; k < n; k++) if (materials[k].type != RT_DIFFUSE && materials[k].type != RT_METAL && materials[k].type != RT_GLASS) c.basic_materials = 0;
    for (int s = 0; s < c.n_padded; s++) {
        const int k = slots[s];
        if (k < 0) continue;
        const float r = spheres[k].radius;
        const float r2 = r * r;                                    // intersections.h:89 radius*radius: one IEEE multiply, the same bits as on the device
        c.h_spheres[sidx(s)] = make_float4(spheres[k].center.e[0], spheres[k].center.e[1], spheres[k].center.e[2], r2);
        c.h_rad[s] = r;
        c.h_mat_color[s] = make_float4(materials[k].color.e[0], materials[k].color.e[1], materials[k].color.e[2], materials[k].param);
        c.h_mat_type[s] = materials[k].type;
        c.h_orig[s] = k;
        c.h_slot_of[k] = s;
    }
    // Group boxes: the tight AABB of the group's spheres, pushed out by one float on conversion.  What makes the culling EXACT is
    // not a static inflation but the per-ray margin the kernel adds (make_box_ray): it covers (a) the rounding of the
    // reference's own fp32 discriminant b*b - a*c, whose error grows like |org - centre|^2 - for a far camera the reference
    // accepts "hits" of rays that geometrically miss a sphere by more than any fixed inflation - and (b) the rounding of the slab test.
    float coord_max = 0.0f;
    double r_min = 1e300;
    float shared_lo[3] = { 0, 0, 0 }, shared_hi[3] = { 0, 0, 0 };
    bool shared_ok[3] = { true, true, true };
    int n_boxes = 0;
    for (int g = n_big_groups; g < c.n_groups; g++) {
        double blo[3] = { 1e300, 1e300, 1e300 }, bhi[3] = { -1e300, -1e300, -1e300 };
        int cnt = 0;
        for (int s = g * G; s < g * G + G; s++) {
            if (slots[s] < 0) continue;
            cnt++;
            r_min = std::min(r_min, (double)radii[slots[s]]);
            for (int a = 0; a < 3; a++) {
                blo[a] = std::min(blo[a], (double)spheres[slots[s]].center.e[a] - radii[slots[s]]);
                bhi[a] = std::max(bhi[a], (double)spheres[slots[s]].center.e[a] + radii[slots[s]]);
            }
        }
        if (cnt == 0) continue;
        float flo[3], fhi[3];
        for (int a = 0; a < 3; a++) {
            flo[a] = std::nextafter((float)blo[a], -INFINITY);
            fhi[a] = std::nextafter((float)bhi[a], INFINITY);
            coord_max = std::max(coord_max, std::max(fabsf(flo[a]), fabsf(fhi[a])));
        }
        for (int a = 0; a < 3; a++) c.h_groups[3 * g + a] = make_float4(flo[a], fhi[a], flo[a], 0.0f);
        for (int a = 0; a < 3; a++) {                                // an axis on which every group box has the same extent?
            if (n_boxes == 0) { shared_lo[a] = flo[a]; shared_hi[a] = fhi[a]; }
            else if (shared_lo[a] != flo[a] || shared_hi[a] != fhi[a]) shared_ok[a] = false;
        }
        n_boxes++;
    }
    c.box_shared_axis = 0;
    for (int a = 2; a >= 0; a--) if (n_boxes > 0 && shared_ok[a]) { c.box_shared_axis = a + 1; c.box_shared_lo = shared_lo[a]; c.box_shared_hi = shared_hi[a]; }
    // Cell tables (rt_params.h, group_needs_cells): for scenes of up to 32 x kCellWordsMax groups, on all three axes.  Bit g of a word = small group g.
    // begins[c] = boxes with lo <= upper edge of cell c, ends[c] = boxes with hi >= lower edge of cell c, both with a slack of kCellSlack cells for
    // the rounding of the device's cell index (x * scale + off in fp32 with |index| <= kCellCount: off by < 2e-5 cells); the last begins-word and the
    // first ends-word hold every box, so that a coordinate beyond the tables' extent - clamped to the first / last cell on the device - rejects
    // nothing it should not.  An axis on which every box has the same extent (spheres resting on a plane: the vertical one) gets no bit in cell_axes:
    // its table could not reject anything.  ubox = the union of the boxes, to which the kernel clips the ray before it looks anything up.
    constexpr double kCellSlack = 1.0e-3;
    c.cell_on = 0;
    c.cell_axes = 0;
    const int cell_words = rt_cell_words(c.n_groups);
    c.h_groups.resize((size_t)c.n_groups * 3 + (size_t)rt_cell_f4(c.n_groups), make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    for (int a = 0; a < 3; a++) { c.ubox[a] = 0.0f; c.ubox[3 + a] = 0.0f; c.cell_scale[a] = 0.0f; c.cell_off[a] = 0.0f; }
    if (n_boxes > 0 && cell_words > 0) {
        const int W = cell_words;
        uint32_t* tab = reinterpret_cast<uint32_t*>(c.h_groups.data() + (size_t)c.n_groups * 3);
        std::vector<char> real(c.n_groups, 0);
        for (int g = n_big_groups; g < c.n_groups; g++) real[g] = c.h_groups[3 * g].x <= c.h_groups[3 * g].y;
        bool ok = true;
        for (int a = 0; a < 3 && ok; a++) {
            double amin = 1e300, amax = -1e300;
            for (int g = n_big_groups; g < c.n_groups; g++) {
                if (!real[g]) continue;
                amin = std::min(amin, (double)c.h_groups[3 * g + a].x);
                amax = std::max(amax, (double)c.h_groups[3 * g + a].y);
            }
            c.ubox[a] = (float)amin; c.ubox[3 + a] = (float)amax;      // (box coordinates are floats: exact)
            const double w = (amax - amin) / kCellCount;
            if (!(w > 1e-30) || !std::isfinite(w) || !std::isfinite(1.0 / w) || !std::isfinite(amin / w)) { ok = false; break; }
            c.cell_scale[a] = (float)(1.0 / w);
            c.cell_off[a] = (float)(-amin / w);
            if (!shared_ok[a]) c.cell_axes |= 1 << a;
            for (int cell = 0; cell < kCellCount; cell++) {
                uint32_t* begins = tab + ((size_t)(2 * a) * kCellCount + cell) * W;
                uint32_t* ends = tab + ((size_t)(2 * a + 1) * kCellCount + cell) * W;
                for (int g = n_big_groups; g < c.n_groups; g++) {
                    if (!real[g]) continue;
                    const int k = g - n_big_groups;
                    if (cell == kCellCount - 1 || (double)c.h_groups[3 * g + a].x <= amin + (cell + 1 + kCellSlack) * w) begins[k >> 5] |= 1u << (k & 31);
                    if (cell == 0 || (double)c.h_groups[3 * g + a].y >= amin + (cell - kCellSlack) * w) ends[k >> 5] |= 1u << (k & 31);
                }
            }
        }
        const bool cells_off = getenv("RT_BOX_CELLS") && getenv("RT_BOX_CELLS")[0] == '0';     // (read at every init: the tests render with and without)
        c.cell_on = (ok && !cells_off) ? 1 : 0;
    }
    // per-ray margin constants
    double cc[3] = { 0, 0, 0 }, rad = 0.0;
    if (!small.empty()) {
        for (int a = 0; a < 3; a++) cc[a] = 0.5 * (lo[a] + hi[a]);
        for (int k : small) {
            double d2 = 0.0;
            for (int a = 0; a < 3; a++) d2 += (spheres[k].center.e[a] - cc[a]) * (spheres[k].center.e[a] - cc[a]);
            rad = std::max(rad, std::sqrt(d2) + radii[k]);
        }
    } else r_min = 1.0;
    const double K_eps = 96.0 * 5.9604645e-8;            // K x 2^-24, see make_box_ray
    for (int a = 0; a < 3; a++) c.cull_c[a] = (float)cc[a];
    c.cull_radius = (float)(rad * 1.000001 + 1e-30);
    c.cull_k1 = (float)(K_eps / (2.0 * std::max(r_min, 1e-30)));
    c.cull_k2 = (float)std::sqrt(K_eps);
    c.cull_k3 = 16.0f * 5.9604645e-8f;
    c.cull_coord_max = coord_max;
    double r_max_small = 0.0;
    for (int k : small) r_max_small = std::max(r_max_small, (double)radii[k]);
    c.pair_k0 = (float)(2.0 * 3.814697265625e-6 * r_max_small * r_max_small * 1.0001);     // 2 x kPairSlack (2^-18) x r_max^2, rounded up
    (void)extent;
}

void cleanup_impl() {
    RenderContext& c = g_ctx;
    for (DeviceState& d : c.devs) free_device(d);
    c.devs.clear();
    if (c.h_ext) { if (c.ext_registered) HIP_CHECK(hipHostUnregister(c.h_ext)); c.h_ext = nullptr; c.ext_registered = false; }
    if (c.h_fb) HIP_CHECK(hipHostFree(c.h_fb));
    c = RenderContext();
}

}  // namespace

extern "C" {

int rtApiVersion(void) { return RT_API_VERSION; }

// sizeof of every struct that crosses this C-ABI, in the order of the RT_SIZEOF_* indices (rt_api.h).  A binding (the ctypes mirror, a cgo / JNI stub)
// compares them with its own view BEFORE the first call that passes one of them: getDefaultRenderOptions and getRenderStats write sizeof(struct)
// bytes through the caller's pointer, so a mirror that is shorter than the library's struct is a heap overrun, not an error message.
int rtStructSizes(int32_t* out, int n) {
    const int32_t sizes[RT_SIZEOF_COUNT] = {
        (int32_t)sizeof(rt_render_options), (int32_t)sizeof(rt_render_stats), (int32_t)sizeof(rt_camera), (int32_t)sizeof(rt_sphere),
        (int32_t)sizeof(rt_material), (int32_t)sizeof(rt_triangle), (int32_t)sizeof(rt_bvh_node), (int32_t)sizeof(rt_mesh),
        (int32_t)sizeof(rt_kernel_scene), (int32_t)sizeof(rt_stexture), (int32_t)sizeof(rt_plane), (int32_t)sizeof(rt_bbox), (int32_t)sizeof(rt_vec3) };
    for (int k = 0; k < n && k < RT_SIZEOF_COUNT; k++) out[k] = sizes[k];
    return RT_SIZEOF_COUNT;
}

int rtDeviceCount(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

void getDefaultRenderOptions(rt_render_options* opt, int is_sphere_scene) {
    if (!opt) return;
    default_options(opt, is_sphere_scene);
}

void initRenderer(const rt_kernel_scene sc, const rt_camera cam, rt_vec3** fb, int nx, int ny, int maxDepth) {
    if (g_ctx.initialised) cleanup_impl();
    RenderContext& c = g_ctx;
    if (!sc.m || !sc.m->tris || !sc.m->bvh || !sc.materials) rt_fail("initRenderer: null scene pointers");
    if (sc.m->numBvhNodes < 4 || (sc.m->numBvhNodes & 1)) rt_fail("initRenderer: numBvhNodes must be even and >= 4");
    if (sc.numPrimitivesPerLeaf <= 0) rt_fail("initRenderer: numPrimitivesPerLeaf must be positive");
    const uint32_t first_leaf = (uint32_t)sc.m->numBvhNodes / 2;                               // kernels.cu:614
    if ((unsigned long long)first_leaf * (unsigned)sc.numPrimitivesPerLeaf > sc.m->numTris)
        rt_fail("initRenderer: leaves * numPrimitivesPerLeaf exceeds numTris (traversal would read out of bounds)");
    if (first_leaf > (1u << 30)) rt_fail("initRenderer: BVH deeper than the 32-bit traversal bit-stack");
    c.is_spheres = false;
    default_options(&c.opt, 0);
    c.h_tris.assign(sc.m->tris, sc.m->tris + sc.m->numTris);                                  // kernels.cu:582-583
    for (const rt_triangle& t : c.h_tris)
        if (t.meshID >= sc.numMaterials && !std::isinf(t.v[0].e[0])) rt_fail("initRenderer: triangle meshID out of range");
    c.num_bvh_nodes = sc.m->numBvhNodes;                                                       // kernels.cu:587-605
    const size_t nfloats = (size_t)c.num_bvh_nodes * 6;
    c.h_bvh.assign((nfloats + 3) / 4 + 1, make_float4(0, 0, 0, 0));
    memcpy(c.h_bvh.data(), sc.m->bvh, nfloats * sizeof(float));
    {   // axis-grouped child-pair records (rt_params.h, bvh_axis): 24 floats per internal node
        const size_t nrec = (size_t)c.num_bvh_nodes / 2;
        c.h_bvh_axis.assign(nrec * 24, 0.0f);
        const float* nodes = reinterpret_cast<const float*>(sc.m->bvh);
        for (size_t i = 0; i < nrec; i++) {
            const float* L = nodes + (2 * i) * 6;
            const float* R = nodes + (2 * i + 1) * 6;
            for (int a = 0; a < 3; a++) {
                float* o = c.h_bvh_axis.data() + i * 24 + a * 8;
                o[0] = L[a]; o[1] = R[a]; o[2] = L[3 + a]; o[3] = R[3 + a];
                o[4] = L[3 + a]; o[5] = R[3 + a]; o[6] = L[a]; o[7] = R[a];
            }
        }
    }
    c.nppl = sc.numPrimitivesPerLeaf;                                                          // kernels.cu:648
    // The leaf loop of kernels.cu:196-214 stops at the first sentinel (inf) triangle of a leaf.  The pair rounds of the mesh
    // kernel test a leaf's triangles in parallel and rely on sentinels being TRAILING (true for every builder that pads
    // leaves at the end); a leaf with a real triangle behind a sentinel sends the kernel to its sequential leaf loop.
    c.leaf_sentinels_trailing = 1;
    for (uint32_t leaf = 0; leaf < first_leaf && c.leaf_sentinels_trailing; leaf++) {
        bool seen = false;
        for (int k = 0; k < c.nppl; k++) {
            const bool sent = std::isinf(c.h_tris[(size_t)leaf * c.nppl + k].v[0].e[0]);
            if (seen && !sent) c.leaf_sentinels_trailing = 0;
            seen = seen || sent;
        }
    }
    // Compact leaf records for the pair rounds (rt_params.h, leaf_tri / leaf_ofs): what triangleHit reads of a triangle and nothing else - v0 and the
    // two edges, e1 = v1 - v0 and e2 = v2 - v0 computed here with the same single fp32 subtraction per component as intersections.h:56-57 (same bits) -
    // for the REAL triangles only.  The caller's 64-byte array stays the ABI of this boundary (helper_structs.h:81-96) and is what a closest hit re-reads.
    c.h_leaf_tri.clear(); c.h_leaf_ofs.clear();
    if (c.leaf_sentinels_trailing && c.nppl <= 255) {
        c.h_leaf_tri.assign((size_t)first_leaf * c.nppl * 3, make_float4(0, 0, 0, 0));
        c.h_leaf_ofs.assign(((size_t)first_leaf + 3) / 4, 0u);
        for (uint32_t leaf = 0; leaf < first_leaf; leaf++) {
            uint32_t cnt = 0;
            for (int k = 0; k < c.nppl; k++) {
                const rt_triangle& t = c.h_tris[(size_t)leaf * c.nppl + k];
                if (std::isinf(t.v[0].e[0])) break;
                volatile float e1[3], e2[3];                         // (volatile: one rounded fp32 subtraction each, never a contracted or widened form)
                for (int a = 0; a < 3; a++) { e1[a] = t.v[1].e[a] - t.v[0].e[a]; e2[a] = t.v[2].e[a] - t.v[0].e[a]; }
                float4* rec = c.h_leaf_tri.data() + ((size_t)leaf * c.nppl + k) * 3;
                rec[0] = make_float4(t.v[0].e[0], t.v[0].e[1], t.v[0].e[2], e1[0]);
                rec[1] = make_float4(e1[1], e1[2], e2[0], e2[1]);
                uint32_t mesh_bits = (uint32_t)t.meshID;
                float mesh_f;
                memcpy(&mesh_f, &mesh_bits, 4);
                rec[2] = make_float4(e2[2], mesh_f, 0.0f, 0.0f);       // (.y: meshID as an integer bit pattern - a closest hit reads its record again for the normal and the material)
                cnt++;
            }
            c.h_leaf_ofs[leaf >> 2] |= cnt << (8 * (leaf & 3));
        }
    }
    c.bounds = sc.m->bounds;
    c.floor = sc.floor;
    c.h_materials.assign(sc.materials, sc.materials + sc.numMaterials);                        // kernels.cu:617-618
    c.h_tex.clear(); c.h_tex_w.clear(); c.h_tex_h.clear();
    for (int t = 0; t < sc.numTextures; t++) {                                                 // kernels.cu:620-645
        const rt_stexture& tx = sc.textures[t];
        if (!tx.data || tx.width <= 0 || tx.height <= 0) rt_fail("initRenderer: bad texture");
        c.h_tex.emplace_back(tx.data, tx.data + (size_t)tx.width * tx.height * 3);
        c.h_tex_w.push_back(tx.width);
        c.h_tex_h.push_back(tx.height);
    }
    for (const rt_material& m : c.h_materials)
        if (m.texId != -1 && (m.texId < 0 || m.texId >= sc.numTextures)) rt_fail("initRenderer: material texId out of range");
    common_init(cam, fb, nx, ny, maxDepth);
}

void initRendererSpheres(const rt_sphere* spheres, const rt_material* materials, int n,
                         const rt_camera cam, rt_vec3** fb, int nx, int ny, int maxDepth) {
    if (g_ctx.initialised) cleanup_impl();
    RenderContext& c = g_ctx;
    if (!spheres || !materials || n <= 0) rt_fail("initRendererSpheres: empty scene");
    c.is_spheres = true;
    default_options(&c.opt, 1);
    for (int k = 0; k < n; k++)
        if (materials[k].type < RT_DIFFUSE || materials[k].type >= RT_MATERIAL_TYPE_COUNT) rt_fail("initRendererSpheres: bad material type");
    build_sphere_groups(spheres, materials, n);
    // Scenes up to ~2100 spheres live in the LDS of every workgroup; larger ones are read from global memory (they stay in L2) by the
    // same kernel (no cost-ordered second phase, no sparse form beyond 4096 groups: slower per ray, same image).
    c.global_scene = rt_sphere_kernel_lds_bytes(c.n_padded, n) > 160 * 1024 ? 1 : 0;
    if (c.n_padded > (1 << 24)) rt_fail("initRendererSpheres: more than 16 M sphere slots");
    common_init(cam, fb, nx, ny, maxDepth);
}

void setRenderOptions(const rt_render_options* opt) {
    RenderContext& c = g_ctx;
    if (!c.initialised) rt_fail("setRenderOptions before init");
    if (!opt) rt_fail("setRenderOptions: null");
    validate_options(*opt);
    const rt_render_options old = c.opt;
    c.opt = *opt;
    bool relayout = old.stripe_rows != opt->stripe_rows || old.part_rank != opt->part_rank ||
                    old.part_world != opt->part_world || old.num_devices != opt->num_devices;
    for (int k = 0; k < RT_MAX_DEVICES && !relayout; k++) relayout = old.devices[k] != opt->devices[k];
    if (relayout) setup_devices();
}

void runRenderer(int ns, int tx, int ty) {
    (void)tx; (void)ty;     // CUDA block shape of the reference (main.cpp:69-70); the wave64 tile is fixed
    RenderContext& c = g_ctx;
    if (!c.initialised) rt_fail("runRenderer before init");
    if (ns <= 0) rt_fail("runRenderer: ns must be positive");
    const auto t0 = std::chrono::steady_clock::now();
    int current = 0;
    HIP_CHECK(hipGetDevice(&current));
    const int nd = (int)c.devs.size();
    const int world = c.opt.part_world * nd;
    const size_t row_bytes = (size_t)c.nx * sizeof(rt_vec3);
    int64_t samples = 0;
    int launches = 0;

    for (int k = 0; k < nd; k++) {
        DeviceState& d = c.devs[k];
        if (d.fb_rows == 0) continue;
        HIP_CHECK(hipSetDevice(d.device));
        RtPartition part;
        part.stripe_rows = c.opt.stripe_rows;
        part.rank = c.opt.part_rank * nd + k;
        part.world = world;
        part.local_rows = (int)d.fb_rows;
        if (c.opt.counters) HIP_CHECK(hipMemsetAsync(d.d_counters, 0, sizeof(RtCounters), d.stream));
        // Finished pixels of the default sphere kernel go straight to the pinned host framebuffer (12 bytes each, spread over the whole frame time):
        // no device-to-host copy after the kernel.  RT_FB_DIRECT=0 keeps the compact device buffer + copy (every other kernel always does).
        const bool fb_direct_env = env_flag("RT_FB_DIRECT", kDefaultFbDirect);
        int spw = ns, chunks = 1;               // sphere path, RT_RNG_COUNTER: samples per work item, work items per pixel
        if (c.is_spheres && c.opt.rng == RT_RNG_COUNTER && (c.opt.variant & 0xFF) == 0) {
            // default: 4 samples per item, but no more items than fill and balance the machine (~32 M): a 3840x2160x4096spp frame cut into 4-sample items
            // would be 8.5 G items and a 100 GB buffer of partial sums - it takes 3 items per pixel instead
            int want = c.opt.samples_per_item;
            if (want <= 0) {
                const long long pixels = std::max<long long>(1, (long long)d.fb_rows * c.nx);
                const long long max_chunks = std::max<long long>(1, (32ll << 20) / pixels);
                want = (int)std::max<long long>(4, (ns + max_chunks - 1) / max_chunks);
            }
            if (want < ns) { spw = want; chunks = (ns + want - 1) / want; }
        }
        const int vk = c.opt.variant & 0xFF, vcb = (c.opt.variant >> 16) & 0xFF;
        // (an external framebuffer that could not be page-locked is reached by the copy path only: see setExternalFramebuffer)
        const bool fb_direct = c.is_spheres && c.max_depth > 0 && fb_direct_env && vk == 0 && (vcb == 0 || vcb == 255) && chunks == 1 &&
                               (!c.h_ext || c.ext_registered);
        rt_vec3* const h_target = c.h_ext ? c.h_ext : c.h_fb;
        // Poison the framebuffer the kernel WRITES (all-ones = NaN): every pixel is written exactly once per frame, so a pixel the work
        // distribution lost shows up as NaN instead of as last frame's (correct-looking) value.  The compact device buffer is filled on the
        // render stream before the timed window; with direct delivery the HOST framebuffer's rows of this partition member are filled by the
        // device itself (RtSphereParams::poison_fb): by the frame's first dispatch, which stores no pixel (the bus writes ride beside its
        // compute: nothing in the frame), or by a small kernel in front of a single dispatch.
        if (!fb_direct && d.fb_rows > 0) HIP_CHECK(hipMemsetAsync(d.d_fb, 0xFF, d.fb_rows * row_bytes, d.stream));
        HIP_CHECK(hipEventRecord(d.ev_start, d.stream));
        if (c.max_depth <= 0) {
            HIP_CHECK(hipMemsetAsync(d.d_fb, 0, d.fb_rows * row_bytes, d.stream));     // loop of kernels.cu:402 never runs
        } else if (c.is_spheres) {
            RtSphereParams p;
            memset(&p, 0, sizeof p);
            p.cam = c.cam; p.nx = c.nx; p.ny = c.ny; p.ns = ns; p.max_depth = c.max_depth;
            p.n = c.n_spheres; p.n_padded = c.n_padded; p.n_groups = c.n_groups; p.n_big_groups = c.n_big_groups; p.n_big = c.n_big;
            p.spheres = d.d_spheres; p.rad = d.d_rad; p.global_scene = c.global_scene; p.basic_materials = c.basic_materials; p.mat_color = d.d_mat_color; p.mat_type = d.d_mat_type;
            p.groups = d.d_groups; p.orig = d.d_orig; p.slot_of = d.d_slot_of;
            p.cull_cx = c.cull_c[0]; p.cull_cy = c.cull_c[1]; p.cull_cz = c.cull_c[2]; p.cull_radius = c.cull_radius;
            p.cull_k1 = c.cull_k1; p.cull_k2 = c.cull_k2; p.cull_k3 = c.cull_k3; p.cull_coord_max = c.cull_coord_max; p.pair_k0 = c.pair_k0; p.box_shared_axis = c.box_shared_axis; p.box_shared_lo = c.box_shared_lo; p.box_shared_hi = c.box_shared_hi;
            p.cell_on = c.cell_on; p.cell_axes = c.cell_axes;
            for (int q = 0; q < 3; q++) { p.cell_scale[q] = c.cell_scale[q]; p.cell_off[q] = c.cell_off[q]; p.ubox[q] = c.ubox[q]; p.ubox[3 + q] = c.ubox[3 + q]; }
            p.fb = d.d_fb; p.part = part;
            p.sky = c.opt.sky; p.rr = c.opt.rr; p.rng_mode = c.opt.rng; p.t_min = c.opt.t_min;
            p.counters = c.opt.counters ? d.d_counters : nullptr;
            p.queue = d.d_queue;
            p.order = d.d_order;
            // work items: one per pixel in the reference-stream mode (a pixel's samples are one sequential RNG stream);
            // with the per-sample counter stream the samples are independent and a pixel is split into chunks
            p.spw = spw; p.chunks = chunks; p.partial = nullptr;
            p.phase = 0; p.s_split = 0; p.px_state = d.d_px_state; p.px_rays = d.d_px_rays; p.ord_state = d.d_ord_state; p.ord_rays = d.d_ord_rays;
            static const int top_thr_env = getenv("RT_TOP_THR") ? atoi(getenv("RT_TOP_THR")) : 0;      // experiments
            p.chain_top_thr = top_thr_env >= 320 ? top_thr_env : 384;                                 // 24 rays per sample
            // Traffic forms of the two-dispatch frame (RtSphereParams::ord_rec / xcd_queues / p1_tile_major; A/B switches read per frame, defaults = what measured best)
            const bool ord_packed_env = env_flag("RT_ORD_PACKED", kDefaultOrdPacked);
            const bool xcd_queues_env = env_flag("RT_XCD_QUEUES", kDefaultXcdQueues);
            const int p1_tile_env = getenv("RT_P1_TILE") ? atoi(getenv("RT_P1_TILE")) : kDefaultP1Tile;     // 0 scattered, 1 tile-major, 2 scattered row segments
            p.ord_rec = ord_packed_env ? d.d_ord_rec : nullptr;
            p.xcd_queues = xcd_queues_env ? kXcdQueues : 0;
            p.p1_tile_major = (p1_tile_env >= 0 && p1_tile_env <= 2) ? p1_tile_env : 0;
            if (chunks > 1) {
                const size_t need = d.fb_rows * c.nx * (size_t)p.chunks * sizeof(rt_vec3);
                if (need > d.partial_bytes) {
                    if (d.d_partial) HIP_CHECK(hipFree(d.d_partial));
                    HIP_CHECK(hipMalloc((void**)&d.d_partial, need));
                    d.partial_bytes = need;
                }
                p.partial = d.d_partial;
            }
            static const char* dbg_path = getenv("RT_WAVE_DEBUG");      // diagnostics: per-wave time stamps -> file
            const size_t dbg_bytes = (size_t)65536 * 8 * sizeof(unsigned long long);
            if (dbg_path) {
                if (!d.d_wave_dbg) HIP_CHECK(hipMalloc((void**)&d.d_wave_dbg, dbg_bytes));
                HIP_CHECK(hipMemsetAsync(d.d_wave_dbg, 0, dbg_bytes, d.stream));
                p.wave_dbg = d.d_wave_dbg;
            }
            if (c.opt.nee) rt_fail("runRenderer: next-event estimation is only defined for mesh scenes");
            if (c.opt.floor) rt_fail("runRenderer: the floor plane is only defined for mesh scenes (kernel_scene.floor)");
            if (fb_direct) {
                void* dp = nullptr;
                HIP_CHECK(hipHostGetDevicePointer(&dp, (void*)h_target, 0));
                p.fb = reinterpret_cast<rt_vec3*>(dp);
                p.fb_global_rows = 1;
                static const bool poison_env = !(getenv("RT_FB_POISON") && getenv("RT_FB_POISON")[0] == '0');      // (experiment switch)
                p.poison_fb = poison_env ? 1 : 0;
            }
            // the device copy of the parameter block (RtSphereParams::self): owned by this DeviceState, refreshed by every frame from a pinned
            // staging copy (runRenderer is synchronous: the previous frame's upload has completed)
            p.self = d.d_params;
            *d.h_params = p;
            HIP_CHECK(hipMemcpyAsync(d.d_params, d.h_params, sizeof(RtSphereParams), hipMemcpyHostToDevice, d.stream));
            HIP_CHECK(c.opt.fp == RT_FP_FAST ? rt_launch_spheres_fast(p, c.opt.variant, d.stream)
                                             : rt_launch_spheres_parity(p, c.opt.variant, d.stream));
            launches++;
        } else {
            RtMeshParams p;
            memset(&p, 0, sizeof p);
            p.cam = c.cam; p.nx = c.nx; p.ny = c.ny; p.ns = ns; p.max_depth = c.max_depth;
            p.tris = d.d_tris; p.bvh4 = d.d_bvh; p.bvh_axis = d.d_bvh_axis;
            static const bool compact_leaves = !(getenv("RT_COMPACT_LEAVES") && getenv("RT_COMPACT_LEAVES")[0] == '0');        // A/B
            p.leaf_tri = compact_leaves ? d.d_leaf_tri : nullptr; p.leaf_ofs = compact_leaves ? d.d_leaf_ofs : nullptr;
            p.first_leaf = (uint32_t)c.num_bvh_nodes / 2; p.nppl = (uint32_t)c.nppl; p.bounds = c.bounds;
            p.leaf_sentinels_trailing = c.leaf_sentinels_trailing;
            p.lean_ok = 1;
            for (const rt_material& m : c.h_materials)
                if ((m.type != RT_DIFFUSE && m.type != RT_METAL && m.type != RT_GLASS) || m.texId != -1) p.lean_ok = 0;
            p.materials = d.d_materials;
            p.tex_data = d.d_tex_data; p.tex_width = d.d_tex_width; p.tex_height = d.d_tex_height;
            p.fb = d.d_fb; p.part = part;
            p.sky = c.opt.sky; p.nee = c.opt.nee; p.rr = c.opt.rr; p.rng_mode = c.opt.rng; p.t_min = c.opt.t_min;
            p.light = c.opt.light; p.lightColor = c.opt.lightColor;
            p.floor_on = c.opt.floor ? 1 : 0; p.floor = c.floor;
            if (c.opt.floor && (c.opt.variant & 0xFF) == 1) rt_fail("runRenderer: the floor plane is not built into the tile-per-wave A/B kernel (variant 1)");
            p.counters = c.opt.counters ? d.d_counters : nullptr;
            p.queue = d.d_queue;
            p.s_split = 0; p.px_state = d.d_px_state; p.px_rays = d.d_px_rays; p.order = d.d_order; p.ord_state = d.d_ord_state; p.ord_rays = d.d_ord_rays;
            // the traffic forms of the two-dispatch frame, as for sphere scenes (the same switches; the mesh frame always renders into the device framebuffer)
            p.ord_rec = env_flag("RT_ORD_PACKED", kDefaultOrdPacked) ? d.d_ord_rec : nullptr;
            p.xcd_queues = env_flag("RT_XCD_QUEUES", kDefaultXcdQueues) ? kXcdQueues : 0;
            p.p1_segments = (getenv("RT_P1_TILE") ? atoi(getenv("RT_P1_TILE")) : kDefaultP1Tile) == 2 ? 1 : 0;
            if (getenv("RT_WAVE_DEBUG")) {                           // diagnostics: phase cycle / lane counters -> file
                const size_t dbg_bytes = (size_t)65536 * 8 * sizeof(unsigned long long);
                if (!d.d_wave_dbg) HIP_CHECK(hipMalloc((void**)&d.d_wave_dbg, dbg_bytes));
                HIP_CHECK(hipMemsetAsync(d.d_wave_dbg, 0, dbg_bytes, d.stream));
                p.dbg = d.d_wave_dbg;
            }
            HIP_CHECK(c.opt.fp == RT_FP_FAST ? rt_launch_mesh_fast(p, c.opt.variant, d.stream)
                                             : rt_launch_mesh_parity(p, c.opt.variant, d.stream));
            launches++;
        }
        HIP_CHECK(hipEventRecord(d.ev_stop, d.stream));

        // gather: local stripe q (rows of the compact buffer) -> global stripe q*world + rank of the pinned framebuffer
        const int sr = c.opt.stripe_rows;
        const size_t stripe_bytes = (size_t)sr * row_bytes;
        const size_t full = d.fb_rows / sr, rem = d.fb_rows % sr;
        char* dst0 = reinterpret_cast<char*>(c.h_ext ? c.h_ext : c.h_fb) + (size_t)part.rank * stripe_bytes;
        const char* src0 = reinterpret_cast<const char*>(d.d_fb);
        if (world == 1) {                                               // the whole image: one linear copy
            if (!fb_direct) HIP_CHECK(hipMemcpyAsync(dst0, src0, d.fb_rows * row_bytes, hipMemcpyDeviceToHost, d.stream));
        } else if (full > 0 && !fb_direct)
            HIP_CHECK(hipMemcpy2DAsync(dst0, (size_t)world * stripe_bytes, src0, stripe_bytes, stripe_bytes, full,
                                       hipMemcpyDeviceToHost, d.stream));
        if (world != 1 && rem > 0 && !fb_direct)
            HIP_CHECK(hipMemcpyAsync(dst0 + full * (size_t)world * stripe_bytes, src0 + full * stripe_bytes, rem * row_bytes,
                                     hipMemcpyDeviceToHost, d.stream));
        samples += (int64_t)d.fb_rows * c.nx * ns;
    }

    double kernel_ms = 0.0;
    rt_render_stats st;
    memset(&st, 0, sizeof st);
    for (int k = 0; k < nd; k++) {
        DeviceState& d = c.devs[k];
        if (d.fb_rows == 0) continue;
        HIP_CHECK(hipSetDevice(d.device));
        HIP_CHECK(hipStreamSynchronize(d.stream));                      // kernels.cu:660-661: blocking
        float ms = 0.0f;
        HIP_CHECK(hipEventElapsedTime(&ms, d.ev_start, d.ev_stop));
        kernel_ms = std::max(kernel_ms, (double)ms);
        if (d.d_wave_dbg && getenv("RT_WAVE_DEBUG")) {
            std::vector<unsigned long long> h((size_t)65536 * 8);
            HIP_CHECK(hipMemcpy(h.data(), d.d_wave_dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            if (FILE* f = fopen(getenv("RT_WAVE_DEBUG"), "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), f); fclose(f); }
            if (d.d_px_state && c.is_spheres) {                      // per-pixel time line of the second phase (see finish())
                std::vector<float> px((size_t)d.fb_rows * c.nx * 4);
                HIP_CHECK(hipMemcpy(px.data(), d.d_px_state, px.size() * sizeof(float), hipMemcpyDeviceToHost));
                const std::string path = std::string(getenv("RT_WAVE_DEBUG")) + ".px";
                if (FILE* f = fopen(path.c_str(), "wb")) { fwrite(px.data(), sizeof(float), px.size(), f); fclose(f); }
            }
        }
        if (c.opt.counters) {
            RtCounters h;
            HIP_CHECK(hipMemcpy(&h, d.d_counters, sizeof h, hipMemcpyDeviceToHost));
            st.rays += h.rays; st.prim_tests += h.prim_tests; st.node_visits += h.node_visits;
            st.exec_tests += h.exec_tests; st.shadow_rays += h.shadow_rays; st.box_tests += h.box_tests;
            for (int q = 0; q < RT_STAT_COUNT; q++) st.ref_stats[q] += h.ref_stats[q];
        }
    }
    HIP_CHECK(hipSetDevice(current));
    const auto t1 = std::chrono::steady_clock::now();
    st.kernel_ms = kernel_ms;
    st.total_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    st.samples = samples;
    st.num_launches = launches;
    c.stats = st;
}

void setExternalFramebuffer(rt_vec3* fb) {
    RenderContext& c = g_ctx;
    if (!c.initialised) rt_fail("setExternalFramebuffer before init");
    if (c.h_ext) { if (c.ext_registered) HIP_CHECK(hipHostUnregister(c.h_ext)); c.h_ext = nullptr; c.ext_registered = false; }
    if (fb) {
        // Page-locking the caller's memory (a /dev/shm mapping shared by the ranks of a node, bench.py) lets the kernels deliver finished pixels
        // straight into it.  If the runtime refuses (locked-memory limit of the account, a mapping it cannot pin) the job must not die on its first
        // multi-GPU node: the renderer falls back to its compact device buffer and plain device-to-host copies of this member's stripes into the
        // (pageable) memory - slower by the copy, same image.  RT_EXT_FB_NO_REGISTER=1 forces that path (tests).
        const char* no_reg = getenv("RT_EXT_FB_NO_REGISTER");
        hipError_t e = (no_reg && no_reg[0] == '1') ? hipErrorNotSupported
                                                    : hipHostRegister(fb, (size_t)c.nx * c.ny * sizeof(rt_vec3), hipHostRegisterDefault);
        if (e != hipSuccess) {
            (void)hipGetLastError();                                 // (clear the sticky error: it has been handled)
            fprintf(stderr, "rt warning: setExternalFramebuffer could not page-lock the caller's framebuffer (%s): stripes are copied into it "
                            "from the device buffer instead of being stored by the kernel\n", hipGetErrorString(e));
        }
        c.ext_registered = e == hipSuccess;
        c.h_ext = fb;
    }
}

void getRenderStats(rt_render_stats* out) {
    if (out) *out = g_ctx.stats;
}

void cleanupRenderer(void) {
    if (!g_ctx.initialised) return;
    cleanup_impl();
    // kernels.cu:679 ends with cudaDeviceReset().  A reset destroys EVERY context of the process on that device - also the one of a
    // host that shares the process (PyTorch in bench.py, a viewer) - so it is opt-in here: RT_CLEANUP_DEVICE_RESET=1 mirrors the reference.
    if (const char* r = getenv("RT_CLEANUP_DEVICE_RESET")) if (r[0] == '1') (void)hipDeviceReset();
}

}  // extern "C"
