/*
 * ref_driver.cpp — thin C-ABI shim over the REFERENCE'S OWN header-only hot-path code.
 * TEST INFRASTRUCTURE ONLY; built only in the container that has /root/reference.
 *
 * Nothing from the reference is copied here: its headers are #included from where they lie
 * (-I/root/reference) and its functions are called as they are.  <cuda_runtime.h>
 * (helper_structs.h:3) resolves to the NVIDIA header that ships in this image inside
 * triton (…/triton/backends/nvidia/include); under a plain host compiler it defines
 * __host__/__device__ to nothing, so no stand-in header is written.
 *
 * Built by oracle/Makefile with ROCm's clang++ (-O2 -ffp-contract=off) into
 * oracle/_ref/libref.so.  clang evaluates the arguments of `vec3(rnd(),rnd(),rnd())`
 * (rnd.h:23,41) left to right — the order hipcc/clang device code uses and the order the
 * oracle pins; g++ would draw z,y,x.
 *
 * Exports:
 *   ref_*            one wrapper per reference function on the hot path (same argument meaning
 *                    as the orc_* function of the same name in rt_oracle.h)
 *   ref_render_mesh      the MESH path's twin: render() / color() / hit() / hitMesh() / hitBvh() / generateShadowRay()
 *                    of kernels.cu:154-224,296-569 as a host loop whose control flow is transcribed from there (kernels.cu
 *                    itself cannot be compiled here: <<<>>>, tex1Dfetch, threadIdx) and whose EVERY arithmetic step is a
 *                    call into the reference's own headers: vec3 operators, ray, hit_bbox, hit_bbox_dist, triangleHit,
 *                    sphereHit, planeHit, material_scatter / the preset scatters, rnd, wang_hash, get_ray, cosf/sinf and
 *                    the double M_PI expressions as written.  Pins oracle/rt_oracle.c's mesh path (tests/test_oracle_vs_ref.py).
 *   ref_generate_shadow_ray   the same generateShadowRay twin on tabulated inputs (pins orc_generate_shadow_ray and,
 *                    through it, the device probe rtProbeShadowRay)
 *   ref_render_spheres   a single-threaded host loop over pixels and samples for SPHERE scenes
 *                    whose every arithmetic step is a call into the reference headers
 *                    (get_ray, sphereHit, material_scatter, rnd, wang_hash, vec3 operators).
 *                    The reference at HEAD has no sphere scene, so the
 *                    loop's control flow is ours; it follows render()/color()/hit() of
 *                    kernels.cu:325-360,396-569 with the sphere list in place of the mesh.
 */
#include <cstdint>
#include <cfloat>
#include <cmath>
#include <cstring>

#include "rnd.h"
#include "camera.h"
#include "intersections.h"
#include "material.h"
#include "scene_materials.h"

#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"
#include <fstream>
#include <sstream>
#include "staircase_scene.h"

#include "../include/rt_types.h"
#include "rt_oracle.h"      /* orc_scatter / orc_counters struct shapes only */

static_assert(sizeof(vec3) == sizeof(rt_vec3), "vec3");
static_assert(sizeof(camera) == sizeof(rt_camera), "camera");
static_assert(sizeof(sphere) == sizeof(rt_sphere), "sphere");
static_assert(sizeof(plane) == sizeof(rt_plane), "plane");
static_assert(sizeof(bbox) == sizeof(rt_bbox), "bbox");
static_assert(sizeof(triangle) == sizeof(rt_triangle), "triangle");
static_assert(sizeof(bvh_node) == sizeof(rt_bvh_node), "bvh_node");
static_assert(sizeof(material) == sizeof(rt_material), "material");
static_assert(sizeof(stexture) == sizeof(rt_stexture), "stexture");
static_assert(sizeof(mesh) == sizeof(rt_mesh), "mesh");
static_assert(sizeof(kernel_scene) == sizeof(rt_kernel_scene), "kernel_scene");
static_assert(offsetof(camera, lens_radius) == offsetof(rt_camera, lens_radius), "camera.lens_radius");
static_assert(offsetof(triangle, meshID) == offsetof(rt_triangle, meshID), "triangle.meshID");
static_assert(offsetof(material, texId) == offsetof(rt_material, texId), "material.texId");
static_assert(offsetof(mesh, bounds) == offsetof(rt_mesh, bounds), "mesh.bounds");
static_assert(offsetof(kernel_scene, numPrimitivesPerLeaf) == offsetof(rt_kernel_scene, numPrimitivesPerLeaf), "ks.nppl");

static inline vec3 V(const float* p) { return vec3(p[0], p[1], p[2]); }
static inline void S(float* p, const vec3& v) { p[0] = v.x(); p[1] = v.y(); p[2] = v.z(); }

extern "C" {

int ref_struct_sizes(int* out, int cap) {
    const int s[] = { (int)sizeof(vec3), (int)sizeof(ray), (int)sizeof(camera), (int)sizeof(sphere), (int)sizeof(plane),
                      (int)sizeof(bbox), (int)sizeof(triangle), (int)sizeof(bvh_node), (int)sizeof(material),
                      (int)sizeof(stexture), (int)sizeof(mesh), (int)sizeof(scene), (int)sizeof(kernel_scene),
                      (int)sizeof(intersection), (int)sizeof(scatter_info), (int)sizeof(path), (int)sizeof(tri_hit) };
    int n = (int)(sizeof(s) / sizeof(s[0]));
    for (int i = 0; i < n && i < cap; i++) out[i] = s[i];
    return n;
}

/* the same handshake as orc_abi_sizes (rt_oracle.h): the structs of OURS that the shim's entry points read or write through caller pointers */
int ref_abi_sizes(int32_t* out, int n) {
    const int32_t s[4] = { (int32_t)sizeof(orc_scene), (int32_t)sizeof(orc_counters), (int32_t)sizeof(orc_scatter), (int32_t)sizeof(rt_render_options) };
    for (int k = 0; k < n && k < 4; k++) out[k] = s[k];
    return 4;
}

uint32_t ref_wang_hash(uint32_t seed) { return wang_hash(seed); }
uint32_t ref_pixel_seed(uint32_t pixel_id) { return (wang_hash(pixel_id) * 336343633) | 1; }
uint32_t ref_xor_shift_32(uint32_t* state) { return xor_shift_32(*state); }
float    ref_rnd(uint32_t* state) { return rnd(*state); }
void     ref_random_in_unit_disk(uint32_t* state, float out[3]) { S(out, random_in_unit_disk(*state)); }
void     ref_random_in_unit_sphere(uint32_t* state, float out[3]) { S(out, random_in_unit_sphere(*state)); }

void ref_make_camera(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov,
                     float aspect, float aperture, float focus_dist, rt_camera* out) {
    /* route every input through volatile so the constructor runs at run time (SURVEY.md §8c finding 2) */
    volatile float in[13];
    for (int i = 0; i < 3; i++) { in[i] = lookfrom[i]; in[3 + i] = lookat[i]; in[6 + i] = vup[i]; }
    in[9] = vfov; in[10] = aspect; in[11] = aperture; in[12] = focus_dist;
    camera c(vec3(in[0], in[1], in[2]), vec3(in[3], in[4], in[5]), vec3(in[6], in[7], in[8]), in[9], in[10], in[11], in[12]);
    memcpy(out, &c, sizeof c);
}

void ref_get_ray(const rt_camera* c, float s, float t, uint32_t* state, float org[3], float dir[3]) {
    camera cam; memcpy(&cam, c, sizeof cam);
    ray r = get_ray(cam, s, t, *state);
    S(org, r.origin()); S(dir, r.direction());
}

float ref_sphere_hit(const rt_sphere* s, const float org[3], const float dir_in[3], float t_min, float t_max) {
    sphere sp; memcpy(&sp, s, sizeof sp);
    return sphereHit(sp, ray(V(org), V(dir_in)), t_min, t_max);
}
float ref_triangle_hit(const rt_triangle* tri, const float org[3], const float dir_in[3], float t_min, float t_max,
                       float* hitU, float* hitV) {
    triangle t; memcpy(&t, tri, sizeof t);
    return triangleHit(t, ray(V(org), V(dir_in)), t_min, t_max, *hitU, *hitV);
}
int ref_hit_bbox(const float bmin[3], const float bmax[3], const float org[3], const float dir_in[3], float t_max) {
    return hit_bbox(V(bmin), V(bmax), ray(V(org), V(dir_in)), t_max) ? 1 : 0;
}
float ref_hit_bbox_dist(const float bmin[3], const float bmax[3], const float org[3], const float dir_in[3], float t_max) {
    return hit_bbox_dist(V(bmin), V(bmax), ray(V(org), V(dir_in)), t_max);
}
float ref_plane_hit(const rt_plane* p, const float org[3], const float dir_in[3], float t_min, float t_max) {
    plane pl; memcpy(&pl, p, sizeof pl);
    return planeHit(pl, ray(V(org), V(dir_in)), t_min, t_max);
}

float ref_schlick(float cosine, float ref_idx) { return schlick(cosine, ref_idx); }
void  ref_reflect(const float v[3], const float n[3], float out[3]) { S(out, reflect(V(v), V(n))); }
void  ref_refract(const float uv[3], const float n[3], float e, float out[3]) { S(out, refract(V(uv), V(n), e)); }

void ref_material_scatter(float inters_t, const float normal[3], int inside, const float wo[3],
                          const rt_material* mat, const float color[3], uint32_t* rng, orc_scatter* out) {
    intersection in; memset((void*)&in, 0, sizeof in);
    in.t = inters_t; in.normal = V(normal); in.inside = inside != 0;
    material m; memcpy(&m, mat, sizeof m);
    scatter_info sc(in);
    material_scatter(sc, in, V(wo), m, V(color), *rng);
    S(out->wi, sc.wi); out->specular = sc.specular; S(out->throughput, sc.throughput);
    out->refracted = sc.refracted; out->t = sc.t;
}

/* The dormant look presets, called exactly as scene_materials.h:22-93 defines them. kind = RT_FLOOR_COAT .. RT_MODEL_SSS */
void ref_preset_scatter(int kind, float inters_t, const float p[3], const float normal[3], int inside, const float wo[3],
                        uint32_t* rng, orc_scatter* out) {
    intersection in; memset((void*)&in, 0, sizeof in);
    in.t = inters_t; in.p = V(p); in.normal = V(normal); in.inside = inside != 0;
    scatter_info sc(in);
    switch (kind) {
    case RT_FLOOR_COAT:        floor_coat_scatter(sc, in, V(wo), *rng); break;
    case RT_FLOOR_DIFFUSE:     floor_diffuse_scatter(sc, in, V(wo), *rng); break;
    case RT_FLOOR_CHECKER:     floor_checker_scatter(sc, in, V(wo), *rng); break;
    case RT_MODEL_COAT:        model_coat_scatter(sc, in, V(wo), *rng); break;
    case RT_MODEL_DIFFUSE:     model_diffuse_scatter(sc, in, V(wo), *rng); break;
    case RT_MODEL_GLOSSY:      model_glossy_scatter(sc, in, V(wo), *rng); break;
    case RT_MODEL_GLASS:       model_glass_scatter(sc, in, V(wo), *rng); break;
    case RT_MODEL_TINTEDGLASS: model_tintedglass_scatter(sc, in, V(wo), *rng); break;
    default:                   model_sss_scatter(sc, in, V(wo), *rng); break;
    }
    S(out->wi, sc.wi); out->specular = sc.specular; S(out->throughput, sc.throughput);
    out->refracted = sc.refracted; out->t = sc.t;
}

uint32_t ref_linear_to_srgb(float x) { return LinearToSRGB(x); }

/* Expression probes: the two vec3 expressions of generateShadowRay whose operator order matters
 * (kernels.cu:379 and :387), evaluated with the reference's own vec3 operators. */
void ref_probe_light_dir(const float su[3], const float sv[3], const float sw[3], float phi, float sinA, float cosA, float out[3]) {
    const vec3 l = V(su) * cosf(phi) * sinA + V(sv) * sinf(phi) * sinA + V(sw) * cosA;
    S(out, l);
}
void ref_probe_light_contribution(const float att[3], const float lightColor[3], float dotl, float cosAMax, float out[3]) {
    const float omega = 2 * M_PI * (1.0f - cosAMax);
    const vec3 c = V(att) * V(lightColor) * dotl * omega / M_PI;
    S(out, c);
}
float ref_probe_phi(float eps2) { const float phi = 2 * M_PI * eps2; return phi; }

/* Host loop for sphere scenes built from the reference's functions (see file header). */
void ref_render_spheres(const rt_sphere* spheres_, const rt_material* mats_, int n, const rt_camera* cam_,
                        int sky_gradient, int rr, float t_min, int counter_rng,
                        int nx, int ny, int ns, int max_depth,
                        int x0, int y0, int x1, int y1, rt_vec3* fb_, orc_counters* cnt) {
    const sphere* spheres = reinterpret_cast<const sphere*>(spheres_);
    const material* mats = reinterpret_cast<const material*>(mats_);
    camera cam; memcpy(&cam, cam_, sizeof cam);
    vec3* fb = reinterpret_cast<vec3*>(fb_);
    if (max_depth > 255) max_depth = 255;

    for (int j = y0; j < y1; j++)
        for (int i = x0; i < x1; i++) {
            path p;
            uint64_t pixelId = j * nx + i;
            p.rng = (wang_hash(pixelId) * 336343633) | 1;
            vec3 col(0, 0, 0);
            for (int s = 0; s < ns; s++) {
                if (counter_rng) p.rng = (wang_hash((uint32_t)pixelId + wang_hash((uint32_t)s) * 0x9E3779B9u) * 336343633) | 1;
                float u = float(i + rnd(p.rng)) / float(nx);
                float v = float(j + rnd(p.rng)) / float(ny);
                ray r0 = get_ray(cam, u, v, p.rng);
                p.origin = r0.origin();
                p.rayDir = r0.direction();
                p.specular = false;
                p.inside = false;
                p.attenuation = vec3(1.0, 1.0, 1.0);
                p.color = vec3(0, 0, 0);
                for (p.bounce = 0; p.bounce < max_depth; p.bounce++) {
                    const ray r(p.origin, p.rayDir);
                    if (cnt) cnt->rays++;
                    float closest = FLT_MAX;
                    int sid = -1;
                    for (int k = 0; k < n; k++) {
                        float t = sphereHit(spheres[k], r, t_min, closest);
                        if (t < closest) { closest = t; sid = k; }
                    }
                    if (cnt) cnt->prim_tests += n;
                    if (sid < 0) {
                        if (sky_gradient) {
                            float t = 0.5f * (p.rayDir.y() + 1.0f);
                            vec3 c = (1.0f - t) * vec3(1.0, 1.0, 1.0) + t * vec3(0.5, 0.7, 1.0);
                            p.color += p.attenuation * c;
                        } else {
                            p.color += p.attenuation * vec3(0.5f, 0.5f, 0.5f);
                        }
                        break;
                    }
                    if (cnt) cnt->hits++;
                    intersection inters;
                    inters.t = closest;
                    inters.p = r.point_at_parameter(inters.t);
                    inters.normal = (inters.p - spheres[sid].center) / spheres[sid].radius;
                    if (dot(r.direction(), inters.normal) > 0.0f)
                        inters.normal = -inters.normal;
                    inters.inside = p.inside;
                    scatter_info scatter(inters);
                    material_scatter(scatter, inters, p.rayDir, mats[sid], mats[sid].color, p.rng);
                    p.origin += scatter.t * p.rayDir;
                    p.rayDir = scatter.wi;
                    p.attenuation *= scatter.throughput;
                    p.specular = scatter.specular;
                    p.inside = scatter.refracted ? !p.inside : p.inside;
                    if (rr && p.bounce > 3) {
                        float m = max(p.attenuation);
                        if (rnd(p.rng) > m) break;
                        p.attenuation *= 1 / m;
                    }
                }
                col += p.color;
                if (cnt) cnt->samples++;
            }
            fb[pixelId] = col / float(ns);
        }
}


} /* extern "C" */

/* ------------------------------------------------------------------------------------------------------------
 * Mesh path twin.  What stands in for kernels.cu's RenderContext (kernels.cu:69-143); the BVH is the plain
 * bvh_node array of the non-texture branch (kernels.cu:174-176: same floats as the 3-texel fetch of :166-173).
 * ------------------------------------------------------------------------------------------------------------ */
namespace {

struct MeshCtx {
    const triangle* tris;
    const bvh_node* bvh;
    uint32_t firstLeafIdx;              /* kernels.cu:614 */
    uint32_t numPrimitivesPerLeaf;
    bbox bounds;
    plane floor;
    int maxDepth;
    sphere light;
    vec3 lightColor;
    const material* materials;
    const rt_stexture* textures;
    /* the reference's compile-time switches, as run-time values */
    bool shadow, russian_roulette, gradient_sky, use_floor;
    float epsilon;                      /* kernels.cu:19 */
    orc_counters* cnt;
    void stat(int k) const { if (cnt) cnt->ref_stats[k]++; }
};

enum { T_NONE, T_TRIMESH, T_PLANE, T_LIGHT };   /* kernels.cu:40-45 */

void twin_pop(unsigned int& bitStack, int& idx) {                 /* kernels.cu:148-152 */
    const int m = __builtin_ffsll(bitStack) - 1;
    bitStack = (bitStack >> m) ^ 1;
    idx = (idx >> m) ^ 1;
}

float twin_hit_bvh(const ray& r, const MeshCtx& cx, float t_min, float t_max, tri_hit& rec, bool isShadow) {   /* kernels.cu:154-224 */
    int idx = 1;
    float closest = t_max;
    unsigned int bitStack = 1;
    while (idx) {
        if (idx < (int)cx.firstLeafIdx) {
            const int idx2 = idx << 1;
            bvh_node left = cx.bvh[idx2];
            bvh_node right = cx.bvh[idx2 + 1];
            if (cx.cnt) cx.cnt->node_visits++;
            const float leftHit = hit_bbox_dist(left.min(), left.max(), r, closest);
            const bool traverseLeft = leftHit < closest;
            const float rightHit = hit_bbox_dist(right.min(), right.max(), r, closest);
            const bool traverseRight = rightHit < closest;
            const bool swap = rightHit < leftHit;
            if (traverseLeft && traverseRight) {
                cx.stat(RT_STAT_NODES_BOTH);
                idx = idx2 + (swap ? 1 : 0);
                bitStack = (bitStack << 1) + 1;
            } else if (traverseLeft || traverseRight) {
                cx.stat(RT_STAT_NODES_SINGLE);
                idx = idx2 + (swap ? 1 : 0);
                bitStack = bitStack << 1;
            } else {
                twin_pop(bitStack, idx);
            }
        } else {
            const int first = (idx - cx.firstLeafIdx) * cx.numPrimitivesPerLeaf;
            for (unsigned i = 0; i < cx.numPrimitivesPerLeaf; i++) {
                const triangle tri = cx.tris[first + i];
                if (isinf(tri.v[0].x())) break;
                float u, v;
                if (cx.cnt) cx.cnt->prim_tests++;
                const float hitT = triangleHit(tri, r, t_min, closest, u, v);
                if (hitT < closest) {
                    if (isShadow) return 0.0f;
                    closest = hitT;
                    rec.triId = first + i;
                    rec.u = u;
                    rec.v = v;
                }
            }
            twin_pop(bitStack, idx);
        }
    }
    return closest;
}

float twin_hit_mesh(const ray& r, const MeshCtx& cx, float t_min, float t_max, tri_hit& rec, bool primary, bool isShadow) {   /* kernels.cu:296-323 */
    if (!hit_bbox(cx.bounds.min, cx.bounds.max, r, t_max)) {
        if (isShadow) cx.stat(RT_STAT_SHADOWS_BBOX_NOHITS);
        else cx.stat(primary ? RT_STAT_PRIMARY_BBOX_NOHITS : RT_STAT_SECONDARY_BBOX_NOHIT);
        return FLT_MAX;
    }
    return twin_hit_bvh(r, cx, t_min, t_max, rec, isShadow);
}

bool twin_hit(const MeshCtx& cx, const path& p, float t_max, bool isShadow, intersection& inters) {             /* kernels.cu:325-360 */
    const ray r = isShadow ? ray(p.origin, p.shadowDir) : ray(p.origin, p.rayDir);
    tri_hit triHit;
    const bool primary = p.bounce == 0;
    inters.objId = T_NONE;
    if ((inters.t = twin_hit_mesh(r, cx, cx.epsilon, t_max, triHit, primary, isShadow)) < t_max) {
        if (isShadow) return true;
        inters.objId = T_TRIMESH;
        triangle tri = cx.tris[triHit.triId];
        inters.meshID = tri.meshID;
        inters.normal = unit_vector(cross(tri.v[1] - tri.v[0], tri.v[2] - tri.v[0]));
        inters.texCoords[0] = (triHit.u * tri.texCoords[1 * 2 + 0] + triHit.v * tri.texCoords[2 * 2 + 0] + (1 - triHit.u - triHit.v) * tri.texCoords[0 * 2 + 0]);
        inters.texCoords[1] = (triHit.u * tri.texCoords[1 * 2 + 1] + triHit.v * tri.texCoords[2 * 2 + 1] + (1 - triHit.u - triHit.v) * tri.texCoords[0 * 2 + 1]);
    } else {
        if (isShadow) return false;
        if (cx.use_floor && (inters.t = planeHit(cx.floor, r, cx.epsilon, FLT_MAX)) < FLT_MAX) {   /* the call site of kernels.cu:341-345, re-enabled */
            inters.objId = T_PLANE;
            inters.normal = cx.floor.norm;
        } else
        if (p.specular && sphereHit(cx.light, r, cx.epsilon, t_max) < t_max) {
            inters.objId = T_LIGHT;
            return true;
        }
    }
    if (inters.objId != T_NONE) {
        inters.p = r.point_at_parameter(inters.t);
        if (dot(r.direction(), inters.normal) > 0.0f)
            inters.normal = -inters.normal;
        return true;
    }
    return false;
}

bool twin_generate_shadow_ray(const MeshCtx& cx, path& p, const intersection& inters, float& lightDist, float* cosAMax_out) {   /* kernels.cu:363-393 */
    const vec3 sw = unit_vector(cx.light.center - p.origin);
    const vec3 su = unit_vector(cross(fabs(sw.x()) > 0.01f ? vec3(0, 1, 0) : vec3(1, 0, 0), sw));
    const vec3 sv = cross(sw, su);
    const float cosAMax = sqrt(1.0f - cx.light.radius * cx.light.radius / (p.origin - cx.light.center).squared_length());
    if (cosAMax_out) *cosAMax_out = cosAMax;
    if (isnan(cosAMax)) return false;
    const float eps1 = rnd(p.rng);
    const float eps2 = rnd(p.rng);
    const float cosA = 1.0f - eps1 + eps1 * cosAMax;
    const float sinA = sqrt(1.0f - cosA * cosA);
    const float phi = 2 * M_PI * eps2;
    const vec3 l = su * cosf(phi) * sinA + sv * sinf(phi) * sinA + sw * cosA;
    const float dotl = dot(l, inters.normal);
    if (dotl <= 0)
        return false;
    p.shadowDir = unit_vector(l);
    const float omega = 2 * M_PI * (1.0f - cosAMax);
    p.lightContribution = p.attenuation * cx.lightColor * dotl * omega / M_PI;
    lightDist = (cx.light.center - p.origin).length() - cx.light.radius;
    return true;
}

void twin_scatter(scatter_info& sc, const intersection& in, const vec3& wo, const material& m, const vec3& albedo, rand_state& rng) {
    switch ((int)m.type) {                                           /* additive preset types call the reference's own preset functions */
    case RT_FLOOR_COAT:        floor_coat_scatter(sc, in, wo, rng); break;
    case RT_FLOOR_DIFFUSE:     floor_diffuse_scatter(sc, in, wo, rng); break;
    case RT_FLOOR_CHECKER:     floor_checker_scatter(sc, in, wo, rng); break;
    case RT_MODEL_COAT:        model_coat_scatter(sc, in, wo, rng); break;
    case RT_MODEL_DIFFUSE:     model_diffuse_scatter(sc, in, wo, rng); break;
    case RT_MODEL_GLOSSY:      model_glossy_scatter(sc, in, wo, rng); break;
    case RT_MODEL_GLASS:       model_glass_scatter(sc, in, wo, rng); break;
    case RT_MODEL_TINTEDGLASS: model_tintedglass_scatter(sc, in, wo, rng); break;
    case RT_MODEL_SSS:         model_sss_scatter(sc, in, wo, rng); break;
    default:                   material_scatter(sc, in, wo, m, albedo, rng); break;     /* kernels.cu:480 */
    }
}

void twin_color(const MeshCtx& cx, path& p) {                                                                    /* kernels.cu:396-533 */
    p.attenuation = vec3(1.0, 1.0, 1.0);
    p.color = vec3(0, 0, 0);
    bool fromMesh = false;
    for (p.bounce = 0; p.bounce < cx.maxDepth; p.bounce++) {
        const bool primary = p.bounce == 0;
        cx.stat(primary ? RT_STAT_PRIMARY : RT_STAT_SECONDARY);
        if (fromMesh) cx.stat(RT_STAT_SECONDARY_MESH);
        if (p.attenuation.length() < 0.01f) cx.stat(RT_STAT_LOW_POWER);
        if (cx.cnt) cx.cnt->rays++;
        intersection inters;
        if (!twin_hit(cx, p, FLT_MAX, false, inters)) {
            if (primary) cx.stat(RT_STAT_PRIMARY_NOHITS);
            else cx.stat(fromMesh ? RT_STAT_SECONDARY_MESH_NOHIT : RT_STAT_SECONDARY_NOHIT);
            if (cx.gradient_sky) {                                   /* kernels.cu:419-421 */
                float t = 0.5f * (p.rayDir.y() + 1.0f);
                vec3 c = (1.0f - t) * vec3(1.0, 1.0, 1.0) + t * vec3(0.5, 0.7, 1.0);
                p.color += p.attenuation * c;
            } else {
                p.color += p.attenuation * vec3(0.5f, 0.5f, 0.5f);   /* kernels.cu:424 */
            }
            return;
        }
        if (cx.cnt) cx.cnt->hits++;
        fromMesh = (inters.objId == T_TRIMESH);
        if (primary && !fromMesh) cx.stat(RT_STAT_PRIMARY_NOHITS);
        if (primary && fromMesh) cx.stat(RT_STAT_PRIMARY_HIT_MESH);
        if (inters.objId == T_LIGHT) {
            if (!cx.shadow) p.color += p.attenuation * cx.lightColor;   /* kernels.cu:440-446 */
            return;
        }
        inters.inside = p.inside;
        scatter_info scatter(inters);
        if (inters.objId == T_TRIMESH) {
            const material& mat = cx.materials[inters.meshID];
            vec3 albedo;
            if (mat.texId != -1) {                                   /* kernels.cu:456-476 */
                int texId = mat.texId;
                int width = cx.textures[texId].width;
                int height = cx.textures[texId].height;
                float tu = inters.texCoords[0];
                tu = tu - floorf(tu);
                float tv = inters.texCoords[1];
                tv = tv - floorf(tv);
                const int tx = (width - 1) * tu;
                const int ty = (height - 1) * tv;
                const int tIdx = ty * width + tx;
                albedo = vec3(cx.textures[texId].data[tIdx * 3 + 0], cx.textures[texId].data[tIdx * 3 + 1], cx.textures[texId].data[tIdx * 3 + 2]);
            } else {
                albedo = mat.color;
            }
            twin_scatter(scatter, inters, p.rayDir, cx.materials[inters.meshID], albedo, p.rng);
        } else
            floor_diffuse_scatter(scatter, inters, p.rayDir, p.rng);   /* kernels.cu:481-482 */

        p.origin += scatter.t * p.rayDir;
        p.rayDir = scatter.wi;
        p.attenuation *= scatter.throughput;
        p.specular = scatter.specular;
        p.inside = scatter.refracted ? !p.inside : p.inside;
        if (cx.shadow) {                                             /* kernels.cu:490-511 */
            float lightDist;
            if (!p.specular && twin_generate_shadow_ray(cx, p, inters, lightDist, nullptr)) {
                if (cx.cnt) cx.cnt->shadow_rays++;
                cx.stat(RT_STAT_SHADOWS);
                if (!twin_hit(cx, p, lightDist, true, inters)) {
                    cx.stat(RT_STAT_SHADOWS_NOHITS);
                    p.color += p.lightContribution;
                }
            }
        }
        if (cx.russian_roulette) {                                   /* kernels.cu:512-527 */
            if (p.bounce > 3) {
                float m = max(p.attenuation);
                if (rnd(p.rng) > m) { cx.stat(RT_STAT_RUSSIAN_KILL); return; }
                p.attenuation *= 1 / m;
            }
        }
    }
    cx.stat(RT_STAT_EXCEED_MAX_BOUNCE);
}

MeshCtx make_ctx(const rt_render_options* opt, int max_depth, orc_counters* cnt) {
    MeshCtx cx;
    memset((void*)&cx, 0, sizeof cx);
    cx.maxDepth = max_depth > 255 ? 255 : max_depth;
    memcpy((void*)&cx.light, &opt->light, sizeof cx.light);
    memcpy((void*)&cx.lightColor, &opt->lightColor, sizeof cx.lightColor);
    cx.shadow = opt->nee != 0; cx.russian_roulette = opt->rr != 0; cx.gradient_sky = opt->sky == RT_SKY_GRADIENT;
    cx.use_floor = opt->floor != 0;
    cx.epsilon = opt->t_min;
    cx.cnt = cnt;
    return cx;
}

}  // namespace

extern "C" {

int ref_generate_shadow_ray(const rt_render_options* opt, const float origin[3], const float attenuation[3], const float normal[3],
                            uint32_t* rng, float out9[9]) {
    MeshCtx cx = make_ctx(opt, 1, nullptr);
    path p;
    p.origin = V(origin); p.attenuation = V(attenuation); p.rng = *rng;
    p.shadowDir = vec3(0, 0, 0); p.lightContribution = vec3(0, 0, 0);
    intersection in; memset((void*)&in, 0, sizeof in);
    in.normal = V(normal);
    float lightDist = 0.0f, cosAMax = 0.0f;
    uint32_t probe = p.rng;
    const bool ok = twin_generate_shadow_ray(cx, p, in, lightDist, &cosAMax);
    int draws = 0;
    while (probe != p.rng && draws < 4) { xor_shift_32(probe); draws++; }
    S(out9, p.shadowDir); S(out9 + 3, p.lightContribution);
    out9[6] = lightDist; out9[7] = cosAMax; out9[8] = (float)draws;
    *rng = p.rng;
    return ok ? 1 : 0;
}

void ref_render_mesh(const rt_triangle* tris_, const rt_bvh_node* bvh_, int num_bvh_nodes, const float bounds6[6], int nppl,
                     const rt_plane* floor_, const rt_material* mats_, const rt_stexture* textures_,
                     const rt_camera* cam_, const rt_render_options* opt,
                     int nx, int ny, int ns, int max_depth,
                     int x0, int y0, int x1, int y1, rt_vec3* fb_, orc_counters* cnt) {                   /* render(), kernels.cu:535-569 */
    MeshCtx cx = make_ctx(opt, max_depth, cnt);
    cx.tris = reinterpret_cast<const triangle*>(tris_);
    cx.bvh = reinterpret_cast<const bvh_node*>(bvh_);
    cx.firstLeafIdx = num_bvh_nodes / 2;
    cx.numPrimitivesPerLeaf = nppl;
    cx.bounds = bbox(V(bounds6), V(bounds6 + 3));
    if (floor_) memcpy((void*)&cx.floor, floor_, sizeof cx.floor);
    cx.materials = reinterpret_cast<const material*>(mats_);
    cx.textures = textures_;
    camera cam; memcpy((void*)&cam, cam_, sizeof cam);
    vec3* fb = reinterpret_cast<vec3*>(fb_);
    for (int j = y0; j < y1; j++)
        for (int i = x0; i < x1; i++) {
            path p;
            uint64_t pixelId = j * nx + i;
            p.rng = (wang_hash(pixelId) * 336343633) | 1;
            vec3 col(0, 0, 0);
            for (int s = 0; s < ns; s++) {
                if (opt->rng == RT_RNG_COUNTER) p.rng = (wang_hash((uint32_t)pixelId + wang_hash((uint32_t)s) * 0x9E3779B9u) * 336343633) | 1;
                float u = float(i + rnd(p.rng)) / float(nx);
                float v = float(j + rnd(p.rng)) / float(ny);
                ray r = get_ray(cam, u, v, p.rng);
                p.origin = r.origin();
                p.rayDir = r.direction();
                p.specular = false;
                p.inside = false;
                twin_color(cx, p);
                col += p.color;
                if (cnt) cnt->samples++;
                if (isnan(p.color)) cx.stat(RT_STAT_NAN);
            }
            fb[pixelId] = col / float(ns);
        }
}

} /* extern "C" */

/* ---- the host I/O either side of the path, the reference's OWN code ----------------------------------------------
 * staircase_scene.h is #included above, so loadBVH (:75-101), writePPM (:32-43) and setup_camera (:62-73) are the reference's functions as they
 * are; these wrappers only move data across the C boundary.  They pin cuda-raytracing-optimized_amd/host/ (rtLoadBvhFile / rtSaveBvhFile,
 * rtWritePPM, rtStaircaseCamera): tests/test_oracle_vs_ref.py, and the fixtures oracle/gen_golden_hostio.py mints from them. */
#include <iostream>

extern "C" {

/* loadBVH on `path`.  Returns an opaque handle (NULL when loadBVH refuses the file); counts3 = numTris, numBvhNodes, numPrimitivesPerLeaf. */
void* ref_load_bvh(const char* path, int counts3[3]) {
    mesh* m = new mesh();
    int nppl = 0;
    {   /* loadBVH reports a bad header on std::cerr: keep the test log quiet */
        std::stringstream sink;
        std::streambuf* old = std::cerr.rdbuf(sink.rdbuf());
        const bool ok = std::ifstream(path).good() && loadBVH(path, *m, nppl);
        std::cerr.rdbuf(old);
        if (!ok) { delete m; return nullptr; }
    }
    counts3[0] = m->numTris; counts3[1] = m->numBvhNodes; counts3[2] = nppl;
    return m;
}
/* copies what loadBVH filled in: triangle[numTris], bvh_node[numBvhNodes], bounds (min, max) */
void ref_bvh_copy(const void* handle, rt_triangle* tris, rt_bvh_node* bvh, float bounds6[6]) {
    const mesh* m = static_cast<const mesh*>(handle);
    memcpy(tris, m->tris, sizeof(triangle) * (size_t)m->numTris);
    memcpy(bvh, m->bvh, sizeof(bvh_node) * (size_t)m->numBvhNodes);
    memcpy(bounds6, &m->bounds, sizeof(bbox));
}
void ref_bvh_free(void* handle) { delete static_cast<mesh*>(handle); }      /* mesh::~mesh frees what loadBVH allocated */

/* writePPM(nx, ny, colors) with std::cout captured.  Returns the number of bytes it wrote; up to `cap` of them land in `out`. */
long ref_write_ppm(int nx, int ny, const float* colors, char* out, long cap) {
    std::stringstream buf;
    std::streambuf* old = std::cout.rdbuf(buf.rdbuf());
    writePPM(nx, ny, reinterpret_cast<const vec3*>(colors));
    std::cout.rdbuf(old);
    const std::string s = buf.str();
    if (out && cap > 0) memcpy(out, s.data(), (size_t)((long)s.size() < cap ? (long)s.size() : cap));
    return (long)s.size();
}

/* setup_camera(nx, ny), the sizes routed through volatile so that the constructor runs at run time (as in ref_make_camera) */
void ref_setup_camera(int nx, int ny, rt_camera* out) {
    volatile int vx = nx, vy = ny;
    const camera c = setup_camera(vx, vy);
    memcpy(out, &c, sizeof c);
}

}  // extern "C"
