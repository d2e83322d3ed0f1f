/*
 * rt_probe.h — additive C-ABI of librt_mi355x.so: per-function device probes.
 *
 * Each entry runs ONE device function of the render hot path on arrays of n inputs (host
 * pointers in, host pointers out; the library does the transfers) so that parity tests can hold
 * every function on the path against the CPU oracle by itself.  `_parity` /
 * `_fast` pick the floating-point build (no FMA contraction / FMA contraction), exactly as
 * rt_render_options.fp does for the render kernels.  Vectors are packed xyz triples.
 * Same error convention as rt_api.h (print + exit(99) on a HIP failure).
 *
 * Reference functions probed (paths relative to /root/reference):
 *   rtProbeRng          wang_hash + seed + rnd                rnd.h:5-18,31-39; kernels.cu:541-542
 *   rtProbeDiskSphere   random_in_unit_disk / _sphere         rnd.h:20-26,41-49
 *   rtProbeGetRay       get_ray (+ ray ctor normalisation)    camera.h:8-12; ray.h:9
 *   rtProbeSphereHit    sphereHit                             intersections.h:85-104
 *   rtProbeTriangleHit  triangleHit                           intersections.h:54-83
 *   rtProbeBbox         hit_bbox_dist, hit_bbox               intersections.h:7-41
 *   rtProbeScatter      material_scatter (3 BSDFs + presets)  scene_materials.h:13-93; material.h:9-143 (p3 = hit point)
 *   rtProbeMath         a/b, sqrt|a|, pow(a,5), unit_vector   vec3.h:79-81,35,194-196; material.h:12
 *   rtProbeShadowRay    generateShadowRay as a whole          kernels.cu:363-393 (out9 = shadowDir, lightContribution, lightDist,
 *                                                             cosAMax, number of draws consumed; generated = its bool result)
 *   rtProbePlaneHit     planeHit                              intersections.h:43-52
 *   rtProbeSinCos       sinf / cosf of generateShadowRay      kernels.cu:378-379 (glibc's algorithm restated: csrc/rt_glibc_sincosf.h)
 *   rtProbeSchlick      schlick (with its powf(x, 5))         material.h:9-13 (glibc's powf restated: csrc/rt_glibc_powf.h)
 */
#ifndef RT_PROBE_H
#define RT_PROBE_H

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RT_PROBE_DECL(sfx) \
void rtProbeRng##sfx(const uint32_t* pixel_ids, int n, uint32_t* seed_out, float* draws4_out, uint32_t* state_out); \
void rtProbeDiskSphere##sfx(const uint32_t* states, int n, float* disk3, uint32_t* st_after_disk, float* sphere3, uint32_t* st_after_sphere); \
void rtProbeGetRay##sfx(const rt_camera* cam, const float* s, const float* t, const uint32_t* states, int n, \
                        float* org3, float* dir3, uint32_t* st_after); \
void rtProbeSphereHit##sfx(const rt_sphere* spheres, const float* org3, const float* dir3, const float* tmin, const float* tmax, \
                           int n, float* t_out); \
void rtProbeTriangleHit##sfx(const rt_triangle* tris, const float* org3, const float* dir3, const float* tmin, const float* tmax, \
                             int n, float* t_out, float* u_out, float* v_out); \
void rtProbeBbox##sfx(const float* bmin3, const float* bmax3, const float* org3, const float* dir3, const float* tmax, \
                      int n, float* dist_out, int* hit_out); \
/* flags: bit0 = specular, bit1 = refracted */ \
void rtProbeScatter##sfx(const float* t, const float* p3, const float* normal3, const int* inside, const float* wo3, const rt_material* mats, \
                         const float* color3, const uint32_t* states, int n, \
                         float* wi3, float* throughput3, int* flags, float* t_out, uint32_t* st_after); \
void rtProbeMath##sfx(const float* a, const float* b, int n, float* quot, float* root, float* p5, float* unit3); \
void rtProbeShadowRay##sfx(const rt_sphere* light, const rt_vec3* lightColor, const float* org3, const float* atten3, const float* normal3, \
                           const uint32_t* states, int n, float* out9, int* generated, uint32_t* st_after); \
void rtProbePlaneHit##sfx(const rt_plane* planes, const float* org3, const float* dir3, const float* tmin, const float* tmax, int n, float* t_out); \
void rtProbeSinCos##sfx(const float* y, int n, float* sin_out, float* cos_out); \
/* out = schlick(cosine, ref_idx); above = (u < schlick(cosine, ref_idx)) as the path decides it (rt_device.h schlick_above) */ \
void rtProbeSchlick##sfx(const float* cosine, const float* ref_idx, const float* u, int n, float* out, int* above);

RT_PROBE_DECL(_parity)
RT_PROBE_DECL(_fast)

#ifdef __cplusplus
}
#endif
#endif
