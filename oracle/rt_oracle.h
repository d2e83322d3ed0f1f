/*
 * rt_oracle.h — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A single-threaded plain-C restatement of the reference's render() hot path
 * (/root/reference/kernels.cu:148-224,296-569 and the header-only math it calls).  It exists to
 * CHECK the HIP path; it is never the thing shipped or measured: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against the reference's
 * own headers compiled in the build container (oracle/ref_driver.cpp -> oracle/_ref/libref.so,
 * tests/test_oracle_vs_ref.py) and against golden vectors minted from them (tests/golden/).
 *
 * Determinism rules: RNG draws are explicit statements in x,y,z order;
 * compile with -ffp-contract=off and never -ffast-math; camera and scene are inputs.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene {
    /* sphere scene (num_spheres > 0) */
    const rt_sphere*   spheres;
    const rt_material* sphere_materials;   /* one per sphere */
    int32_t            num_spheres;
    /* mesh scene (num_bvh_nodes > 0): layout contract of kernels.cu:154-224,572-648 */
    const rt_triangle* tris;
    int32_t            num_tris;
    const rt_bvh_node* bvh;                /* heap-indexed, node k at bvh[k], root = 1 */
    int32_t            num_bvh_nodes;
    rt_bbox            bounds;
    int32_t            nppl;               /* numPrimitivesPerLeaf */
    const rt_material* materials;          /* indexed by triangle.meshID */
    int32_t            num_materials;
    const rt_stexture* textures;
    int32_t            num_textures;
    rt_plane           floor;              /* kernel_scene.floor; only read when rt_render_options.floor = 1 */
} orc_scene;

typedef struct orc_counters {
    uint64_t samples;
    uint64_t rays;          /* closest-hit queries (color() loop iterations)  */
    uint64_t shadow_rays;
    uint64_t prim_tests;    /* sphereHit / triangleHit calls                  */
    uint64_t node_visits;   /* BVH internal nodes visited                     */
    uint64_t hits;
    uint64_t rng_draws;
    uint64_t ref_stats[RT_STAT_COUNT];  /* the reference's `#ifdef STATS` counters, kernels.cu:47-67,399-561 (same indices) */
} orc_counters;

/* rnd.h */
uint32_t orc_wang_hash(uint32_t seed);
uint32_t orc_pixel_seed(uint32_t pixel_id);                 /* kernels.cu:542 */
uint32_t orc_xor_shift_32(uint32_t* state);
float    orc_rnd(uint32_t* state);
void     orc_random_in_unit_disk(uint32_t* state, float out[3]);
void     orc_random_in_unit_sphere(uint32_t* state, float out[3]);

/* helper_structs.h:194-207 */
void orc_make_camera(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov,
                     float aspect, float aperture, float focus_dist, rt_camera* out);
/* camera.h:8-12; dir is the normalised ray direction (ray.h:9) */
void orc_get_ray(const rt_camera* c, float s, float t, uint32_t* state, float org[3], float dir[3]);

/* intersections.h; `dir` is normalised first exactly as ray's constructor does (ray.h:9) */
float orc_sphere_hit(const rt_sphere* s, const float org[3], const float dir_in[3], float t_min, float t_max);
float orc_triangle_hit(const rt_triangle* tri, const float org[3], const float dir_in[3], float t_min, float t_max,
                       float* hitU, float* hitV);
int   orc_hit_bbox(const float bmin[3], const float bmax[3], const float org[3], const float dir_in[3], float t_max);
float orc_hit_bbox_dist(const float bmin[3], const float bmax[3], const float org[3], const float dir_in[3], float t_max);
float orc_plane_hit(const rt_plane* p, const float org[3], const float dir_in[3], float t_min, float t_max);

/* material.h */
float orc_schlick(float cosine, float ref_idx);
void  orc_reflect(const float v[3], const float n[3], float out[3]);
void  orc_refract(const float uv[3], const float n[3], float etai_over_etat, float out[3]);

typedef struct orc_scatter {    /* scatter_info, helper_structs.h:38-46 */
    float wi[3];
    int32_t specular;
    float throughput[3];
    int32_t refracted;
    float t;
} orc_scatter;

/* scene_materials.h:13-20 on an intersection {t, normal, inside} */
void orc_material_scatter(float inters_t, const float normal[3], int inside, const float wo[3],
                          const rt_material* mat, const float color[3], uint32_t* rng, orc_scatter* out);

/* same with the hit point p (the checker preset reads it, material.h:33-36); mat->type may be a preset (RT_FLOOR_COAT ...) */
void orc_material_scatter_p(float inters_t, const float p[3], const float normal[3], int inside, const float wo[3],
                            const rt_material* mat, const float color[3], uint32_t* rng, orc_scatter* out);

/* kernels.cu:154-224; returns closest t (or t_max); *tri_id,*u,*v valid when result < t_max */
float orc_hit_bvh(const orc_scene* sc, const float org[3], const float dir_in[3], float t_min, float t_max,
                  int is_shadow, uint32_t* tri_id, float* u, float* v, orc_counters* cnt);

/* generateShadowRay, kernels.cu:363-393, on (path origin, path attenuation, shading normal, rng).  Returns 1 when a shadow ray is
 * generated; out9 = {shadowDir xyz, lightContribution xyz, lightDist, cosAMax, draws consumed}; *rng is advanced. */
int orc_generate_shadow_ray(const rt_render_options* opt, const float origin[3], const float attenuation[3], const float normal[3],
                            uint32_t* rng, float out9[9]);

/* kernels.cu:535-569 restricted to the pixel rectangle [x0,x1) x [y0,y1); fb is the FULL nx*ny
 * framebuffer (untouched outside the rectangle). counters may be NULL. */
void orc_render(const orc_scene* sc, const rt_camera* cam, const rt_render_options* opt,
                int nx, int ny, int ns, int max_depth,
                int x0, int y0, int x1, int y1, rt_vec3* fb, orc_counters* counters);

/* staircase_scene.h:22-30 */
uint32_t orc_linear_to_srgb(float x);

/* main.cpp:117-125 */
double orc_rmse(const rt_vec3* f, const rt_vec3* g, int n);

/* {sizeof orc_scene, orc_counters, orc_scatter, rt_render_options} as this library was compiled; returns 4 (binding handshake) */
int orc_abi_sizes(int32_t* out, int n);

#ifdef __cplusplus
}
#endif
#endif
