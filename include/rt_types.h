/*
 * rt_types.h — plain-old-data types shared by the C-ABI boundary, the CPU oracle and the HIP kernels.
 *
 * Every struct here has the SAME size, alignment and member offsets as the type of the same
 * name in the reference (x86-64 SysV), so a host program compiled against the reference's
 * own headers can link against our shared library unchanged:
 *
 *   vec3          /root/reference/vec3.h:9-41            12 B  {float e[3]}
 *   camera        /root/reference/helper_structs.h:191-215  88 B
 *   sphere        /root/reference/helper_structs.h:168-175  16 B
 *   plane         /root/reference/helper_structs.h:159-166  24 B
 *   bbox          /root/reference/helper_structs.h:73-79    24 B
 *   triangle      /root/reference/helper_structs.h:81-96    64 B
 *   bvh_node      /root/reference/helper_structs.h:98-110   24 B
 *   material      /root/reference/helper_structs.h:127-138  24 B
 *   stexture      /root/reference/helper_structs.h:140-147  16 B
 *   mesh          /root/reference/helper_structs.h:112-125  56 B
 *   kernel_scene  /root/reference/helper_structs.h:217-228  64 B
 *
 * The reference types are C++ classes with trivial copy constructors/destructors (mesh has a
 * destructor but is only ever passed by pointer), so by-value arguments >16 B travel in memory
 * exactly like the C structs below.  Names are prefixed rt_ so this header can be included next
 * to the reference's headers (the drop-in link test does that); the layout is what matters.
 *
 * Plain C99 / C++ / HIP compatible: no constructors, no methods.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
#define RT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define RT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

typedef struct rt_vec3 { float e[3]; } rt_vec3;

typedef struct rt_camera {
    rt_vec3 origin;
    rt_vec3 lower_left_corner;
    rt_vec3 horizontal;
    rt_vec3 vertical;
    rt_vec3 u, v, w;
    float lens_radius;
} rt_camera;

typedef struct rt_sphere { rt_vec3 center; float radius; } rt_sphere;

typedef struct rt_plane { rt_vec3 norm; rt_vec3 point; } rt_plane;

typedef struct rt_bbox { rt_vec3 min; rt_vec3 max; } rt_bbox;

typedef struct rt_triangle {
    rt_vec3 v[3];
    float texCoords[6];
    unsigned char meshID;
    unsigned char _pad[3];
} rt_triangle;

typedef struct rt_bvh_node { rt_vec3 a; rt_vec3 b; } rt_bvh_node;

/* material_type, helper_structs.h:127-131 */
enum { RT_DIFFUSE = 0, RT_METAL = 1, RT_GLASS = 2,
       /* Additive: the reference's dormant look presets, scene_materials.h:22-93 — hard-coded
        * parameters, reachable from no scene at HEAD; material.color / param are ignored for them. */
       RT_FLOOR_COAT = 3,        /* floor_coat_scatter        :22-28  coat_bsdf(ior 1.5, base 0x511845)            */
       RT_FLOOR_DIFFUSE = 4,     /* floor_diffuse_scatter     :30-33                                               */
       RT_FLOOR_CHECKER = 5,     /* floor_checker_scatter     :35-44  checker_layer(0.2) 0x511845 / 0xff5733       */
       RT_MODEL_COAT = 6,        /* model_coat_scatter        :46-52  coat_bsdf(ior 1.1)                           */
       RT_MODEL_DIFFUSE = 7,     /* model_diffuse_scatter     :54-57                                               */
       RT_MODEL_GLOSSY = 8,      /* model_glossy_scatter      :59-63                                               */
       RT_MODEL_GLASS = 9,       /* model_glass_scatter       :65-71  dielectric ior 1.1                           */
       RT_MODEL_TINTEDGLASS = 10,/* model_tintedglass_scatter :73-81  Beer-Lambert absorption                      */
       RT_MODEL_SSS = 11,        /* model_sss_scatter         :83-93  subsurface_dielectric_bsdf                   */
       RT_MATERIAL_TYPE_COUNT = 12 };

typedef struct rt_material {
    int32_t type;       /* RT_DIFFUSE / RT_METAL / RT_GLASS */
    rt_vec3 color;
    float param;        /* METAL: fuzz; GLASS: index of refraction */
    int32_t texId;      /* -1 = use color */
} rt_material;

typedef struct rt_stexture {
    float* data;        /* width*height RGB float triplets */
    int32_t width;
    int32_t height;
} rt_stexture;

typedef struct rt_mesh {
    rt_triangle* tris;
    uint32_t numTris;
    rt_bvh_node* bvh;
    int32_t numBvhNodes;
    rt_bbox bounds;
} rt_mesh;

typedef struct rt_kernel_scene {
    rt_mesh* m;
    rt_plane floor;
    rt_material* materials;
    int32_t numMaterials;
    rt_stexture* textures;
    int32_t numTextures;
    int32_t numPrimitivesPerLeaf;
} rt_kernel_scene;

RT_STATIC_ASSERT(sizeof(rt_vec3) == 12, "vec3");
RT_STATIC_ASSERT(sizeof(rt_camera) == 88, "camera");
RT_STATIC_ASSERT(offsetof(rt_camera, lower_left_corner) == 12, "camera.llc");
RT_STATIC_ASSERT(offsetof(rt_camera, horizontal) == 24, "camera.horizontal");
RT_STATIC_ASSERT(offsetof(rt_camera, vertical) == 36, "camera.vertical");
RT_STATIC_ASSERT(offsetof(rt_camera, u) == 48, "camera.u");
RT_STATIC_ASSERT(offsetof(rt_camera, w) == 72, "camera.w");
RT_STATIC_ASSERT(offsetof(rt_camera, lens_radius) == 84, "camera.lens_radius");
RT_STATIC_ASSERT(sizeof(rt_sphere) == 16, "sphere");
RT_STATIC_ASSERT(sizeof(rt_plane) == 24, "plane");
RT_STATIC_ASSERT(sizeof(rt_bbox) == 24, "bbox");
RT_STATIC_ASSERT(sizeof(rt_triangle) == 64, "triangle");
RT_STATIC_ASSERT(offsetof(rt_triangle, texCoords) == 36, "triangle.texCoords");
RT_STATIC_ASSERT(offsetof(rt_triangle, meshID) == 60, "triangle.meshID");
RT_STATIC_ASSERT(sizeof(rt_bvh_node) == 24, "bvh_node");
RT_STATIC_ASSERT(sizeof(rt_material) == 24, "material");
RT_STATIC_ASSERT(offsetof(rt_material, color) == 4, "material.color");
RT_STATIC_ASSERT(offsetof(rt_material, param) == 16, "material.param");
RT_STATIC_ASSERT(offsetof(rt_material, texId) == 20, "material.texId");
RT_STATIC_ASSERT(sizeof(rt_stexture) == 16, "stexture");
RT_STATIC_ASSERT(sizeof(rt_mesh) == 56, "mesh");
RT_STATIC_ASSERT(offsetof(rt_mesh, bvh) == 16, "mesh.bvh");
RT_STATIC_ASSERT(offsetof(rt_mesh, numBvhNodes) == 24, "mesh.numBvhNodes");
RT_STATIC_ASSERT(offsetof(rt_mesh, bounds) == 28, "mesh.bounds");
RT_STATIC_ASSERT(sizeof(rt_kernel_scene) == 64, "kernel_scene");
RT_STATIC_ASSERT(offsetof(rt_kernel_scene, floor) == 8, "kernel_scene.floor");
RT_STATIC_ASSERT(offsetof(rt_kernel_scene, materials) == 32, "kernel_scene.materials");
RT_STATIC_ASSERT(offsetof(rt_kernel_scene, numMaterials) == 40, "kernel_scene.numMaterials");
RT_STATIC_ASSERT(offsetof(rt_kernel_scene, textures) == 48, "kernel_scene.textures");
RT_STATIC_ASSERT(offsetof(rt_kernel_scene, numTextures) == 56, "kernel_scene.numTextures");
RT_STATIC_ASSERT(offsetof(rt_kernel_scene, numPrimitivesPerLeaf) == 60, "kernel_scene.nppl");

/* ------------------------------------------------------------------------------------------
 * Additive types (not in the reference): run-time switches for what the reference fixes with
 * compile-time #defines (kernels.cu:13-24), plus the multi-GPU stripe partition.
 * ------------------------------------------------------------------------------------------ */

enum { RT_SKY_CONST_GREY = 0,   /* kernels.cu:424   color += attenuation * 0.5            */
       RT_SKY_GRADIENT   = 1 }; /* kernels.cu:419-421 (commented variant; sphere scenes)  */

enum { RT_RNG_REFERENCE_STREAM = 0,  /* xorshift32 per pixel, rnd.h:5-18, kernels.cu:541-542 */
       RT_RNG_COUNTER          = 1 };/* (pixel,sample)-keyed hash stream; statistical parity  */

enum { RT_FP_PARITY = 0,   /* no FMA contraction, literal operation order: bit-exact vs oracle */
       RT_FP_FAST   = 1 }; /* FMA + hoisted invariants; parity within stated tolerance         */

#define RT_MAX_DEVICES 8

typedef struct rt_render_options {
    int32_t sky;            /* RT_SKY_*                                                        */
    int32_t nee;            /* 1 = next-event estimation towards `light` (#define SHADOW)      */
    int32_t rr;             /* 1 = Russian roulette after bounce 3 (#define RUSSIAN_ROULETTE)  */
    float   t_min;          /* kernels.cu:19 EPSILON 0.01f (mesh); 0.001f for sphere scenes    */
    int32_t rng;            /* RT_RNG_*                                                        */
    int32_t fp;             /* RT_FP_*                                                         */
    rt_sphere light;        /* kernels.cu:93                                                   */
    rt_vec3 lightColor;     /* kernels.cu:94                                                   */
    int32_t stripe_rows;    /* rows per interleaved stripe (multi-GPU partition), default 8    */
    int32_t num_devices;    /* in-process devices to spread stripes over; 0/1 = current device */
    int32_t devices[RT_MAX_DEVICES];
    int32_t part_rank;      /* this process renders stripes k with k % part_world == part_rank */
    int32_t part_world;     /* 1 = whole image                                                 */
    int32_t variant;        /* kernel variant selector, 0 = default            */
    int32_t counters;       /* 1 = count rays / primitive tests / node visits (getRenderStats) */
    int32_t samples_per_item; /* RT_RNG_COUNTER only: samples per work item (a pixel's samples are independent there and
                               are split over lanes; partial sums are added in chunk order). 0 = default (4).    */
    /* --- appended in API version 1001 --- */
    int32_t floor;          /* mesh scenes: 1 = rays that miss the mesh are tested against kernel_scene.floor (planeHit, intersections.h:43-52)
                               and scatter with floor_diffuse_scatter (scene_materials.h:30-33): the call site the reference keeps
                               commented out at HEAD (kernels.cu:341-345, 481-482).  Default 0 = HEAD behaviour.                 */
} rt_render_options;

/* The reference's `#ifdef STATS` ray statistics (kernels.cu:47-67: NUM_RAYS_* / NUM_NODES_*), same order.  Filled when
 * rt_render_options.counters = 1.  PLANE-related distinctions do not arise (the floor's call site is commented out at HEAD,
 * kernels.cu:341-345), so SECONDARY_MESH = SECONDARY and SECONDARY_NOHIT = 0 exactly as in a STATS build of HEAD. */
enum { RT_STAT_PRIMARY = 0, RT_STAT_PRIMARY_HIT_MESH, RT_STAT_PRIMARY_NOHITS, RT_STAT_PRIMARY_BBOX_NOHITS,
       RT_STAT_SECONDARY, RT_STAT_SECONDARY_MESH, RT_STAT_SECONDARY_NOHIT, RT_STAT_SECONDARY_MESH_NOHIT, RT_STAT_SECONDARY_BBOX_NOHIT,
       RT_STAT_SHADOWS, RT_STAT_SHADOWS_BBOX_NOHITS, RT_STAT_SHADOWS_NOHITS, RT_STAT_LOW_POWER, RT_STAT_EXCEED_MAX_BOUNCE,
       RT_STAT_RUSSIAN_KILL, RT_STAT_NAN, RT_STAT_NODES_BOTH, RT_STAT_NODES_SINGLE, RT_STAT_COUNT };

typedef struct rt_render_stats {
    double  kernel_ms;      /* HIP-event time of the render kernel(s) of the last runRenderer, max over devices */
    double  total_ms;       /* wall time of the last runRenderer incl. gather                                   */
    int64_t samples;        /* pixels*ns rendered by this process in the last runRenderer                       */
    int32_t num_launches;   /* kernel launches in the last runRenderer                                          */
    int32_t vgprs;          /* reserved */
    uint64_t rays;          /* rays traced (only when built/launched with counters on; else 0)                  */
    uint64_t prim_tests;    /* sphere or triangle tests                                                         */
    uint64_t node_visits;   /* BVH internal-node visits                                                         */
    uint64_t exec_tests;    /* sphere tests actually executed per lane after group culling (sphere scenes)     */
    /* --- appended in API version 1001 (older callers that pass the shorter struct must not call getRenderStats of 1001+) --- */
    uint64_t shadow_rays;   /* shadow rays traced (mesh scenes with NEE)                                        */
    uint64_t box_tests;     /* sphere scenes: group / node bounding-box slab tests executed by the culling      */
    uint64_t ref_stats[RT_STAT_COUNT];  /* the reference's STATS counters (kernels.cu:47-67), same indices, same meaning   */
} rt_render_stats;

#endif /* RT_TYPES_H */
