/*
 * rt_host.h — C-ABI of librt_host.so: the host-side pieces that sit either side of the render
 * hot path in the reference's main.cpp / staircase_scene.h (camera construction, scene set-up,
 * BVH file + builder, PPM / .ref output, RMSE).  Plain C++ inside, no HIP, no torch: loads on a
 * machine without a GPU.
 *
 * Scene and camera are produced ONCE here and handed as data to every backend (HIP library,
 * CPU oracle), never recomputed per side.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* camera::camera, /root/reference/helper_structs.h:194-207 (runs on the host there too). */
void rtMakeCamera(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov,
                  float aspect, float aperture, float focus_dist, rt_camera* out);

/* random_float, /root/reference/main.cpp:17-20: state = 214013*state + 2531011;
 * return ((state >> 16) & 0x7FFF) / 32767.f */
float rtRandomFloat(uint32_t* state);

/* C1: ground + glass + metal; camera lookfrom(0,0,1) lookat(0,0,-1) vfov 60
 * aperture 0 focus 2.  Writes 3 spheres / 3 materials; returns 3. */
int rtSceneThreeSpheres(rt_sphere* spheres, rt_material* materials, int cap, int nx, int ny, rt_camera* cam);

/* C2/C3/C5: the "Ray Tracing in One Weekend" cover scene as the README-era
 * renderer built it — ground, 22x22 grid of small spheres jittered by the main.cpp LCG
 * (seeded with `seed`, 0 for the benchmark), three big spheres: 488 spheres.  Camera
 * lookfrom(13,2,3) lookat(0,0,0) vfov 30 aperture 0.1 focus 10, aspect nx/ny.
 * Returns the number of spheres written (488), or -needed if cap is too small. */
int rtSceneRandomSpheres(uint32_t seed, rt_sphere* spheres, rt_material* materials, int cap,
                         int nx, int ny, rt_camera* cam);

/* setup_camera, /root/reference/staircase_scene.h:62-73. */
void rtStaircaseCamera(int nx, int ny, rt_camera* cam);

/* ---- mesh scenes: BVH in the layout hitBvh expects (kernels.cu:154-224) ---- */

typedef struct rt_host_mesh rt_host_mesh;    /* opaque; owns tris + bvh */

/* Builds the complete-binary-tree BVH over `tris` (copied): heap-indexed nodes, root at index 1,
 * node 0 unused, leaves = power of two, nppl triangles per leaf padded with sentinel triangles
 * whose v[0].x is +inf (kernels.cu:202).  Returns NULL on bad arguments. */
rt_host_mesh* rtBuildBvh(const rt_triangle* tris, int num_tris, int nppl);
/* The same with `extra_levels` tree levels beyond the smallest complete tree that holds the triangles (rtBuildBvh: 1): more leaf
 * slots for the SAH cuts to use - fewer node visits and triangle tests per ray, twice the (small) arrays per level. */
rt_host_mesh* rtBuildBvhLevels(const rt_triangle* tris, int num_tris, int nppl, int extra_levels);

/* loadBVH, /root/reference/staircase_scene.h:75-101: file "BVH_00.04\0", int numTris,
 * triangle[numTris], int numBvhNodes, bvh_node[numBvhNodes], vec3 min, vec3 max, int nppl. */
rt_host_mesh* rtLoadBvhFile(const char* path);
int  rtSaveBvhFile(const rt_host_mesh* m, const char* path);        /* 0 on success */
void rtFreeMesh(rt_host_mesh* m);
/* Fills an rt_mesh view (pointers stay owned by m) and returns nppl. */
int  rtMeshView(const rt_host_mesh* m, rt_mesh* out);

/* Procedural stand-in for the staircase asset (absent from the reference snapshot, SURVEY.md
 * §7.4 H6): a room with a flight of steps, a glass ball, a metal ball and boxes, sized to the
 * staircase camera and light of staircase_scene.h:62-73 / kernels.cu:93.  Writes up to cap
 * triangles and 20 materials (indices as staircase_scene.h:139-158, texId = -1 unless
 * with_textures); returns the triangle count (or -needed). `detail` >= 1 scales tessellation. */
int rtSceneStaircaseProcedural(int detail, rt_triangle* tris, int cap, rt_material* materials20);

/* ---- output / verification harness (main.cpp:25-60,105-128; staircase_scene.h:22-43) ---- */

uint32_t rtLinearToSRGB(float x);
/* P3 PPM, rows top to bottom (j = ny-1 .. 0). 0 on success. */
int rtWritePPM(const char* path, int nx, int ny, const rt_vec3* colors);
/* "REF_00.01\0", int nx, int ny, vec3[nx*ny]. 0 on success; load returns -1 missing/bad header, -2 size mismatch. */
int rtSaveReference(const char* path, int nx, int ny, const rt_vec3* colors);
int rtLoadReference(const char* path, rt_vec3* reference, int nx, int ny);
/* sqrt( sum_i sum_c (f-g)^2 / 3 / (nx*ny) ), accumulated in double. */
double rtRmse(const rt_vec3* f, const rt_vec3* g, int nx, int ny);

#ifdef __cplusplus
}
#endif
#endif
