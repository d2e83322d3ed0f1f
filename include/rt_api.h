/*
 * rt_api.h — C-ABI of librt_mi355x.so, the MI355X-native replacement for the host API of the
 * reference renderer (/root/reference/kernels.h:6-8, defined in kernels.cu:571-680).
 *
 * The first three entry points are the reference's own symbols, with byte-identical signatures
 * (struct layouts in rt_types.h); the rest are additive.  No torch types, no C++ types.
 *
 * Error convention (kernels.cu:27-38): every function returns void; any HIP runtime failure
 * prints "HIP error = <string> at <file>:<line> '<expr>'" on stderr and calls exit(99).
 * Misuse (run before init, bad sizes) prints "rt error: ..." and exit(99) as well.
 *
 * State (kernels.cu:145): one global render context per process; not re-entrant; the caller is
 * single-threaded.  runRenderer is synchronous; on return *fb is host-readable.
 */
#ifndef RT_API_H
#define RT_API_H

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* --- reference symbols ------------------------------------------------------------------- */

/* replaces kernels.cu:571-650.  Deep-copies sc.m->tris, sc.m->bvh, sc.materials and
 * sc.textures[i].data to the device (borrowed only for the duration of the call), allocates the
 * framebuffer (nx*ny vec3, linear RGB, row 0 = bottom row) and returns it through *fb.  The
 * framebuffer is host memory owned by the renderer, valid until cleanupRenderer. */
void initRenderer(const rt_kernel_scene sc, const rt_camera cam, rt_vec3** fb, int nx, int ny, int maxDepth);

/* replaces kernels.cu:652-664.  Renders ns samples per pixel into the framebuffer; blocking.
 * tx,ty are the reference's CUDA block shape (main.cpp:69-70); accepted and ignored — the
 * wave64 tile shape is fixed by the kernel. May be called repeatedly: same image every time. */
void runRenderer(int ns, int tx, int ty);

/* replaces kernels.cu:666-680.  Frees everything, the framebuffer included. */
void cleanupRenderer(void);

/* --- additive symbols -------------------------------------------------------------------- */

/* Sphere scenes (the README-era benchmark; kernel_scene has no sphere array,
 * helper_structs.h:217-228).  materials[k] belongs to spheres[k].  Defaults for this path:
 * gradient sky, NEE off, RR off, t_min 0.001. */
void initRendererSpheres(const rt_sphere* spheres, const rt_material* materials, int n,
                         const rt_camera cam, rt_vec3** fb, int nx, int ny, int maxDepth);

/* Fills *opt with the defaults of the mesh path (is_sphere_scene = 0: kernels.cu:13-24,93-94)
 * or of the sphere path (is_sphere_scene = 1). */
void getDefaultRenderOptions(rt_render_options* opt, int is_sphere_scene);

/* Takes effect for the following runRenderer calls.  Call after init*. */
void setRenderOptions(const rt_render_options* opt);

/* Makes the following runRenderer calls deliver into caller-owned host memory (nx*ny vec3) instead of the
 * library's own framebuffer: the buffer is page-locked with hipHostRegister and becomes the target of the
 * device-to-host stripe copies.  Used for the multi-process host gather: every rank passes the same shared
 * mapping and writes only the stripes it owns.  NULL switches back.  The caller keeps ownership.
 * If the runtime cannot page-lock the memory, a warning is printed and the stripes are copied into it as pageable
 * memory (same image, no direct delivery by the kernel): never a reason to exit. */
void setExternalFramebuffer(rt_vec3* fb);

/* Timing / counters of the last runRenderer. */
void getRenderStats(rt_render_stats* out);

/* Number of HIP devices visible to the process (0 if none). Never exits. */
int rtDeviceCount(void);

/* Library/ABI version: major*1000 + minor.  Bumped whenever rt_render_options or rt_render_stats (or any other struct of rt_types.h)
 * changes size or layout; a binding refuses a library whose version differs from the RT_API_VERSION it was written against. */
#define RT_API_VERSION 1002
int rtApiVersion(void);

/* sizeof of every struct that crosses this boundary, as the LIBRARY was compiled: out[RT_SIZEOF_*], at most n entries written; returns
 * RT_SIZEOF_COUNT.  getDefaultRenderOptions / getRenderStats write sizeof(struct) bytes through the caller's pointer, so a binding checks
 * these against its own mirror before the first such call (the Python mirror does so in load_renderer(): a mismatch is an ImportError). */
enum { RT_SIZEOF_RENDER_OPTIONS = 0, RT_SIZEOF_RENDER_STATS, RT_SIZEOF_CAMERA, RT_SIZEOF_SPHERE, RT_SIZEOF_MATERIAL, RT_SIZEOF_TRIANGLE,
       RT_SIZEOF_BVH_NODE, RT_SIZEOF_MESH, RT_SIZEOF_KERNEL_SCENE, RT_SIZEOF_STEXTURE, RT_SIZEOF_PLANE, RT_SIZEOF_BBOX, RT_SIZEOF_VEC3,
       RT_SIZEOF_COUNT };
int rtStructSizes(int32_t* out, int n);

#ifdef __cplusplus
}
#endif

#endif /* RT_API_H */
