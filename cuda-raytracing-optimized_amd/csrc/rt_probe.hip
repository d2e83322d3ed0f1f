// rt_probe.hip — per-function device probes behind the C-ABI (include/rt_probe.h).
//
// Each probe runs ONE device function of the hot path (rt_device.h) on arrays of inputs, one lane per
// element, and returns the outputs to the host, so the parity tests can compare every row of the
// hot-path table (RNG, get_ray, sphereHit, triangleHit, hit_bbox(_dist), BSDFs) against
// the CPU oracle in isolation.  Compiled twice like the render kernels (PARITY / FAST).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/rt_probe.h"
#include "rt_device.h"

using namespace rtd;

#if defined(RT_MODE_PARITY)
#define PROBE(name) name##_parity
#elif defined(RT_MODE_FAST)
#define PROBE(name) name##_fast
#else
#error "define RT_MODE_PARITY or RT_MODE_FAST"
#endif

namespace {

#define HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error = %s at %s:%d '%s' \n", hipGetErrorString(e_), __FILE__, __LINE__, #expr); exit(99); } } while (0)

// device buffer initialised from (or copied back to) a host array
template <typename T>
struct Buf {
    T* d = nullptr;
    T* h;
    size_t n;
    bool out;
    Buf(const T* host, size_t count) : h(const_cast<T*>(host)), n(count), out(false) {
        HIP_CHECK(hipMalloc((void**)&d, bytes()));
        HIP_CHECK(hipMemcpy(d, h, bytes(), hipMemcpyHostToDevice));
    }
    Buf(T* host, size_t count, bool) : h(host), n(count), out(true) {
        HIP_CHECK(hipMalloc((void**)&d, bytes()));
        HIP_CHECK(hipMemset(d, 0, bytes()));
    }
    size_t bytes() const { return (n ? n : 1) * sizeof(T); }
    ~Buf() {
        if (out) HIP_CHECK(hipMemcpy(h, d, n * sizeof(T), hipMemcpyDeviceToHost));
        HIP_CHECK(hipFree(d));
    }
};
template <typename T> Buf<T> in(const T* p, size_t n) { return Buf<T>(p, n); }
template <typename T> Buf<T> outb(T* p, size_t n) { return Buf<T>(p, n, true); }

inline dim3 grid_for(int n) { return dim3((n + 255) / 256); }
inline void sync() { HIP_CHECK(hipGetLastError()); HIP_CHECK(hipDeviceSynchronize()); }

__device__ __forceinline__ f3 ld(const float* p, int k) { return F3(p[3 * k], p[3 * k + 1], p[3 * k + 2]); }
__device__ __forceinline__ void stv(float* p, int k, f3 v) { p[3 * k] = v.x; p[3 * k + 1] = v.y; p[3 * k + 2] = v.z; }

__global__ void k_rng(const uint32_t* ids, int n, uint32_t* seed, float* draws, uint32_t* state) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t st = pixel_seed(ids[k]);
    seed[k] = st;
    for (int q = 0; q < 4; q++) draws[4 * k + q] = rnd(st);
    state[k] = st;
}

__global__ void k_disk_sphere(const uint32_t* states, int n, float* disk, uint32_t* st_disk, float* sph, uint32_t* st_sph) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t st = states[k];
    stv(disk, k, random_in_unit_disk(st));
    st_disk[k] = st;
    st = states[k];
    stv(sph, k, random_in_unit_sphere(st));
    st_sph[k] = st;
}

__global__ void k_get_ray(rt_camera cam, const float* s, const float* t, const uint32_t* states, int n,
                          float* org, float* dir, uint32_t* st_after) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t st = states[k];
    f3 o, d;
    get_ray(cam, s[k], t[k], st, o, d);
    stv(org, k, o);
    stv(dir, k, unit(d));
    st_after[k] = st;
}

__global__ void k_sphere_hit(const rt_sphere* sp, const float* org, const float* dir, const float* tmin, const float* tmax, int n, float* t_out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const Ray r = make_ray(ld(org, k), ld(dir, k));
    t_out[k] = sphere_hit(ld3(sp[k].center), sp[k].radius, r, tmin[k], tmax[k]);
}

__global__ void k_triangle_hit(const rt_triangle* tris, const float* org, const float* dir, const float* tmin, const float* tmax, int n,
                               float* t_out, float* u_out, float* v_out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const Ray r = make_ray(ld(org, k), ld(dir, k));
    float u = 0.0f, v = 0.0f;
    t_out[k] = triangle_hit(ld3(tris[k].v[0]), ld3(tris[k].v[1]), ld3(tris[k].v[2]), r, tmin[k], tmax[k], u, v);
    u_out[k] = u; v_out[k] = v;
}

__global__ void k_bbox(const float* bmin, const float* bmax, const float* org, const float* dir, const float* tmax, int n,
                       float* dist_out, int* hit_out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const Ray r = make_ray(ld(org, k), ld(dir, k));
    dist_out[k] = hit_bbox_dist(ld(bmin, k), ld(bmax, k), r, tmax[k]);
    hit_out[k] = hit_bbox(ld(bmin, k), ld(bmax, k), r, tmax[k]) ? 1 : 0;
}

__global__ void k_scatter(const float* t, const float* hp, const float* normal, const int* inside, const float* wo, const rt_material* mats,
                          const float* color, const uint32_t* states, int n,
                          float* wi, float* throughput, int* flags, float* t_out, uint32_t* st_after) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t st = states[k];
    Scatter sc;
    sc.wi = F3(0, 0, 0);
    material_scatter(sc, t[k], ld(hp, k), ld(normal, k), inside[k] != 0, ld(wo, k), mats[k].type, ld(color, k), mats[k].param, st);
    stv(wi, k, sc.wi);
    stv(throughput, k, sc.throughput);
    flags[k] = (sc.specular ? 1 : 0) | (sc.refracted ? 2 : 0);
    t_out[k] = sc.t;
    st_after[k] = st;
}

__global__ void k_math(const float* a, const float* b, int n, float* quot, float* root, float* p5, float* unit3) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    quot[k] = (F3(a[k], 0.0f, 1.0f) / b[k]).x;                     // vec3 / float as the path computes it (rt_div64.h)
    root[k] = rt_sqrt(fabsf(a[k]));
    p5[k] = pow5(a[k]);
    stv(unit3, k, unit(F3(a[k], b[k], a[k] - b[k])));
}

__global__ void k_sincos(const float* y, int n, float* s_out, float* c_out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    float s, c;
    rt_sincosf(y[k], s, c);
    s_out[k] = s; c_out[k] = c;
}

__global__ void k_schlick(const float* cosine, const float* ref_idx, const float* u, int n, float* out, int* above) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    out[k] = schlick(cosine[k], ref_idx[k]);
    if (u && above) above[k] = schlick_above(u[k], cosine[k], ref_idx[k]) ? 1 : 0;
}

__global__ void k_shadow_ray(rt_sphere light, rt_vec3 lightColor, const float* org, const float* atten, const float* normal,
                             const uint32_t* states, int n, float* out9, int* ok, uint32_t* st_after) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t st = states[k];
    ShadowSample s;
    s.dir = F3(0, 0, 0); s.contrib = F3(0, 0, 0); s.dist = 0.0f; s.cosAMax = 0.0f;
    const bool g = generate_shadow_ray(ld3(light.center), light.radius, ld3(lightColor), ld(org, k), ld(atten, k), ld(normal, k), st, s);
    uint32_t probe = states[k];
    int draws = 0;
    while (probe != st && draws < 4) { (void)rnd(probe); draws++; }
    float* o = out9 + 9 * k;
    o[0] = s.dir.x; o[1] = s.dir.y; o[2] = s.dir.z; o[3] = s.contrib.x; o[4] = s.contrib.y; o[5] = s.contrib.z;
    o[6] = s.dist; o[7] = s.cosAMax; o[8] = (float)draws;
    ok[k] = g ? 1 : 0;
    st_after[k] = st;
}

__global__ void k_plane_hit(const rt_plane* pl, const float* org, const float* dir, const float* tmin, const float* tmax, int n, float* t_out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const Ray r = make_ray(ld(org, k), ld(dir, k));
    t_out[k] = plane_hit(ld3(pl[k].norm), ld3(pl[k].point), r, tmin[k], tmax[k]);
}

}  // namespace

extern "C" {

void PROBE(rtProbeShadowRay)(const rt_sphere* light, const rt_vec3* lightColor, const float* org3, const float* atten3, const float* normal3,
                             const uint32_t* states, int n, float* out9, int* generated, uint32_t* st_after) {
    auto a = in(org3, (size_t)3 * n); auto b = in(atten3, (size_t)3 * n); auto c = in(normal3, (size_t)3 * n); auto d = in(states, n);
    auto e = outb(out9, (size_t)9 * n); auto f = outb(generated, n); auto g = outb(st_after, n);
    hipLaunchKernelGGL(k_shadow_ray, grid_for(n), dim3(256), 0, 0, *light, *lightColor, a.d, b.d, c.d, d.d, n, e.d, f.d, g.d);
    sync();
}

void PROBE(rtProbePlaneHit)(const rt_plane* planes, const float* org3, const float* dir3, const float* tmin, const float* tmax, int n, float* t_out) {
    auto a = in(planes, n); auto b = in(org3, (size_t)3 * n); auto c = in(dir3, (size_t)3 * n); auto d = in(tmin, n); auto e = in(tmax, n);
    auto f = outb(t_out, n);
    hipLaunchKernelGGL(k_plane_hit, grid_for(n), dim3(256), 0, 0, a.d, b.d, c.d, d.d, e.d, n, f.d);
    sync();
}

void PROBE(rtProbeRng)(const uint32_t* pixel_ids, int n, uint32_t* seed_out, float* draws4_out, uint32_t* state_out) {
    auto a = in(pixel_ids, n); auto b = outb(seed_out, n); auto c = outb(draws4_out, (size_t)4 * n); auto d = outb(state_out, n);
    hipLaunchKernelGGL(k_rng, grid_for(n), dim3(256), 0, 0, a.d, n, b.d, c.d, d.d);
    sync();
}

void PROBE(rtProbeDiskSphere)(const uint32_t* states, int n, float* disk3, uint32_t* st_after_disk, float* sphere3, uint32_t* st_after_sphere) {
    auto a = in(states, n); auto b = outb(disk3, (size_t)3 * n); auto c = outb(st_after_disk, n);
    auto d = outb(sphere3, (size_t)3 * n); auto e = outb(st_after_sphere, n);
    hipLaunchKernelGGL(k_disk_sphere, grid_for(n), dim3(256), 0, 0, a.d, n, b.d, c.d, d.d, e.d);
    sync();
}

void PROBE(rtProbeGetRay)(const rt_camera* cam, const float* s, const float* t, const uint32_t* states, int n,
                          float* org3, float* dir3, uint32_t* st_after) {
    auto a = in(s, n); auto b = in(t, n); auto c = in(states, n);
    auto d = outb(org3, (size_t)3 * n); auto e = outb(dir3, (size_t)3 * n); auto f = outb(st_after, n);
    hipLaunchKernelGGL(k_get_ray, grid_for(n), dim3(256), 0, 0, *cam, a.d, b.d, c.d, n, d.d, e.d, f.d);
    sync();
}

void PROBE(rtProbeSphereHit)(const rt_sphere* spheres, const float* org3, const float* dir3, const float* tmin, const float* tmax,
                             int n, float* t_out) {
    auto a = in(spheres, n); auto b = in(org3, (size_t)3 * n); auto c = in(dir3, (size_t)3 * n); auto d = in(tmin, n); auto e = in(tmax, n);
    auto f = outb(t_out, n);
    hipLaunchKernelGGL(k_sphere_hit, grid_for(n), dim3(256), 0, 0, a.d, b.d, c.d, d.d, e.d, n, f.d);
    sync();
}

void PROBE(rtProbeTriangleHit)(const rt_triangle* tris, const float* org3, const float* dir3, const float* tmin, const float* tmax,
                               int n, float* t_out, float* u_out, float* v_out) {
    auto a = in(tris, n); auto b = in(org3, (size_t)3 * n); auto c = in(dir3, (size_t)3 * n); auto d = in(tmin, n); auto e = in(tmax, n);
    auto f = outb(t_out, n); auto g = outb(u_out, n); auto h = outb(v_out, n);
    hipLaunchKernelGGL(k_triangle_hit, grid_for(n), dim3(256), 0, 0, a.d, b.d, c.d, d.d, e.d, n, f.d, g.d, h.d);
    sync();
}

void PROBE(rtProbeBbox)(const float* bmin3, const float* bmax3, const float* org3, const float* dir3, const float* tmax,
                        int n, float* dist_out, int* hit_out) {
    auto a = in(bmin3, (size_t)3 * n); auto b = in(bmax3, (size_t)3 * n); auto c = in(org3, (size_t)3 * n); auto d = in(dir3, (size_t)3 * n);
    auto e = in(tmax, n); auto f = outb(dist_out, n); auto g = outb(hit_out, n);
    hipLaunchKernelGGL(k_bbox, grid_for(n), dim3(256), 0, 0, a.d, b.d, c.d, d.d, e.d, n, f.d, g.d);
    sync();
}

void PROBE(rtProbeScatter)(const float* t, const float* p3, const float* normal3, const int* inside, const float* wo3, const rt_material* mats,
                           const float* color3, const uint32_t* states, int n,
                           float* wi3, float* throughput3, int* flags, float* t_out, uint32_t* st_after) {
    auto a = in(t, n); auto pp = in(p3, (size_t)3 * n); auto b = in(normal3, (size_t)3 * n); auto c = in(inside, n); auto d = in(wo3, (size_t)3 * n);
    auto e = in(mats, n); auto f = in(color3, (size_t)3 * n); auto g = in(states, n);
    auto h = outb(wi3, (size_t)3 * n); auto i = outb(throughput3, (size_t)3 * n); auto j = outb(flags, n);
    auto k = outb(t_out, n); auto l = outb(st_after, n);
    hipLaunchKernelGGL(k_scatter, grid_for(n), dim3(256), 0, 0, a.d, pp.d, b.d, c.d, d.d, e.d, f.d, g.d, n, h.d, i.d, j.d, k.d, l.d);
    sync();
}

void PROBE(rtProbeMath)(const float* a_, const float* b_, int n, float* quot, float* root, float* p5, float* unit3) {
    auto a = in(a_, n); auto b = in(b_, n);
    auto c = outb(quot, n); auto d = outb(root, n); auto e = outb(p5, n); auto f = outb(unit3, (size_t)3 * n);
    hipLaunchKernelGGL(k_math, grid_for(n), dim3(256), 0, 0, a.d, b.d, n, c.d, d.d, e.d, f.d);
    sync();
}

void PROBE(rtProbeSinCos)(const float* y_, int n, float* s_out, float* c_out) {
    auto y = in(y_, n);
    auto a = outb(s_out, n); auto b = outb(c_out, n);
    hipLaunchKernelGGL(k_sincos, grid_for(n), dim3(256), 0, 0, y.d, n, a.d, b.d);
    sync();
}

void PROBE(rtProbeSchlick)(const float* cosine_, const float* ref_idx_, const float* u_, int n, float* out_, int* above_) {
    auto a = in(cosine_, n); auto b = in(ref_idx_, n); auto u = in(u_, n);
    auto c = outb(out_, n); auto d = outb(above_, n);
    hipLaunchKernelGGL(k_schlick, grid_for(n), dim3(256), 0, 0, a.d, b.d, u.d, n, c.d, d.d);
    sync();
}

}  // extern "C"
