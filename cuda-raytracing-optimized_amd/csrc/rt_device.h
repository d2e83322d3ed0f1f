// rt_device.h — device-side fp32 vector math, RNG and BSDFs for the gfx950 render kernels.
//
// Operation order is part of the contract: in the PARITY build (-ffp-contract=off) every
// expression below rounds exactly like the reference's header-only code, so the HIP path is
// bit-identical to the CPU oracle.  Each helper cites the reference lines it implements
// (relative to /root/reference/).  The FAST build compiles the same source with FMA
// contraction enabled.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_types.h"

namespace rtd {

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 F3(float x, float y, float z) { return { x, y, z }; }
__device__ __forceinline__ f3 ld3(const rt_vec3& v) { return { v.e[0], v.e[1], v.e[2] }; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }   // vec3.h:59
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }   // vec3.h:63
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }   // vec3.h:67
__device__ __forceinline__ f3 operator*(float t, f3 a) { return { t * a.x, t * a.y, t * a.z }; }       // vec3.h:75,83
__device__ __forceinline__ f3 operator/(f3 a, float t) { return { a.x / t, a.y / t, a.z / t }; }       // vec3.h:79
__device__ __forceinline__ f3 operator-(f3 a) { return { -a.x, -a.y, -a.z }; }                         // vec3.h:23
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }         // vec3.h:87
__device__ __forceinline__ f3 cross(f3 a, f3 b) {                                                      // vec3.h:91
    return { (a.y * b.z - a.z * b.y), (-(a.x * b.z - a.z * b.x)), (a.x * b.y - a.y * b.x) };
}
__device__ __forceinline__ float sqlen(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }             // vec3.h:36
__device__ __forceinline__ float len(f3 a) { return __fsqrt_rn(a.x * a.x + a.y * a.y + a.z * a.z); }   // vec3.h:35
__device__ __forceinline__ f3 unit(f3 a) { return a / len(a); }                                        // vec3.h:194
__device__ __forceinline__ float max3(f3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }                    // vec3.h:113

// ---- rnd.h ------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t wang_hash(uint32_t seed) {     // rnd.h:31-39
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}
__device__ __forceinline__ uint32_t pixel_seed(uint32_t pixel_id) {   // kernels.cu:541-542
    return (wang_hash(pixel_id) * 336343633u) | 1u;
}
// RT_RNG_COUNTER: one stream per (pixel, sample); same formula as oracle/rt_oracle.c sample_seed
__device__ __forceinline__ uint32_t sample_seed(uint32_t pixel_id, uint32_t s) {
    return (wang_hash(pixel_id + wang_hash(s) * 0x9E3779B9u) * 336343633u) | 1u;
}
__device__ __forceinline__ float rnd(uint32_t& state) {             // rnd.h:5-18
    uint32_t x = state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 15;
    state = x;
    return (float)(x & 0xFFFFFF) / 16777216.0f;
}
__device__ __forceinline__ f3 random_in_unit_disk(uint32_t& state) {    // rnd.h:20-26, draws x then y
    f3 p;
    do {
        const float rx = rnd(state);
        const float ry = rnd(state);
        p = 2.0f * F3(rx, ry, 0.0f) - F3(1.0f, 1.0f, 0.0f);
    } while (dot(p, p) >= 1.0f);
    return p;
}
__device__ __forceinline__ f3 random_in_unit_sphere(uint32_t& state) {  // rnd.h:41-49, draws x,y,z
    f3 p;
    do {
        const float rx = rnd(state);
        const float ry = rnd(state);
        const float rz = rnd(state);
        p = 2.0f * F3(rx, ry, rz) - F3(1.0f, 1.0f, 1.0f);
    } while (sqlen(p) >= 1.0f);
    return p;
}

// ---- camera.h:8-12 ----------------------------------------------------------------------------
// Returns origin and the UN-normalised direction; the caller normalises as ray's ctor does (ray.h:9).
__device__ __forceinline__ void get_ray(const rt_camera& c, float s, float t, uint32_t& state, f3& org, f3& dir) {
    const f3 rd = c.lens_radius * random_in_unit_disk(state);
    const f3 offset = rd.x * ld3(c.u) + rd.y * ld3(c.v);
    org = ld3(c.origin) + offset;
    dir = ld3(c.lower_left_corner) + s * ld3(c.horizontal) + t * ld3(c.vertical) - ld3(c.origin) - offset;
}

// ---- material.h -------------------------------------------------------------------------------

// pow(x, 5.0f), material.h:12.  Evaluated in fp64 and rounded once: the correctly rounded value
// except with probability ~2^-29.  glibc's powf (what the oracle calls) differs from that in
// 0.07 % of arguments by one ulp, which can flip the `rnd < schlick` decision only when the
// 24-bit draw lands between the two values (probability <= 2^-24 per affected call).
__device__ __forceinline__ float pow5(float x) {
    const double d = (double)x;
    const double d2 = d * d;
    return (float)(d2 * d2 * d);
}
__device__ __forceinline__ float schlick(float cosine, float ref_idx) {     // material.h:9-13
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5(1.0f - cosine);
}
__device__ __forceinline__ f3 refract(f3 uv, f3 n, float etai_over_etat) { // material.h:15-21
    const float cos_theta = fminf(dot(-uv, n), 1.0f);
    const f3 r_out_parallel = etai_over_etat * (uv + cos_theta * n);
    const float sl = sqlen(r_out_parallel);
    const f3 r_out_perp = sl >= 1.0f ? F3(0, 0, 0) : (-__fsqrt_rn(1.0f - sl)) * n;
    return r_out_parallel + r_out_perp;
}
__device__ __forceinline__ f3 reflect(f3 v, f3 n) {                        // material.h:23-25
    return v - (2.0f * dot(v, n)) * n;
}

struct Scatter {        // scatter_info, helper_structs.h:38-46
    f3 wi;
    f3 throughput;
    float t;
    bool specular;
    bool refracted;
};

// material_scatter, scene_materials.h:13-20, with the three BSDFs of material.h:27-31,46-53,55-60,73-92.
// `normal` faces the ray; `inside` is the path's inside flag; `wo` the un-renormalised path direction.
__device__ __forceinline__ void material_scatter(Scatter& out, float hit_t, f3 normal, bool inside, f3 wo,
                                                 int type, f3 color, float param, uint32_t& rng) {
    out.specular = false;
    out.throughput = F3(1.0f, 1.0f, 1.0f);
    out.refracted = false;
    out.t = hit_t;
    if (type == RT_DIFFUSE) {
        out.wi = unit(normal + random_in_unit_sphere(rng));
        out.throughput = color;
    } else if (type == RT_METAL) {
        f3 reflected = reflect(wo, normal);
        if (param > 0.0001f) reflected = reflected + param * random_in_unit_sphere(rng);
        out.wi = unit(reflected);
        out.throughput = out.throughput * color;
        out.specular = true;
    } else {
        // dielectric_bsdf(ior = param, tint = color, fuzz 0, absorption 0)
        if (inside) {
            // exp(-absorption * t) with absorption == 0: expf(-0.0f * t) == 1 for every finite t (material.h:77)
            out.throughput = F3(1.0f, 1.0f, 1.0f);
        }
        const float etai_over_etat = inside ? param : (1.0f / param);
        const float cos_theta = fminf(dot(-wo, normal), 1.0f);
        const float sin_theta = __fsqrt_rn(1.0f - cos_theta * cos_theta);
        bool reflect_it = etai_over_etat * sin_theta > 1.0f;
        if (!reflect_it) reflect_it = rnd(rng) < schlick(cos_theta, etai_over_etat);
        if (reflect_it) {
            out.wi = unit(reflect(wo, normal));      // glossy_bsdf with fuzz 0
            out.throughput = out.throughput * color;
        } else {
            out.wi = unit(refract(wo, normal, etai_over_etat));
            out.refracted = true;
        }
        out.specular = true;
    }
}

// Sky, kernels.cu:419-421 (gradient) / :424 (constant grey)
__device__ __forceinline__ f3 sky_color(int sky_mode, f3 rayDir) {
    if (sky_mode == RT_SKY_GRADIENT) {
        const float t = 0.5f * (rayDir.y + 1.0f);
        return (1.0f - t) * F3(1.0f, 1.0f, 1.0f) + t * F3(0.5f, 0.7f, 1.0f);
    }
    return F3(0.5f, 0.5f, 0.5f);
}

}  // namespace rtd
