// rt_device.h — device-side fp32 vector math, RNG and BSDFs for the gfx950 render kernels.
//
// Operation order is part of the contract: in the PARITY build (-ffp-contract=off) every
// expression below rounds exactly like the reference's header-only code, so the HIP path is
// bit-identical to the CPU oracle.  Each helper cites the reference lines it implements
// (relative to /root/reference/).  The FAST build compiles the same source with FMA
// contraction enabled.
#pragma once

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "../../include/rt_types.h"

#define RT_SINCOS_FN __device__ __forceinline__
#include "rt_glibc_sincosf.h"      // glibc's sinf / cosf / sincosf restated (fp64 polynomial, the FMA build's fusions): the CPU side's bits
// glibc's powf for the exponent 5 restated the same way (its log2 / exp2 tables in constant memory).  A real function: it runs only when the cheap
// bracket of schlick_above cannot decide (about once per 10^6 calls), and its fp64 temporaries stay out of the shading code's register budget
#define RT_POWF_FN __device__ __attribute__((noinline))
#define RT_POWF_TABLE static __constant__ const
#include "rt_glibc_powf.h"

namespace rtd {

// sinf / cosf as the CPU oracle's libm computes them (|y| < 120: bit-identical, pinned exhaustively on the host); OCML beyond that range
__device__ __forceinline__ void rt_sincosf(float y, float& s, float& c) {
    if (!rt_glibc_sincosf(y, &s, &c)) sincosf(y, &s, &c);
}
__device__ __forceinline__ float rt_sinf(float y) { float s, c; rt_sincosf(y, s, c); return s; }
// checker_layer's sin x * sin y * sin z (material.h:33-36): a dormant look preset.  A real function, not inlined: three inlined copies of the fp64
// polynomial inside the shading code cost every kernel two VGPRs for a hoisted constant and a spill, whether a scene uses the preset or not.
__device__ __attribute__((noinline)) float rt_checker_sines(float x, float y, float z) { return rt_sinf(x) * rt_sinf(y) * rt_sinf(z); }

struct f3 { float x, y, z; };

// IEEE-754 correctly rounded square root (llvm.sqrt.f32 under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt).  NOT __fsqrt_rn: in this ROCm that intrinsic is
// __ocml_native_sqrt_f32 (v_sqrt_f32, 1 ulp) and breaks bit-parity with the CPU.
__device__ __forceinline__ float rt_sqrt(float x) { return __builtin_sqrtf(x); }

__device__ __forceinline__ f3 F3(float x, float y, float z) { return { x, y, z }; }
__device__ __forceinline__ f3 ld3(const rt_vec3& v) { return { v.e[0], v.e[1], v.e[2] }; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }   // vec3.h:59
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }   // vec3.h:63
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }   // vec3.h:67
__device__ __forceinline__ f3 operator*(float t, f3 a) { return { t * a.x, t * a.y, t * a.z }; }       // vec3.h:75,83
// ---- IEEE division and square root through fp64 (rt_div64.h: the argument and the host twin) ------------------------------------------------
// vec3 / float (vec3.h:79) and unit_vector (vec3.h:194): the same bits as the plain operators for ~16 / ~28 instead of ~33 / ~47 instructions, three call
// sites per ray.  RT_DIV64=0 (A/B) and the FAST build keep the compiler's expansion.
#ifndef RT_DIV64
#define RT_DIV64 0          // measured on the GPU: bit-exact (162 tests) and SLOWER than the compiler's fp32 expansion - C2 -7 %, C5 -3 %, C4 -4 % (profiles/r04_ab_div64_*.txt):
                            // the fp64 chain is shorter in instructions and longer in cycles.  Kept as a tested alternative.
#endif
#ifdef RT_MODE_FAST
#undef RT_DIV64
#define RT_DIV64 0
#endif
#define RT_DIV64_FN __device__ __forceinline__
#include "rt_div64.h"
// (the plain operators: one real function each - they run for operands outside the fast path's range, never on the render path's unit vectors and radii -
// so that the compiler's division / square-root expansions exist once per kernel instead of once per call site)
__device__ __attribute__((noinline)) f3 rt_div3_plain(f3 a, float t) { return { a.x / t, a.y / t, a.z / t }; }
__device__ __forceinline__ f3 operator/(f3 a, float t) {                                               // vec3.h:79
#if RT_DIV64
    const float at = fabsf(t);
    if (at >= 0x1p-60f && at <= 0x1p60f) {
        float q[3];
        if (rt_div3_64(a.x, a.y, a.z, rt_recip64((double)t, (double)__builtin_amdgcn_rcpf(t)), q)) return { q[0], q[1], q[2] };
    }
    return rt_div3_plain(a, t);
#else
    return { a.x / t, a.y / t, a.z / t };
#endif
}
__device__ __forceinline__ f3 operator-(f3 a) { return { -a.x, -a.y, -a.z }; }                         // vec3.h:23
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }         // vec3.h:87
__device__ __forceinline__ f3 cross(f3 a, f3 b) {                                                      // vec3.h:91
    return { (a.y * b.z - a.z * b.y), (-(a.x * b.z - a.z * b.x)), (a.x * b.y - a.y * b.x) };
}
__device__ __forceinline__ float sqlen(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }             // vec3.h:36
__device__ __forceinline__ float len(f3 a) { return rt_sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }   // vec3.h:35
__device__ __forceinline__ f3 unit(f3 a) {                                                             // vec3.h:194: a / sqrt(squared_length)
#if RT_DIV64
    const float s = a.x * a.x + a.y * a.y + a.z * a.z;                                                 // vec3.h:35-36, the reference's roundings
    if (s >= 0x1p-100f && s <= 0x1p100f) {
        double h;
        const float l = rt_sqrt64(s, (double)__builtin_amdgcn_rsqf(s), &h);                            // == __builtin_sqrtf(s)
        float q[3];
        if (rt_div3_64(a.x, a.y, a.z, rt_recip64((double)l, h + h), q)) return { q[0], q[1], q[2] };   // (l differs from sqrt(s) by <= 2^-25: a fine seed)
        return rt_div3_plain(a, l);
    }
    return rt_div3_plain(a, rt_sqrt(s));
#else
    return a / len(a);
#endif
}
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
// 2.0f * rnd(state) - 1.0f, the bits the two roundings of rnd.h:23,45 produce, in one: rnd is n * 2^-24 with n < 2^24 (the conversion and the scaling are
// exact), 2.0f * that is exact as well, so the subtraction is the only rounding - RN(n * 2^-23 - 1), which is what the fused multiply-add returns.
// (The rejection loops below run max-over-lanes times per wave: this is 6 of their 38 instructions.)
__device__ __forceinline__ float rnd_pm1(uint32_t& state) {
    uint32_t x = state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 15;
    state = x;
    return __builtin_fmaf((float)(x & 0xFFFFFF), 1.1920928955078125e-7f, -1.0f);
}
__device__ __forceinline__ f3 random_in_unit_disk(uint32_t& state) {    // rnd.h:20-26, draws x then y
    f3 p;
    do {
        const float px = rnd_pm1(state);
        const float py = rnd_pm1(state);
        p = F3(px, py, 0.0f);                                            // (2.0f * 0.0f - 0.0f)
    } while (dot(p, p) >= 1.0f);
    return p;
}
__device__ __forceinline__ f3 random_in_unit_sphere(uint32_t& state) {  // rnd.h:41-49, draws x,y,z
    f3 p;
    do {
        const float px = rnd_pm1(state);
        const float py = rnd_pm1(state);
        const float pz = rnd_pm1(state);
        p = F3(px, py, pz);
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

// pow(x, 5.0f), material.h:12: the reference calls powf, which on the CPU side is glibc's; rt_glibc_powf.h restates that algorithm (equal to libm on
// every float, tests/test_oracle_golden.py::test_glibc_powf5_twin_is_libm), so schlick is bit-exact like everything else on the path.
__device__ __forceinline__ float pow5(float x) { return rt_glibc_powf5(x); }
__device__ __forceinline__ float schlick(float cosine, float ref_idx) {     // material.h:9-13
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5(1.0f - cosine);
}
// `u < schlick(cosine, ref_idx)` - the only use the path makes of schlick (material.h:58: rnd(...) < schlick(...)) - decided without running powf almost always:
// the fp64 product x^5 rounded once lies within kPow5Bracket ulps of glibc's powf(x, 5) for every float x in [0, 2.5] (held on ALL of them by
// tests/test_oracle_golden.py::test_glibc_powf5_twin_is_libm: the largest distance is 1 ulp), and p -> fl(r0 + fl(k p)) is monotone (each rounding is), so
// the reference's schlick lies between the values at the bracket's two ends; when `u` is on the same side of both, that is the answer.  Otherwise - u within
// a few ulps of the threshold: u is a multiple of 2^-24, about one call in a million - the exact function runs.  Same bits as the plain comparison.
constexpr uint32_t kPow5Bracket = 2u;
__device__ __forceinline__ bool schlick_above(float u, float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    const float k = 1.0f - r0, x = 1.0f - cosine;
    const double d = (double)x, d2 = d * d;
    const uint32_t pb = __float_as_uint((float)(d2 * d2 * d));
    const float s_a = r0 + k * __uint_as_float(pb > kPow5Bracket ? pb - kPow5Bracket : 0u);
    const float s_b = r0 + k * __uint_as_float(pb + kPow5Bracket);
    const bool a = u < s_a, b = u < s_b;
    if (x >= 0.0f && x <= 2.5f && a == b) return a;
    return u < schlick(cosine, ref_idx);
}
__device__ __forceinline__ f3 refract(f3 uv, f3 n, float etai_over_etat) { // material.h:15-21
    const float cos_theta = fminf(dot(-uv, n), 1.0f);
    const f3 r_out_parallel = etai_over_etat * (uv + cos_theta * n);
    const float sl = sqlen(r_out_parallel);
    const f3 r_out_perp = sl >= 1.0f ? F3(0, 0, 0) : (-rt_sqrt(1.0f - sl)) * n;
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
__device__ __forceinline__ f3 hex_color(int hexValue) {                    // scene_materials.h:6-11
    const float r = (float)((hexValue >> 16) & 0xFF);
    const float g = (float)((hexValue >> 8) & 0xFF);
    const float b = (float)((hexValue) & 0xFF);
    return F3(r, g, b) / 255.0f;
}

// fresnel_layer, material.h:55-60: true = the glossy (reflecting) layer is chosen; draws at most one random number
__device__ __forceinline__ bool fresnel_layer(f3 normal, bool inside, f3 wo, float ior, uint32_t& rng) {
    const float etai_over_etat = inside ? ior : (1.0f / ior);
    const float cos_theta = fminf(dot(-wo, normal), 1.0f);
    const float sin_theta = rt_sqrt(1.0f - cos_theta * cos_theta);
    bool r = etai_over_etat * sin_theta > 1.0f;
    if (!r) { const float u = rnd(rng); r = schlick_above(u, cos_theta, etai_over_etat); }
    return r;
}

// material_scatter, scene_materials.h:13-20, with the BSDFs of material.h.  `normal` faces the ray; `inside` is the
// path's inside flag; `wo` the un-renormalised path direction; `hp` the hit point (only the checker preset reads it).
// type >= RT_FLOOR_COAT selects one of the reference's dormant look presets (scene_materials.h:22-93; additive).
// BASIC (the sphere kernel's instantiation for scenes whose materials are all RT_DIFFUSE / RT_METAL / RT_GLASS - every scene of the README-era renderer): the
// same three BSDFs without the presets' parameter tables, layers, absorption and subsurface events around them - the same operations on the same operands,
// so the same bits (1.0f * tint is tint), in a third of the code and registers.
template <bool BASIC = false>
__device__ __forceinline__ void material_scatter(Scatter& out, float hit_t, f3 hp, f3 normal, bool inside, f3 wo,
                                                 int type, f3 color, float param, uint32_t& rng) {
    if (BASIC) {
        const bool diffuse = type == RT_DIFFUSE, metal = type == RT_METAL;
        bool fresnel = false;
        if (!diffuse && !metal) fresnel = fresnel_layer(normal, inside, wo, param, rng);        // dielectric_bsdf, material.h:73-92
        f3 rs = F3(0, 0, 0);
        if (diffuse || (metal && param > 0.0001f)) rs = random_in_unit_sphere(rng);
        f3 v, thr = color;
        bool refracted = false;
        if (diffuse) {                                           // diffuse_bsdf, material.h:27-31
            v = normal + rs;
        } else if (metal) {                                      // glossy_bsdf, material.h:46-53
            v = reflect(wo, normal);
            if (param > 0.0001f) v = v + param * rs;
        } else if (fresnel) {
            v = reflect(wo, normal);                             // glossy_bsdf, fuzz 0; throughput (1, 1, 1) * tint
        } else {
            v = refract(wo, normal, inside ? param : (1.0f / param));
            refracted = true;
            thr = F3(1.0f, 1.0f, 1.0f);
        }
        out.wi = unit(v);
        out.throughput = thr;
        out.specular = !diffuse;
        out.refracted = refracted;
        out.t = hit_t;
        return;
    }
    // Every branch is reduced to: an un-normalised outgoing direction v (normalised once at the end, except for the
    // subsurface scattering event which the reference leaves un-normalised, material.h:128), a throughput, flags and t.
    // (Written per branch as `out.throughput = out.throughput * color`, hipcc 7.2 dropped the x component of the
    // product in the glass-reflect branch; tests/test_gpu_probe_functions.py::test_material_scatter pins this.)
    bool specular = true, refracted = false, normalise = true;
    f3 v;
    f3 thr = F3(1.0f, 1.0f, 1.0f);
    float t_out = hit_t;

    // ---- which BSDF, with which parameters
    enum { B_DIFFUSE, B_GLOSSY, B_COAT, B_DIELECTRIC, B_SSS } bsdf;
    f3 albedo = color;                  // diffuse albedo
    f3 tint = color;                    // glossy / dielectric tint
    float ior = param, fuzz = 0.0f;
    f3 absorption = F3(0, 0, 0);
    const f3 model_base = F3(0.0972942f, 0.0482054f, 0.000273194f);
    switch (type) {
    case RT_DIFFUSE: bsdf = B_DIFFUSE; break;
    case RT_METAL:   bsdf = B_GLOSSY; fuzz = param; break;
    case RT_GLASS:   bsdf = B_DIELECTRIC; break;                                 // tint = color, absorption 0
    case RT_FLOOR_COAT:    bsdf = B_COAT; ior = 1.5f; albedo = hex_color(0x511845); tint = F3(1, 1, 1); break;
    case RT_FLOOR_DIFFUSE: bsdf = B_DIFFUSE; albedo = hex_color(0x511845); break;
    case RT_FLOOR_CHECKER: {                                                     // checker_layer, material.h:33-36
        const float sines = rt_checker_sines(0.2f * hp.x, 0.2f * hp.y, 0.2f * hp.z);
        bsdf = B_DIFFUSE; albedo = (sines < 0) ? hex_color(0x511845) : hex_color(0xff5733);
        break;
    }
    case RT_MODEL_COAT:    bsdf = B_COAT; ior = 1.1f; albedo = model_base; tint = F3(1, 1, 1); break;
    case RT_MODEL_DIFFUSE: bsdf = B_DIFFUSE; albedo = model_base; break;
    case RT_MODEL_GLOSSY:  bsdf = B_GLOSSY; tint = F3(1, 1, 1); break;
    case RT_MODEL_GLASS:   bsdf = B_DIELECTRIC; ior = 1.1f; tint = F3(1, 1, 1); break;
    case RT_MODEL_TINTEDGLASS:
        bsdf = B_DIELECTRIC; ior = 1.1f; tint = F3(1, 1, 1);
        absorption = (-F3(logf(model_base.x), logf(model_base.y), logf(model_base.z))) / 10.0f;
        break;
    default:               bsdf = B_SSS; ior = 1.333f; tint = F3(1, 1, 1); absorption = F3(0.9f, 0.3f, 0.02f); break;
    }

    // Lanes of different materials reach their random_in_unit_sphere draw (diffuse bounce, metal fuzz, subsurface event) in
    // different branches; a wave would run the rejection loop once per branch (max-over-lanes iterations each time).  The
    // draw is hoisted to ONE place: everything a lane draws BEFORE it (coat layer, subsurface distance, and the Fresnel choice below) is
    // done first - every lane's own draw order is the reference's.
    bool scattered = false;
    if (bsdf == B_DIELECTRIC || bsdf == B_SSS) {             // dielectric_bsdf (material.h:73-92) / subsurface_dielectric_bsdf (:119-143)
        if (inside) {
            if (bsdf == B_SSS) {
                const float d = -logf(rnd(rng)) / 2.0f;      // scatteringDistance 2.0 (scene_materials.h:91)
                if (d < hit_t) { scattered = true; t_out = d; }
            }
            // exp(-absorption * t) (material.h:77,126); with absorption == 0 that is expf(-0.0f * t) == 1 exactly
            if (absorption.x != 0.0f || absorption.y != 0.0f || absorption.z != 0.0f) {
                const f3 e = t_out * (-absorption);
                thr = F3(expf(e.x), expf(e.y), expf(e.z));
            }
        }
    }
    // The Fresnel choice (fresnel_layer: at most one draw, and powf) of the coat presets (material.h:62-70, before the layer's own scatter) and of the
    // dielectrics (material.h:80,131, the path's last draw: a lane that takes it never draws a unit-sphere point) at ONE site: a lane calls it at most once.
    bool fresnel = false;
    if (bsdf == B_COAT || ((bsdf == B_DIELECTRIC || bsdf == B_SSS) && !scattered)) fresnel = fresnel_layer(normal, inside, wo, ior, rng);
    if (bsdf == B_COAT) bsdf = fresnel ? B_GLOSSY : B_DIFFUSE;
    f3 rs = F3(0, 0, 0);
    if (bsdf == B_DIFFUSE || (bsdf == B_GLOSSY && fuzz > 0.0001f) || scattered) rs = random_in_unit_sphere(rng);

    if (bsdf == B_DIFFUSE) {                                 // diffuse_bsdf, material.h:27-31
        v = normal + rs;
        thr = albedo;
        specular = false;
    } else if (bsdf == B_GLOSSY) {                           // glossy_bsdf, material.h:46-53 (throughput 1 * tint == tint)
        v = reflect(wo, normal);
        if (fuzz > 0.0001f) v = v + fuzz * rs;
        thr = tint;
    } else if (scattered) {
        v = rs;
        normalise = false;                                   // material.h:128: wi is NOT normalised
    } else if (fresnel) {
        v = reflect(wo, normal);                             // glossy_bsdf, fuzz 0
        thr = thr * tint;
    } else {
        v = refract(wo, normal, inside ? ior : (1.0f / ior));
        refracted = true;
    }
    out.wi = normalise ? unit(v) : v;
    out.throughput = thr;
    out.specular = specular;
    out.refracted = refracted;
    out.t = t_out;
}

// Sky, kernels.cu:419-421 (gradient) / :424 (constant grey)
__device__ __forceinline__ f3 sky_color(int sky_mode, f3 rayDir) {
    if (sky_mode == RT_SKY_GRADIENT) {
        const float t = 0.5f * (rayDir.y + 1.0f);
        return (1.0f - t) * F3(1.0f, 1.0f, 1.0f) + t * F3(0.5f, 0.7f, 1.0f);
    }
    return F3(0.5f, 0.5f, 0.5f);
}

// ---- ray.h / intersections.h --------------------------------------------------------------------

struct Ray {
    f3 o, d, inv;       // origin, unit direction, 1/direction
};

__device__ __forceinline__ Ray make_ray(f3 o, f3 dir) {     // ray.h:9 + the hoisted invD of intersections.h:28
    Ray r;
    r.o = o;
    r.d = unit(dir);
    r.inv = F3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    return r;
}

// one axis of the slab test, intersections.h:27-36
__device__ __forceinline__ void slab(float bmin, float bmax, float o, float invD, float& t_min, float& t_max) {
    float t0 = (bmin - o) * invD;
    float t1 = (bmax - o) * invD;
    if (invD < 0.0f) { const float tmp = t0; t0 = t1; t1 = tmp; }
    // `t0 > t_min ? t0 : t_min` / `t1 < t_max ? t1 : t_max` as ONE v_max_f32 / v_min_f32 each (IEEE maxNum/minNum) instead of
    // compare + select: t_min (>= 0.001) and t_max (a hit distance or FLT_MAX) are never NaN, so a NaN t0/t1 (0 * inf on a
    // slab plane) leaves them unchanged in both forms; the only other difference, the sign of a zero t_max, cannot reach a
    // result (t_max is only compared, against values >= 0.001).  Checked bit for bit by the hit_bbox probes.
    t_min = fmaxf(t0, t_min);
    t_max = fminf(t1, t_max);
}

__device__ __forceinline__ float hit_bbox_dist(f3 bmin, f3 bmax, const Ray& r, float t_max) {  // intersections.h:25-41
    float t_min = 0.001f;
    slab(bmin.x, bmax.x, r.o.x, r.inv.x, t_min, t_max);
    slab(bmin.y, bmax.y, r.o.y, r.inv.y, t_min, t_max);
    slab(bmin.z, bmax.z, r.o.z, r.inv.z, t_min, t_max);
    return (t_max < t_min) ? FLT_MAX : t_min;
}

// The same test with the two results apart: `hit` = the slabs overlap, t_entry = the entry distance (hit_bbox_dist returns
// hit ? t_entry : FLT_MAX).  For callers that only compare the result (`< closest`, `right < left`): the compares fold into mask logic.
__device__ __forceinline__ bool hit_bbox_entry(f3 bmin, f3 bmax, const Ray& r, float t_max, float& t_entry) {
    float t_min = 0.001f;
    slab(bmin.x, bmax.x, r.o.x, r.inv.x, t_min, t_max);
    slab(bmin.y, bmax.y, r.o.y, r.inv.y, t_min, t_max);
    slab(bmin.z, bmax.z, r.o.z, r.inv.z, t_min, t_max);
    t_entry = t_min;
    return !(t_max < t_min);
}

// The early-out of the reference matters for ONE thing: a NaN produced on a later axis (0 * inf)
// can only appear after an earlier axis already failed... it cannot un-fail the test, because NaN
// compares false and leaves t_min/t_max unchanged.  So the branch-free form is exact.
__device__ __forceinline__ bool hit_bbox(f3 bmin, f3 bmax, const Ray& r, float t_max) {        // intersections.h:7-23
    return hit_bbox_dist(bmin, bmax, r, t_max) != FLT_MAX;
}

// triangleHit, intersections.h:54-83.  `1.0 / a` there is a double divide narrowed to float, which
// equals the correctly rounded float quotient (53 >= 2*24+2), so 1.0f / a is bit-identical.
// ... on (v0, edge1, edge2): the mesh kernel's compact leaf records hold the two edges, rounded on the host exactly like the two subtractions below
__device__ __forceinline__ float triangle_hit_edges(f3 v0, f3 edge1, f3 edge2, const Ray& r, float t_min, float t_max, float& hitU, float& hitV) {
    const float EPS = 0.0000001f;
    const f3 h = cross(r.d, edge2);
    const float a = dot(edge1, h);
    if (a > -EPS && a < EPS) return FLT_MAX;
    const float f = 1.0f / a;
    const f3 s = r.o - v0;
    const float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return FLT_MAX;
    const f3 q = cross(s, edge1);
    const float v = f * dot(r.d, q);
    if (v < 0.0f || u + v > 1.0f) return FLT_MAX;
    const float t = f * dot(edge2, q);
    if (t > t_min && t < t_max) { hitU = u; hitV = v; return t; }
    return FLT_MAX;
}

__device__ __forceinline__ float triangle_hit(f3 v0, f3 v1, f3 v2, const Ray& r, float t_min, float t_max, float& hitU, float& hitV) {
    const float EPS = 0.0000001f;
    const f3 edge1 = v1 - v0;
    const f3 edge2 = v2 - v0;
    const f3 h = cross(r.d, edge2);
    const float a = dot(edge1, h);
    if (a > -EPS && a < EPS) return FLT_MAX;
    const float f = 1.0f / a;
    const f3 s = r.o - v0;
    const float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return FLT_MAX;
    const f3 q = cross(s, edge1);
    const float v = f * dot(r.d, q);
    if (v < 0.0f || u + v > 1.0f) return FLT_MAX;
    const float t = f * dot(edge2, q);
    if (t > t_min && t < t_max) { hitU = u; hitV = v; return t; }
    return FLT_MAX;
}

__device__ __forceinline__ float sphere_hit(f3 center, float radius, const Ray& r, float t_min, float t_max) {   // intersections.h:85-104
    const f3 oc = r.o - center;
    const float a = dot(r.d, r.d);
    const float b = dot(oc, r.d);
    const float c = dot(oc, oc) - radius * radius;
    const float discriminant = b * b - a * c;
    if (discriminant > 0) {
        const float sq = rt_sqrt(discriminant);
        float temp = (-b - sq) / a;
        if (temp < t_max && temp > t_min) return temp;
        temp = (-b + sq) / a;
        if (temp < t_max && temp > t_min) return temp;
    }
    return FLT_MAX;
}

__device__ __forceinline__ float plane_hit(f3 norm, f3 point, const Ray& r, float t_min, float t_max) {   // intersections.h:43-52
    const float denom = dot(norm, r.d);
    if (denom > -0.000001f) return FLT_MAX;
    const f3 po = point - r.o;
    const float t = dot(po, norm) / denom;
    if (t < t_min || t > t_max) return FLT_MAX;
    return t;
}

// ---- generateShadowRay, kernels.cu:363-393 ------------------------------------------------------------------------
// Samples the spherical light by solid angle from the path's (already advanced) origin.  Returns false BEFORE any draw when
// cosAMax is NaN (origin inside the light's sphere, :371-372), false AFTER exactly two draws when the sampled direction is
// below the surface (:382-383).  `2 * M_PI * x` is a double product narrowed to float at the same two places as the
// reference (:378, :386); `/ M_PI` divides by (float)M_PI (vec3.h:79).  cosf/sinf: glibc's algorithm restated (rt_glibc_sincosf.h) - the
// bits the CPU oracle's libm produces, so this function is bit-exact against the oracle like everything else.
struct ShadowSample {
    f3 dir;             // p.shadowDir
    f3 contrib;         // p.lightContribution
    float dist;         // lightDist
    float cosAMax;
};

__device__ __forceinline__ bool generate_shadow_ray(f3 lightC, float lightR, f3 lightColor, f3 org, f3 atten, f3 normal,
                                                    uint32_t& rng, ShadowSample& out) {
    const f3 sw = unit(lightC - org);
    const f3 su = unit(cross(fabsf(sw.x) > 0.01f ? F3(0, 1, 0) : F3(1, 0, 0), sw));
    const f3 sv = cross(sw, su);
    const float cosAMax = rt_sqrt(1.0f - lightR * lightR / sqlen(org - lightC));
    out.cosAMax = cosAMax;
    if (isnan(cosAMax)) return false;
    const float eps1 = rnd(rng);
    const float eps2 = rnd(rng);
    const float cosA = 1.0f - eps1 + eps1 * cosAMax;
    const float sinA = rt_sqrt(1.0f - cosA * cosA);
    const float phi = (float)(2 * M_PI * (double)eps2);
    float sphi, cphi;
    rt_sincosf(phi, sphi, cphi);
    const f3 l = sinA * (cphi * su) + sinA * (sphi * sv) + cosA * sw;
    const float dotl = dot(l, normal);
    if (!(dotl > 0)) return false;
    out.dir = unit(l);
    const float omega = (float)(2 * M_PI * (double)(1.0f - cosAMax));
    out.contrib = (omega * (dotl * (atten * lightColor))) / (float)M_PI;
    out.dist = len(lightC - org) - lightR;
    return true;
}

}  // namespace rtd
