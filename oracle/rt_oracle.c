/*
 * rt_oracle.c — CPU ORACLE. TEST INFRASTRUCTURE ONLY (see rt_oracle.h).
 *
 * Plain C99, single-threaded.  Build: gcc -O2 -ffp-contract=off (never -ffast-math).
 * Every function cites the reference lines it restates; citations are relative to
 * /root/reference/.  All arithmetic is fp32 unless a reference expression is double
 * (those are kept in double and narrowed at the same place the reference narrows).
 */
#define _GNU_SOURCE 1   /* M_PI under -std=c99 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* ---------------------------------------------------------------- vec3.h ------------------ */

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 ld(const float* p) { return V(p[0], p[1], p[2]); }
static inline v3 ldv(rt_vec3 a) { return V(a.e[0], a.e[1], a.e[2]); }
static inline void st(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }            /* vec3.h:59-61 */
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }            /* vec3.h:63-65 */
static inline v3 mulv(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }           /* vec3.h:67-69 */
static inline v3 muls(float t, v3 a) { return V(t * a.x, t * a.y, t * a.z); }              /* vec3.h:75-77,83-85 */
static inline v3 divs(v3 a, float t) { return V(a.x / t, a.y / t, a.z / t); }              /* vec3.h:79-81 */
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }                                 /* vec3.h:23 */
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          /* vec3.h:87-89 */
static inline v3 cross(v3 a, v3 b) {                                                       /* vec3.h:91-95 */
    return V((a.y * b.z - a.z * b.y), (-(a.x * b.z - a.z * b.x)), (a.x * b.y - a.y * b.x));
}
static inline float sqlen(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }              /* vec3.h:36 */
static inline float len(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }         /* vec3.h:35 */
static inline v3 unit(v3 a) { return divs(a, len(a)); }                                    /* vec3.h:194-196 */
static inline float max3(v3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }                     /* vec3.h:113-115 */

/* ---------------------------------------------------------------- rnd.h ------------------- */

uint32_t orc_xor_shift_32(uint32_t* state) {    /* rnd.h:5-13 */
    uint32_t x = *state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 15;
    *state = x;
    return x;
}

float orc_rnd(uint32_t* state) {                /* rnd.h:15-18 */
    return (float)(orc_xor_shift_32(state) & 0xFFFFFF) / 16777216.0f;
}

uint32_t orc_wang_hash(uint32_t seed) {         /* rnd.h:31-39 */
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}

uint32_t orc_pixel_seed(uint32_t pixel_id) {    /* kernels.cu:541-542 */
    return (orc_wang_hash(pixel_id) * 336343633u) | 1u;
}

/* RT_RNG_COUNTER (additive, not in the reference): one xorshift32 stream per (pixel, sample)
 * instead of per pixel, so samples are independent work items. */
static uint32_t sample_seed(uint32_t pixel_id, uint32_t s) {
    return (orc_wang_hash(pixel_id + orc_wang_hash(s) * 0x9E3779B9u) * 336343633u) | 1u;
}

static uint64_t* g_draws = 0;
static inline float rnd(uint32_t* st_) { if (g_draws) (*g_draws)++; return orc_rnd(st_); }

static v3 random_in_unit_disk(uint32_t* state) {    /* rnd.h:20-26; draws pinned x then y */
    v3 p;
    do {
        float rx = rnd(state);
        float ry = rnd(state);
        p = sub(muls(2.0f, V(rx, ry, 0.0f)), V(1.0f, 1.0f, 0.0f));
    } while (dot(p, p) >= 1.0f);
    return p;
}

static v3 random_in_unit_sphere(uint32_t* state) {  /* rnd.h:41-49; draws pinned x,y,z */
    v3 p;
    do {
        float rx = rnd(state);
        float ry = rnd(state);
        float rz = rnd(state);
        p = sub(muls(2.0f, V(rx, ry, rz)), V(1.0f, 1.0f, 1.0f));
    } while (sqlen(p) >= 1.0f);
    return p;
}

void orc_random_in_unit_disk(uint32_t* state, float out[3]) { st(out, random_in_unit_disk(state)); }
void orc_random_in_unit_sphere(uint32_t* state, float out[3]) { st(out, random_in_unit_sphere(state)); }

/* ---------------------------------------------------------------- ray.h / camera ---------- */

typedef struct { v3 A, B; } ray_t;
static inline ray_t mkray(v3 a, v3 b) { ray_t r; r.A = a; r.B = unit(b); return r; }       /* ray.h:9 */
static inline v3 point_at(const ray_t* r, float t) { return add(r->A, muls(t, r->B)); }     /* ray.h:12 */

void orc_make_camera(const float lookfrom_[3], const float lookat_[3], const float vup_[3], float vfov,
                     float aspect, float aperture, float focus_dist, rt_camera* out) {
    /* helper_structs.h:194-207 */
    v3 lookfrom = ld(lookfrom_), lookat = ld(lookat_), vup = ld(vup_);
    float lens_radius = aperture / 2.0f;
    float theta = vfov * ((float)M_PI) / 180.0f;
    float half_height = tanf(theta / 2.0f);
    float half_width = aspect * half_height;
    v3 origin = lookfrom;
    v3 w = unit(sub(lookfrom, lookat));
    v3 u = unit(cross(vup, w));
    v3 v = cross(w, u);
    /* origin - half_width*focus_dist*u - half_height*focus_dist*v - focus_dist*w  (left to right) */
    v3 llc = sub(sub(sub(origin, muls(half_width * focus_dist, u)), muls(half_height * focus_dist, v)), muls(focus_dist, w));
    v3 horizontal = muls(2.0f * half_width * focus_dist, u);
    v3 vertical = muls(2.0f * half_height * focus_dist, v);
    st(out->origin.e, origin);
    st(out->lower_left_corner.e, llc);
    st(out->horizontal.e, horizontal);
    st(out->vertical.e, vertical);
    st(out->u.e, u); st(out->v.e, v); st(out->w.e, w);
    out->lens_radius = lens_radius;
}

static ray_t get_ray(const rt_camera* c, float s, float t, uint32_t* state) {   /* camera.h:8-12 */
    v3 rd = muls(c->lens_radius, random_in_unit_disk(state));
    v3 offset = add(muls(rd.x, ldv(c->u)), muls(rd.y, ldv(c->v)));
    v3 o = add(ldv(c->origin), offset);
    v3 d = sub(sub(add(add(ldv(c->lower_left_corner), muls(s, ldv(c->horizontal))), muls(t, ldv(c->vertical))),
                   ldv(c->origin)), offset);
    return mkray(o, d);
}

void orc_get_ray(const rt_camera* c, float s, float t, uint32_t* state, float org[3], float dir[3]) {
    ray_t r = get_ray(c, s, t, state);
    st(org, r.A); st(dir, r.B);
}

/* ---------------------------------------------------------------- intersections.h --------- */

static inline float comp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

static int hit_bbox(v3 bmin, v3 bmax, const ray_t* r, float t_max) {            /* intersections.h:7-23 */
    float t_min = 0.001f;
    for (int a = 0; a < 3; a++) {
        float invD = 1.0f / comp(r->B, a);
        float t0 = (comp(bmin, a) - comp(r->A, a)) * invD;
        float t1 = (comp(bmax, a) - comp(r->A, a)) * invD;
        if (invD < 0.0f) { float tmp = t0; t0 = t1; t1 = tmp; }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max < t_min) return 0;
    }
    return 1;
}

static float hit_bbox_dist(v3 bmin, v3 bmax, const ray_t* r, float t_max) {     /* intersections.h:25-41 */
    float t_min = 0.001f;
    for (int a = 0; a < 3; a++) {
        float invD = 1.0f / comp(r->B, a);
        float t0 = (comp(bmin, a) - comp(r->A, a)) * invD;
        float t1 = (comp(bmax, a) - comp(r->A, a)) * invD;
        if (invD < 0.0f) { float tmp = t0; t0 = t1; t1 = tmp; }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max < t_min) return FLT_MAX;
    }
    return t_min;
}

static float plane_hit(const rt_plane* p, const ray_t* r, float t_min, float t_max) {   /* intersections.h:43-52 */
    float denom = dot(ldv(p->norm), r->B);
    if (denom > -0.000001f) return FLT_MAX;
    v3 po = sub(ldv(p->point), r->A);
    float t = dot(po, ldv(p->norm)) / denom;
    if (t < t_min || t > t_max) return FLT_MAX;
    return t;
}

static float triangle_hit(const rt_triangle* tri, const ray_t* r, float t_min, float t_max, float* hitU, float* hitV) {
    /* intersections.h:54-83 */
    const float EPS = 0.0000001;                 /* double literal narrowed to float, :55 */
    v3 v0 = ldv(tri->v[0]);
    v3 edge1 = sub(ldv(tri->v[1]), v0);
    v3 edge2 = sub(ldv(tri->v[2]), v0);
    v3 h = cross(r->B, edge2);
    float a = dot(edge1, h);
    if (a > -EPS && a < EPS) return FLT_MAX;
    float f = (float)(1.0 / (double)a);          /* :64 double divide, narrowed */
    v3 s = sub(r->A, v0);
    float u = f * dot(s, h);
    if ((double)u < 0.0 || (double)u > 1.0) return FLT_MAX;
    v3 q = cross(s, edge1);
    float v = f * dot(r->B, q);
    if ((double)v < 0.0 || (double)(u + v) > 1.0) return FLT_MAX;
    float t = f * dot(edge2, q);
    if (t > t_min && t < t_max) { *hitU = u; *hitV = v; return t; }
    return FLT_MAX;
}

static float sphere_hit(const rt_sphere* s, const ray_t* r, float t_min, float t_max) {  /* intersections.h:85-104 */
    v3 oc = sub(r->A, ldv(s->center));
    float a = dot(r->B, r->B);
    float b = dot(oc, r->B);
    float c = dot(oc, oc) - s->radius * s->radius;
    float discriminant = b * b - a * c;
    if (discriminant > 0) {
        float temp = (-b - sqrtf(discriminant)) / a;
        if (temp < t_max && temp > t_min) return temp;
        temp = (-b + sqrtf(discriminant)) / a;
        if (temp < t_max && temp > t_min) return temp;
    }
    return FLT_MAX;
}

float orc_sphere_hit(const rt_sphere* s, const float org[3], const float dir_in[3], float t_min, float t_max) {
    ray_t r = mkray(ld(org), ld(dir_in));
    return sphere_hit(s, &r, t_min, t_max);
}
float orc_triangle_hit(const rt_triangle* tri, const float org[3], const float dir_in[3], float t_min, float t_max,
                       float* hitU, float* hitV) {
    ray_t r = mkray(ld(org), ld(dir_in));
    return triangle_hit(tri, &r, t_min, t_max, hitU, hitV);
}
int orc_hit_bbox(const float bmin[3], const float bmax[3], const float org[3], const float dir_in[3], float t_max) {
    ray_t r = mkray(ld(org), ld(dir_in));
    return hit_bbox(ld(bmin), ld(bmax), &r, t_max);
}
float orc_hit_bbox_dist(const float bmin[3], const float bmax[3], const float org[3], const float dir_in[3], float t_max) {
    ray_t r = mkray(ld(org), ld(dir_in));
    return hit_bbox_dist(ld(bmin), ld(bmax), &r, t_max);
}
float orc_plane_hit(const rt_plane* p, const float org[3], const float dir_in[3], float t_min, float t_max) {
    ray_t r = mkray(ld(org), ld(dir_in));
    return plane_hit(p, &r, t_min, t_max);
}

/* ---------------------------------------------------------------- material.h -------------- */

typedef struct {            /* intersection, helper_structs.h:16-36 (the members the path uses) */
    uint32_t objId;
    unsigned char meshID;
    float t;
    v3 p;
    v3 normal;
    int inside;
    float texCoords[2];
} inters_t;

typedef struct { v3 wi; int specular; v3 throughput; int refracted; float t; } scat_t;   /* helper_structs.h:38-46 */

float orc_schlick(float cosine, float ref_idx) {        /* material.h:9-13 */
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * powf((1.0f - cosine), 5.0f);
}

static v3 refract_(v3 uv, v3 n, float etai_over_etat) { /* material.h:15-21 */
    float cos_theta = fminf(dot(neg(uv), n), 1.0f);
    v3 r_out_parallel = muls(etai_over_etat, add(uv, muls(cos_theta, n)));
    float sl = sqlen(r_out_parallel);
    v3 r_out_perp = sl >= 1.0f ? V(0, 0, 0) : muls(-sqrtf(1.0f - sl), n);
    return add(r_out_parallel, r_out_perp);
}

static v3 reflect_(v3 v, v3 n) {                        /* material.h:23-25 */
    return sub(v, muls(2.0f * dot(v, n), n));
}

void orc_reflect(const float v[3], const float n[3], float out[3]) { st(out, reflect_(ld(v), ld(n))); }
void orc_refract(const float uv[3], const float n[3], float e, float out[3]) { st(out, refract_(ld(uv), ld(n), e)); }

static void diffuse_bsdf(scat_t* out, const inters_t* i, v3 albedo, uint32_t* rng) {    /* material.h:27-31 */
    out->wi = unit(add(i->normal, random_in_unit_sphere(rng)));
    out->throughput = albedo;
    out->specular = 0;
}

static void glossy_bsdf(scat_t* out, const inters_t* i, v3 wo, v3 tint, float fuzz, uint32_t* rng) { /* material.h:46-53 */
    v3 reflected = reflect_(wo, i->normal);
    if (fuzz > 0.0001f)
        reflected = add(reflected, muls(fuzz, random_in_unit_sphere(rng)));
    out->wi = unit(reflected);
    out->throughput = mulv(out->throughput, tint);
    out->specular = 1;
}

static int fresnel_layer(const inters_t* i, v3 wo, float ior, uint32_t* rng) {           /* material.h:55-60 */
    float etai_over_etat = i->inside ? ior : (1.0f / ior);
    float cos_theta = fminf(dot(neg(wo), i->normal), 1.0f);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    return (etai_over_etat * sin_theta > 1.0f || rnd(rng) < orc_schlick(cos_theta, etai_over_etat));
}

static void dielectric_bsdf(scat_t* out, const inters_t* i, v3 wo, float layer_ior, v3 glossy_tint, float glossy_fuzz,
                            v3 absorption, uint32_t* rng) {                              /* material.h:73-92 */
    if (i->inside) {
        v3 e = muls(i->t, neg(absorption));          /* -absorptionCoefficient * i.t */
        out->throughput = V(expf(e.x), expf(e.y), expf(e.z));
    }
    if (fresnel_layer(i, wo, layer_ior, rng)) {
        glossy_bsdf(out, i, wo, glossy_tint, glossy_fuzz, rng);
    } else {
        float etai_over_etat = i->inside ? layer_ior : (1.0f / layer_ior);
        out->wi = unit(refract_(wo, i->normal, etai_over_etat));
        out->refracted = 1;
    }
    out->specular = 1;
}

static int checker_layer(const inters_t* i, float frequency) {     /* material.h:33-36 */
    float sines = sinf(frequency * i->p.x) * sinf(frequency * i->p.y) * sinf(frequency * i->p.z);
    return sines < 0;
}

static void coat_bsdf(scat_t* out, const inters_t* i, v3 wo, float layer_ior, v3 glossy_tint, float glossy_fuzz, v3 diffuse_albedo,
                      uint32_t* rng) {                                /* material.h:62-70 */
    if (fresnel_layer(i, wo, layer_ior, rng)) glossy_bsdf(out, i, wo, glossy_tint, glossy_fuzz, rng);
    else diffuse_bsdf(out, i, diffuse_albedo, rng);
}

static void subsurface_dielectric_bsdf(scat_t* out, const inters_t* i, v3 wo, float layer_ior, v3 glossy_tint, float glossy_fuzz,
                                       v3 absorption, float scatteringDistance, uint32_t* rng) {    /* material.h:119-143 */
    int scattered = 0;
    if (i->inside) {
        float d = -logf(rnd(rng)) / scatteringDistance;
        if (d < i->t) { scattered = 1; out->t = d; }
        v3 e = muls(out->t, neg(absorption));
        out->throughput = V(expf(e.x), expf(e.y), expf(e.z));
    }
    if (scattered) {
        out->wi = random_in_unit_sphere(rng);            /* not normalised in the reference (:128) */
    } else {
        if (fresnel_layer(i, wo, layer_ior, rng)) {
            glossy_bsdf(out, i, wo, glossy_tint, glossy_fuzz, rng);
        } else {
            float etai_over_etat = i->inside ? layer_ior : (1.0f / layer_ior);
            out->wi = unit(refract_(wo, i->normal, etai_over_etat));
            out->refracted = 1;
        }
    }
    out->specular = 1;
}

static v3 hex_color(int hexValue) {                     /* scene_materials.h:6-11 */
    float r = (float)((hexValue >> 16) & 0xFF);
    float g = (float)((hexValue >> 8) & 0xFF);
    float b = (float)((hexValue) & 0xFF);
    return divs(V(r, g, b), 255.0f);
}

/* the dormant look presets, scene_materials.h:22-93 */
static void preset_scatter(scat_t* out, const inters_t* i, v3 wo, int type, uint32_t* rng) {
    const v3 one = V(1, 1, 1);
    const v3 model_base = V(0.0972942f, 0.0482054f, 0.000273194f);
    switch (type) {
    case RT_FLOOR_COAT:    coat_bsdf(out, i, wo, 1.5f, one, 0.0f, hex_color(0x511845), rng); break;
    case RT_FLOOR_DIFFUSE: diffuse_bsdf(out, i, hex_color(0x511845), rng); break;
    case RT_FLOOR_CHECKER:
        if (checker_layer(i, 0.2f)) diffuse_bsdf(out, i, hex_color(0x511845), rng);
        else diffuse_bsdf(out, i, hex_color(0xff5733), rng);
        break;
    case RT_MODEL_COAT:    coat_bsdf(out, i, wo, 1.1f, one, 0.0f, model_base, rng); break;
    case RT_MODEL_DIFFUSE: diffuse_bsdf(out, i, model_base, rng); break;
    case RT_MODEL_GLOSSY:  glossy_bsdf(out, i, wo, one, 0.0f, rng); break;
    case RT_MODEL_GLASS:   dielectric_bsdf(out, i, wo, 1.1f, one, 0.0f, V(0, 0, 0), rng); break;
    case RT_MODEL_TINTEDGLASS: {
        const float absorptionDistance = 10;
        const v3 absorption = divs(neg(V(logf(model_base.x), logf(model_base.y), logf(model_base.z))), absorptionDistance);
        dielectric_bsdf(out, i, wo, 1.1f, one, 0.0f, absorption, rng);
        break;
    }
    default:               subsurface_dielectric_bsdf(out, i, wo, 1.333f, one, 0.0f, V(0.9f, 0.3f, 0.02f), 2.0f, rng); break;
    }
}

static void material_scatter(scat_t* out, const inters_t* i, v3 wo, const rt_material* mat, v3 color, uint32_t* rng) {
    /* scene_materials.h:13-20 */
    if (mat->type >= RT_FLOOR_COAT) { preset_scatter(out, i, wo, mat->type, rng); return; }
    if (mat->type == RT_DIFFUSE)
        diffuse_bsdf(out, i, color, rng);
    else if (mat->type == RT_METAL)
        glossy_bsdf(out, i, wo, color, mat->param, rng);
    else
        dielectric_bsdf(out, i, wo, mat->param, color, 0.0f, V(0, 0, 0), rng);
}

static inline scat_t scat_init(const inters_t* i) {     /* helper_structs.h:45 */
    scat_t s;
    s.wi = V(0, 0, 0);
    s.specular = 0; s.throughput = V(1, 1, 1); s.refracted = 0; s.t = i->t;
    return s;
}

void orc_material_scatter_p(float inters_t_, const float p[3], const float normal[3], int inside, const float wo[3],
                            const rt_material* mat, const float color[3], uint32_t* rng, orc_scatter* out) {
    inters_t i; memset(&i, 0, sizeof i);
    i.t = inters_t_; i.p = ld(p); i.normal = ld(normal); i.inside = inside;
    scat_t s = scat_init(&i);
    material_scatter(&s, &i, ld(wo), mat, ld(color), rng);
    st(out->wi, s.wi); out->specular = s.specular; st(out->throughput, s.throughput);
    out->refracted = s.refracted; out->t = s.t;
}

void orc_material_scatter(float inters_t_, const float normal[3], int inside, const float wo[3],
                          const rt_material* mat, const float color[3], uint32_t* rng, orc_scatter* out) {
    inters_t i; memset(&i, 0, sizeof i);
    i.t = inters_t_; i.normal = ld(normal); i.inside = inside;
    scat_t s = scat_init(&i);
    material_scatter(&s, &i, ld(wo), mat, ld(color), rng);
    st(out->wi, s.wi); out->specular = s.specular; st(out->throughput, s.throughput);
    out->refracted = s.refracted; out->t = s.t;
}

/* ---------------------------------------------------------------- kernels.cu -------------- */

enum { OBJ_NONE = 0, OBJ_TRIMESH = 1, OBJ_PLANE = 2, OBJ_LIGHT = 3, OBJ_SPHERE = 4 };   /* kernels.cu:40-45 (+SPHERE) */

typedef struct {            /* path, helper_structs.h:48-71 */
    v3 origin, rayDir, color;
    int specular;
    v3 shadowDir, lightContribution;
    int bounce;
    v3 attenuation;
    uint32_t rng;
    int inside;
} path_t;

typedef struct {
    const orc_scene* sc;
    const rt_render_options* opt;
    orc_counters* cnt;
    int max_depth;
} ctx_t;

#define STAT(c, k) do { if ((c)->cnt) (c)->cnt->ref_stats[k]++; } while (0)     /* context.rayStat(), kernels.cu:103-105 */

static inline void pop_bitstack(uint32_t* bitStack, int* idx) {     /* kernels.cu:148-152 */
    int m = __builtin_ffsll((long long)*bitStack) - 1;
    *bitStack = (*bitStack >> m) ^ 1u;
    *idx = (*idx >> m) ^ 1;
}

static float hit_bvh(const orc_scene* sc, const ray_t* r, float t_min, float t_max, int is_shadow,
                     uint32_t* triId, float* hu, float* hv, orc_counters* cnt) {         /* kernels.cu:154-224 */
    int idx = 1;
    float closest = t_max;
    uint32_t bitStack = 1;
    const uint32_t firstLeafIdx = (uint32_t)(sc->num_bvh_nodes / 2);                     /* kernels.cu:614 */
    while (idx) {
        if ((uint32_t)idx < firstLeafIdx) {
            int idx2 = idx << 1;
            const rt_bvh_node* L = &sc->bvh[idx2];
            const rt_bvh_node* R = &sc->bvh[idx2 + 1];
            if (cnt) cnt->node_visits++;
            float leftHit = hit_bbox_dist(ldv(L->a), ldv(L->b), r, closest);
            int traverseLeft = leftHit < closest;
            float rightHit = hit_bbox_dist(ldv(R->a), ldv(R->b), r, closest);
            int traverseRight = rightHit < closest;
            int swap = rightHit < leftHit;
            if (traverseLeft && traverseRight) {
                if (cnt) cnt->ref_stats[RT_STAT_NODES_BOTH]++;          /* BVH_COUNT, kernels.cu:184-186,219-222 */
                idx = idx2 + (swap ? 1 : 0);
                bitStack = (bitStack << 1) + 1;
            } else if (traverseLeft || traverseRight) {
                if (cnt) cnt->ref_stats[RT_STAT_NODES_SINGLE]++;
                idx = idx2 + (swap ? 1 : 0);
                bitStack = bitStack << 1;
            } else {
                pop_bitstack(&bitStack, &idx);
            }
        } else {
            int first = (int)((uint32_t)idx - firstLeafIdx) * sc->nppl;
            for (int i = 0; i < sc->nppl; i++) {
                const rt_triangle* tri = &sc->tris[first + i];
                if (isinf(tri->v[0].e[0])) break;
                float u, v;
                if (cnt) cnt->prim_tests++;
                float hitT = triangle_hit(tri, r, t_min, closest, &u, &v);
                if (hitT < closest) {
                    if (is_shadow) return 0.0f;
                    closest = hitT;
                    *triId = (uint32_t)(first + i);
                    *hu = u; *hv = v;
                }
            }
            pop_bitstack(&bitStack, &idx);
        }
    }
    return closest;
}

float orc_hit_bvh(const orc_scene* sc, const float org[3], const float dir_in[3], float t_min, float t_max,
                  int is_shadow, uint32_t* tri_id, float* u, float* v, orc_counters* cnt) {
    ray_t r = mkray(ld(org), ld(dir_in));
    return hit_bvh(sc, &r, t_min, t_max, is_shadow, tri_id, u, v, cnt);
}

static float hit_mesh(const ctx_t* c, const ray_t* r, float t_min, float t_max, int is_shadow, int primary,
                      uint32_t* triId, float* hu, float* hv) {                           /* kernels.cu:296-323 */
    if (!hit_bbox(ldv(c->sc->bounds.min), ldv(c->sc->bounds.max), r, t_max)) {
        if (is_shadow) STAT(c, RT_STAT_SHADOWS_BBOX_NOHITS);                             /* kernels.cu:298-301 */
        else STAT(c, primary ? RT_STAT_PRIMARY_BBOX_NOHITS : RT_STAT_SECONDARY_BBOX_NOHIT);
        return FLT_MAX;
    }
    return hit_bvh(c->sc, r, t_min, t_max, is_shadow, triId, hu, hv, c->cnt);
}

/* Sphere-scene closest hit: the linear scan the README-era renderer did over its sphere list,
 * each test being sphereHit (intersections.h:85-104) against the running closest t.  For
 * shadow rays any hit ends the scan (same any-hit rule as hitBvh, kernels.cu:205). */
static float hit_spheres(const ctx_t* c, const ray_t* r, float t_min, float t_max, int is_shadow, int* sid) {
    float closest = t_max;
    for (int k = 0; k < c->sc->num_spheres; k++) {
        if (c->cnt) c->cnt->prim_tests++;
        float t = sphere_hit(&c->sc->spheres[k], r, t_min, closest);
        if (t < closest) {
            if (is_shadow) return 0.0f;
            closest = t;
            *sid = k;
        }
    }
    return closest;
}

/* hit(), kernels.cu:325-360.  Mesh scenes follow it line by line.  Sphere scenes replace the
 * hitMesh call by hit_spheres and the triangle normal by (p - center) / radius
 * (intersections.h:95, the commented rec.normal); everything else is shared. */
static int hit(const ctx_t* c, const path_t* p, float t_max, int is_shadow, inters_t* in, int* sid) {
    const ray_t r = is_shadow ? mkray(p->origin, p->shadowDir) : mkray(p->origin, p->rayDir);
    const float eps = c->opt->t_min;
    in->objId = OBJ_NONE;
    if (c->sc->num_spheres > 0) {
        int k = -1;
        if ((in->t = hit_spheres(c, &r, eps, t_max, is_shadow, &k)) < t_max) {
            if (is_shadow) return 1;
            in->objId = OBJ_SPHERE;
            *sid = k;
            in->p = point_at(&r, in->t);
            in->normal = divs(sub(in->p, ldv(c->sc->spheres[k].center)), c->sc->spheres[k].radius);
        } else {
            if (is_shadow) return 0;
        }
    } else {
        uint32_t triId = 0; float hu = 0, hv = 0;
        const int primary = p->bounce == 0;                                            /* kernels.cu:328 */
        if ((in->t = hit_mesh(c, &r, eps, t_max, is_shadow, primary, &triId, &hu, &hv)) < t_max) {
            if (is_shadow) return 1;
            in->objId = OBJ_TRIMESH;
            const rt_triangle* tri = &c->sc->tris[triId];
            in->meshID = tri->meshID;
            in->normal = unit(cross(sub(ldv(tri->v[1]), ldv(tri->v[0])), sub(ldv(tri->v[2]), ldv(tri->v[0]))));
            in->texCoords[0] = (hu * tri->texCoords[1 * 2 + 0] + hv * tri->texCoords[2 * 2 + 0] + (1 - hu - hv) * tri->texCoords[0 * 2 + 0]);
            in->texCoords[1] = (hu * tri->texCoords[1 * 2 + 1] + hv * tri->texCoords[2 * 2 + 1] + (1 - hu - hv) * tri->texCoords[0 * 2 + 1]);
        } else {
            if (is_shadow) return 0;
            /* kernels.cu:341-345: the floor, commented out at HEAD; rt_render_options.floor = 1 re-enables the call site */
            if (c->opt->floor && (in->t = plane_hit(&c->sc->floor, &r, eps, FLT_MAX)) < FLT_MAX) {
                in->objId = OBJ_PLANE;
                in->normal = ldv(c->sc->floor.norm);
            } else
            if (p->specular && sphere_hit(&c->opt->light, &r, eps, t_max) < t_max) {   /* kernels.cu:346 */
                in->objId = OBJ_LIGHT;
                return 1;
            }
        }
    }
    if (in->objId != OBJ_NONE) {
        if (in->objId == OBJ_TRIMESH || in->objId == OBJ_PLANE) in->p = point_at(&r, in->t);
        if (dot(r.B, in->normal) > 0.0f) in->normal = neg(in->normal);
        return 1;
    }
    return 0;
}

static int generate_shadow_ray(const rt_render_options* opt, path_t* p, v3 normal, float* lightDist, float* cosAMax_out) {  /* kernels.cu:363-393 */
    const rt_sphere* light = &opt->light;
    const v3 lc = ldv(light->center);
    const v3 sw = unit(sub(lc, p->origin));
    const v3 su = unit(cross(fabsf(sw.x) > 0.01f ? V(0, 1, 0) : V(1, 0, 0), sw));
    const v3 sv = cross(sw, su);

    const float cosAMax = sqrtf(1.0f - light->radius * light->radius / sqlen(sub(p->origin, lc)));
    if (cosAMax_out) *cosAMax_out = cosAMax;
    if (isnan(cosAMax)) return 0;

    const float eps1 = rnd(&p->rng);
    const float eps2 = rnd(&p->rng);
    const float cosA = 1.0f - eps1 + eps1 * cosAMax;
    const float sinA = sqrtf(1.0f - cosA * cosA);
    const float phi = (float)(2 * M_PI * (double)eps2);                  /* :378 double product, narrowed */
    const v3 l = add(add(muls(sinA, muls(cosf(phi), su)), muls(sinA, muls(sinf(phi), sv))), muls(cosA, sw));

    const float dotl = dot(l, normal);
    if (dotl <= 0) return 0;

    p->shadowDir = unit(l);
    const float omega = (float)(2 * M_PI * (double)(1.0f - cosAMax));    /* :386 */
    p->lightContribution = divs(muls(omega, muls(dotl, mulv(p->attenuation, ldv(opt->lightColor)))), (float)M_PI); /* :387 */
    *lightDist = len(sub(lc, p->origin)) - light->radius;
    return 1;
}

int orc_generate_shadow_ray(const rt_render_options* opt, const float origin[3], const float attenuation[3], const float normal[3],
                            uint32_t* rng, float out9[9]) {
    path_t p; memset(&p, 0, sizeof p);
    p.origin = ld(origin); p.attenuation = ld(attenuation); p.rng = *rng;
    uint64_t draws = 0, *saved = g_draws;
    g_draws = &draws;
    float lightDist = 0.0f, cosAMax = 0.0f;
    const int ok = generate_shadow_ray(opt, &p, ld(normal), &lightDist, &cosAMax);
    g_draws = saved;
    st(out9, p.shadowDir); st(out9 + 3, p.lightContribution);
    out9[6] = lightDist; out9[7] = cosAMax; out9[8] = (float)draws;
    *rng = p.rng;
    return ok;
}

static void color(const ctx_t* c, path_t* p) {                                          /* kernels.cu:396-533 */
    p->attenuation = V(1.0f, 1.0f, 1.0f);
    p->color = V(0, 0, 0);
    const int maxDepth = c->max_depth > 255 ? 255 : c->max_depth;       /* uint8_t bounce, helper_structs.h:58 */
    int fromMesh = 0;                                                   /* STATS, kernels.cu:399-401 */
    for (p->bounce = 0; p->bounce < maxDepth; p->bounce++) {
        const int primary = p->bounce == 0;                             /* STATS, kernels.cu:403-408 */
        STAT(c, primary ? RT_STAT_PRIMARY : RT_STAT_SECONDARY);
        if (fromMesh) STAT(c, RT_STAT_SECONDARY_MESH);
        if (len(p->attenuation) < 0.01f) STAT(c, RT_STAT_LOW_POWER);
        inters_t in; memset(&in, 0, sizeof in);
        int sid = -1;
        if (c->cnt) c->cnt->rays++;
        if (!hit(c, p, FLT_MAX, 0, &in, &sid)) {
            if (primary) STAT(c, RT_STAT_PRIMARY_NOHITS);               /* kernels.cu:414-417 */
            else STAT(c, fromMesh ? RT_STAT_SECONDARY_MESH_NOHIT : RT_STAT_SECONDARY_NOHIT);
            if (c->opt->sky == RT_SKY_GRADIENT) {                       /* kernels.cu:419-421 */
                float t = 0.5f * (p->rayDir.y + 1.0f);
                v3 sky = add(muls((1.0f - t), V(1.0f, 1.0f, 1.0f)), muls(t, V(0.5f, 0.7f, 1.0f)));
                p->color = add(p->color, mulv(p->attenuation, sky));
            } else {                                                    /* kernels.cu:424 */
                p->color = add(p->color, mulv(p->attenuation, V(0.5f, 0.5f, 0.5f)));
            }
            return;
        }
        if (c->cnt) c->cnt->hits++;
        fromMesh = (in.objId == OBJ_TRIMESH || in.objId == OBJ_SPHERE); /* kernels.cu:428-432 (a sphere scene's spheres are its mesh) */
        if (primary && !fromMesh) STAT(c, RT_STAT_PRIMARY_NOHITS);
        if (primary && fromMesh) STAT(c, RT_STAT_PRIMARY_HIT_MESH);
        if (in.objId == OBJ_LIGHT) {                                    /* kernels.cu:433-447 */
            if (!c->opt->nee)                                           /* #ifndef SHADOW branch, :444-445 */
                p->color = add(p->color, mulv(p->attenuation, ldv(c->opt->lightColor)));
            return;
        }

        in.inside = p->inside;
        scat_t sc = scat_init(&in);
        if (in.objId == OBJ_SPHERE) {
            const rt_material* mat = &c->sc->sphere_materials[sid];
            material_scatter(&sc, &in, p->rayDir, mat, ldv(mat->color), &p->rng);
        } else if (in.objId == OBJ_TRIMESH) {                           /* kernels.cu:452-480 */
            const rt_material* mat = &c->sc->materials[in.meshID];
            v3 albedo;
            if (mat->texId != -1) {
                int texId = mat->texId;
                int width = c->sc->textures[texId].width;
                int height = c->sc->textures[texId].height;
                float tu = in.texCoords[0];
                tu = tu - floorf(tu);
                float tv = in.texCoords[1];
                tv = tv - floorf(tv);
                const int tx = (int)((float)(width - 1) * tu);
                const int ty = (int)((float)(height - 1) * tv);
                const int tIdx = ty * width + tx;
                const float* d = c->sc->textures[texId].data;
                albedo = V(d[tIdx * 3 + 0], d[tIdx * 3 + 1], d[tIdx * 3 + 2]);
            } else {
                albedo = ldv(mat->color);
            }
            material_scatter(&sc, &in, p->rayDir, mat, albedo, &p->rng);
        } else {                                                        /* kernels.cu:481-482: the floor */
            preset_scatter(&sc, &in, p->rayDir, RT_FLOOR_DIFFUSE, &p->rng);
        }

        p->origin = add(p->origin, muls(sc.t, p->rayDir));             /* kernels.cu:485-489 */
        p->rayDir = sc.wi;
        p->attenuation = mulv(p->attenuation, sc.throughput);
        p->specular = sc.specular;
        p->inside = sc.refracted ? !p->inside : p->inside;

        if (c->opt->nee) {                                              /* kernels.cu:490-511 */
            float lightDist;
            if (!p->specular && generate_shadow_ray(c->opt, p, in.normal, &lightDist, 0)) {
                inters_t sh; memset(&sh, 0, sizeof sh);
                int s2 = -1;
                if (c->cnt) c->cnt->shadow_rays++;
                STAT(c, RT_STAT_SHADOWS);
                if (!hit(c, p, lightDist, 1, &sh, &s2)) {
                    STAT(c, RT_STAT_SHADOWS_NOHITS);
                    p->color = add(p->color, p->lightContribution);
                }
            }
        }
        if (c->opt->rr) {                                               /* kernels.cu:512-527 */
            if (p->bounce > 3) {
                float m = max3(p->attenuation);
                if (rnd(&p->rng) > m) { STAT(c, RT_STAT_RUSSIAN_KILL); return; }
                float k = 1 / m;
                p->attenuation = V(p->attenuation.x * k, p->attenuation.y * k, p->attenuation.z * k); /* vec3.h:177-182 */
            }
        }
    }
    STAT(c, RT_STAT_EXCEED_MAX_BOUNCE);                                 /* kernels.cu:529-531 */
}

void orc_render(const orc_scene* scn, const rt_camera* cam, const rt_render_options* opt,
                int nx, int ny, int ns, int max_depth,
                int x0, int y0, int x1, int y1, rt_vec3* fb, orc_counters* counters) {  /* kernels.cu:535-569 */
    ctx_t c; c.sc = scn; c.opt = opt; c.cnt = counters; c.max_depth = max_depth;
    g_draws = counters ? &counters->rng_draws : 0;
    for (int j = y0; j < y1; j++) {
        for (int i = x0; i < x1; i++) {
            path_t p; memset(&p, 0, sizeof p);
            uint32_t pixelId = (uint32_t)(j * nx + i);
            p.rng = orc_pixel_seed(pixelId);
            v3 col = V(0, 0, 0);
            for (int s = 0; s < ns; s++) {
                if (opt->rng == RT_RNG_COUNTER) p.rng = sample_seed(pixelId, (uint32_t)s);
                float u = ((float)i + rnd(&p.rng)) / (float)nx;
                float v = ((float)j + rnd(&p.rng)) / (float)ny;
                ray_t r = get_ray(cam, u, v, &p.rng);
                p.origin = r.A;
                p.rayDir = r.B;
                p.specular = 0;
                p.inside = 0;
                color(&c, &p);
                col = add(col, p.color);
                if (counters) counters->samples++;
                if (counters && (isnan(p.color.x) || isnan(p.color.y) || isnan(p.color.z))) counters->ref_stats[RT_STAT_NAN]++;   /* kernels.cu:559-561 */
            }
            st(fb[pixelId].e, divs(col, (float)ns));
        }
    }
    g_draws = 0;
}

/* ---------------------------------------------------------------- harness pieces ---------- */

uint32_t orc_linear_to_srgb(float x) {          /* staircase_scene.h:22-30 */
    x = fmaxf(x, 0.0f);
    x = fmaxf(1.055f * powf(x, 0.416666667f) - 0.055f, 0.0f);
    uint32_t u = (uint32_t)(x * 255.9f);
    u = u < 255u ? u : 255u;
    return u;
}

double orc_rmse(const rt_vec3* f, const rt_vec3* g, int n) {    /* main.cpp:117-125 */
    double error = 0.0;
    for (int i = 0; i < n; i++)
        for (int c = 0; c < 3; c++)
            error += (f[i].e[c] - g[i].e[c]) * (f[i].e[c] - g[i].e[c]) / 3.0;
    return sqrt(error / n);
}

/* sizeof of every struct the Python binding mirrors (oracle/oracle.py checks them at load time: orc_render writes sizeof(orc_counters)
 * bytes through the caller's pointer, so a stale mirror would be a heap overrun, not an error message). */
int orc_abi_sizes(int32_t* out, int n) {
    const int32_t s[4] = { (int32_t)sizeof(orc_scene), (int32_t)sizeof(orc_counters), (int32_t)sizeof(orc_scatter), (int32_t)sizeof(rt_render_options) };
    for (int k = 0; k < n && k < 4; k++) out[k] = s[k];
    return 4;
}

/* ---- the device's sine / cosine, checked against THIS machine's libm -----------------------------------------------------------------
 * cuda-raytracing-optimized_amd/csrc/rt_glibc_sincosf.h restates glibc's sinf / cosf / sincosf (the functions generate_shadow_ray above calls)
 * for the device; the same text is compiled here for the host and compared with libm.  Test infrastructure (tests/test_oracle_golden.py). */
#define RT_SINCOS_FN static inline
#include "../cuda-raytracing-optimized_amd/csrc/rt_glibc_sincosf.h"

int orc_glibc_sincosf_twin(float y, float* s, float* c) { return rt_glibc_sincosf(y, s, c); }

/* mode 0: the 2^24 arguments of generateShadowRay, phi_k = (float)(2 * M_PI * (k / 16777216.0f)) for k in [k0, k1);
 * mode 1: arguments k * step for k in [k0, k1) (signed sweep).  Returns the number of arguments on which the twin differs in any bit from libm's
 * sinf, cosf or sincosf; *first_bad receives the first such argument. */
long orc_glibc_sincosf_twin_mismatches(int mode, long k0, long k1, float step, float* first_bad) {
    long bad = 0;
    for (long k = k0; k < k1; k++) {
        const float y = mode == 0 ? (float)(2 * M_PI * (double)((float)k / 16777216.0f)) : (float)k * step;
        float ts, tc, ls, lc;
        if (!rt_glibc_sincosf(y, &ts, &tc)) continue;
        const float s1 = sinf(y), c1 = cosf(y);
        sincosf(y, &ls, &lc);
        if (memcmp(&ts, &s1, 4) || memcmp(&tc, &c1, 4) || memcmp(&ts, &ls, 4) || memcmp(&tc, &lc, 4)) {
            if (bad == 0 && first_bad) *first_bad = y;
            bad++;
        }
    }
    return bad;
}

/* libm's sincosf over an array (the expected values of the device probe rtProbeSinCos) */
void orc_libm_sincosf(const float* y, int n, float* s, float* c) {
    for (int k = 0; k < n; k++) { s[k] = sinf(y[k]); c[k] = cosf(y[k]); }
}

/* ---- the device's powf(x, 5), checked against THIS machine's libm ---------------------------------------------------------------------
 * cuda-raytracing-optimized_amd/csrc/rt_glibc_powf.h restates glibc's powf for the exponent 5 (orc_schlick above calls powf: material.h:12) for the
 * device; the same text is compiled here for the host and compared with libm.  Test infrastructure (tests/test_oracle_golden.py). */
#define RT_POWF_FN static inline
#include "../cuda-raytracing-optimized_amd/csrc/rt_glibc_powf.h"

float orc_glibc_powf5_twin(float x) { return rt_glibc_powf5(x); }

#include <pthread.h>
typedef struct { uint64_t lo, hi, stride; long bad; uint32_t first_bad; uint32_t max_ulps; } powf5_job;
static void* powf5_worker(void* arg) {
    powf5_job* j = (powf5_job*)arg;
    for (uint64_t b = j->lo; b < j->hi; b += j->stride) {
        const uint32_t u = (uint32_t)b;
        float x, t, l;
        memcpy(&x, &u, 4);
        t = rt_glibc_powf5(x);
        l = powf(x, 5.0f);
        if (memcmp(&t, &l, 4) && !(t != t && l != l)) {           /* (NaN payloads: any NaN equals any NaN) */
            if (j->bad == 0) j->first_bad = u;
            j->bad++;
        }
        if (x >= 0.0f && x <= 2.5f) {                              /* the bracket of rt_device.h schlick_above: |libm - fl(x^5 in fp64)| in ulps */
            const double d = (double)x, d2 = d * d;
            const float f = (float)(d2 * d2 * d);
            uint32_t fb, lb;
            memcpy(&fb, &f, 4); memcpy(&lb, &l, 4);
            const uint32_t dist = fb > lb ? fb - lb : lb - fb;
            if (dist > j->max_ulps) j->max_ulps = dist;
        }
    }
    return 0;
}
/* Bit patterns [lo, hi) in steps of `stride`, split over `threads` host threads.  Returns the number of arguments on which the twin differs from
 * libm's powf(x, 5.0f) in any bit; *first_bad receives the bit pattern of one such argument; *max_ulps is raised to the largest distance, in units of
 * the last place, between libm's result and the fp64 product x^5 rounded once, over the arguments in [0, 2.5]. */
long orc_glibc_powf5_twin_mismatches(uint64_t lo, uint64_t hi, uint64_t stride, int threads, uint32_t* first_bad, uint32_t* max_ulps) {
    pthread_t th[64];
    powf5_job job[64];
    long bad = 0;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    const uint64_t steps = (hi - lo + stride - 1) / stride, per = (steps + threads - 1) / threads;
    for (int k = 0; k < threads; k++) {
        job[k].lo = lo + (uint64_t)k * per * stride;
        job[k].hi = job[k].lo + per * stride < hi ? job[k].lo + per * stride : hi;
        if (job[k].lo > hi) job[k].lo = hi;
        job[k].stride = stride; job[k].bad = 0; job[k].first_bad = 0; job[k].max_ulps = 0;
        pthread_create(&th[k], 0, powf5_worker, &job[k]);
    }
    for (int k = 0; k < threads; k++) {
        pthread_join(th[k], 0);
        if (job[k].bad && bad == 0 && first_bad) *first_bad = job[k].first_bad;
        bad += job[k].bad;
        if (max_ulps && job[k].max_ulps > *max_ulps) *max_ulps = job[k].max_ulps;
    }
    return bad;
}

/* libm's powf(x, 5.0f) and orc_schlick over arrays (the expected values of the device probes) */
void orc_libm_powf5(const float* x, int n, float* out) {
    for (int k = 0; k < n; k++) out[k] = powf(x[k], 5.0f);
}
void orc_schlick_array(const float* cosine, const float* ref_idx, int n, float* out) {
    for (int k = 0; k < n; k++) out[k] = orc_schlick(cosine[k], ref_idx[k]);
}

/* ---- the device's vec3 / float and unit_vector through fp64, checked against IEEE division and sqrtf ----------------------------------------
 * cuda-raytracing-optimized_amd/csrc/rt_div64.h computes (ax, ay, az) / t and sqrt(s) through an fp64 reciprocal / iteration; the same text is compiled
 * here and compared with the plain fp32 operators, with the hardware seeds (v_rcp_f32, v_rsq_f32: 1 ulp) replaced by fp32 values pushed off by up to
 * +-2 ulp - the result must not depend on them.  Test infrastructure (tests/test_oracle_golden.py). */
#define RT_DIV64_FN static inline
#include "../cuda-raytracing-optimized_amd/csrc/rt_div64.h"

typedef struct { uint64_t seed; long n; int mode; long bad, done; float bad_x, bad_y; } div64_job;
static inline uint64_t d64_next(uint64_t* s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }
static inline float d64_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float d64_nudge(float f, int k) { uint32_t u; memcpy(&u, &f, 4); u += (uint32_t)k; memcpy(&f, &u, 4); return f; }
static void* div64_worker(void* arg) {
    div64_job* j = (div64_job*)arg;
    uint64_t st = j->seed * 0x9E3779B97F4A7C15ull + 1;
    for (long i = 0; i < j->n; i++) {
        const uint64_t a = d64_next(&st), b = d64_next(&st);
        const int nud = (int)(b >> 61) - 3 + ((b >> 60) & 1);                /* -3 .. +4 -> clamp to +-2 below */
        const int k = nud < -2 ? -2 : (nud > 2 ? 2 : nud);
        if (j->mode == 0 || j->mode == 1) {                                 /* division: random operands (0) or quotients beside a rounding boundary (1) */
            float y = d64_bits(0x3F800000u - (40u << 23) + (uint32_t)((a >> 8) % (80u << 23)));       /* 2^-40 .. 2^40 */
            float x;
            if (j->mode == 0) x = d64_bits((uint32_t)(b & 0x7FFFFFFFu) % 0x7F000000u) * ((a & 1) ? -1.0f : 1.0f);
            else {                                                          /* x = RN(y * (2M + 1) 2^-25 * 2^e): x / y lies within an ulp of x of a boundary */
                const double m = (double)((((b >> 8) & 0xFFFFFFu) | 0x800000u) * 2u + 1u) * 0x1p-25;
                x = (float)((double)y * m * (double)d64_bits(0x3F800000u - (20u << 23) + (uint32_t)((b >> 40) % 40u) * (1u << 23)));
            }
            if (a & 2) y = -y;
            const float ay = fabsf(y);
            if (!(ay >= 0x1p-60f && ay <= 0x1p60f)) continue;
            const float x2 = d64_bits((uint32_t)(a >> 32)), x3 = x * 0.333f;
            float q[3];
            const double r = rt_recip64((double)y, (double)d64_nudge(1.0f / y, k));
            if (!rt_div3_64(x, x2, x3, r, q)) continue;                     /* left to the plain operators */
            const float e0 = x / y, e1 = x2 / y, e2 = x3 / y;
            j->done++;
            if (memcmp(&q[0], &e0, 4) || (memcmp(&q[1], &e1, 4) && !(q[1] != q[1] && e1 != e1)) || memcmp(&q[2], &e2, 4)) { if (!j->bad) { j->bad_x = x; j->bad_y = y; } j->bad++; }
        } else {                                                            /* square root: random (2) or beside a boundary (3), then the unit-vector division by it */
            float s;
            if (j->mode == 2) s = d64_bits(0x3F800000u - (90u << 23) + (uint32_t)((a >> 8) % (180u << 23)));
            else {
                const double m = (double)((((b >> 8) & 0xFFFFFFu) | 0x800000u) * 2u + 1u) * 0x1p-25 * ((a & 4) ? 1.0 : 1.4142135623730951);
                s = (float)(m * m * (double)d64_bits(0x3F800000u - (20u << 23) + (uint32_t)((b >> 40) % 40u) * (1u << 23)));
            }
            if (!(s >= 0x1p-100f && s <= 0x1p100f)) continue;
            double h;
            const float l = rt_sqrt64(s, (double)d64_nudge(1.0f / sqrtf(s), k), &h), e = sqrtf(s);
            int bad = memcmp(&l, &e, 4) != 0;
            float q[3];
            j->done++;
            const float x = d64_bits((uint32_t)(b & 0x7FFFFFFFu) % 0x7F000000u) * 1e-19f;
            if (!bad && rt_div3_64(x, l, -l, rt_recip64((double)l, h + h), q)) {
                const float e0 = x / l, e1 = l / l, e2 = -l / l;
                bad = memcmp(&q[0], &e0, 4) || memcmp(&q[1], &e1, 4) || memcmp(&q[2], &e2, 4);
            }
            if (bad) { if (!j->bad) { j->bad_x = s; j->bad_y = l; } j->bad++; }
        }
    }
    return 0;
}
/* n cases of `mode` (0 / 1 division: random / beside a rounding boundary; 2 / 3 square root likewise) over `threads` host threads: the number of cases in
 * which rt_div64.h's result differs from the plain fp32 operator in any bit; bad2 = the operands of one such case. */
long orc_div64_twin_mismatches(int mode, long n, uint64_t seed, int threads, float* bad2, long* compared) {
    pthread_t th[64];
    div64_job job[64];
    long bad = 0;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    for (int k = 0; k < threads; k++) {
        job[k].seed = seed * 64 + (uint64_t)k; job[k].n = n / threads; job[k].mode = mode; job[k].bad = 0; job[k].done = 0; job[k].bad_x = job[k].bad_y = 0.0f;
        pthread_create(&th[k], 0, div64_worker, &job[k]);
    }
    for (int k = 0; k < threads; k++) {
        pthread_join(th[k], 0);
        if (job[k].bad && bad == 0 && bad2) { bad2[0] = job[k].bad_x; bad2[1] = job[k].bad_y; }
        bad += job[k].bad;
        if (compared) *compared += job[k].done;
    }
    return bad;
}
