// rt_scenes.cpp — host-side camera + sphere-scene set-up (librt_host.so).
//
// Compiled -O2 -ffp-contract=off so the camera vectors are the same bits whichever compiler
// builds this file (SURVEY.md §8c finding 2/3); all inputs arrive as run-time arguments.
#include "../../include/rt_host.h"

#include <cmath>
#include <cstring>

namespace {

struct f3 { float x, y, z; };
inline f3 F(const float* p) { return { p[0], p[1], p[2] }; }
inline f3 operator-(f3 a, f3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline f3 scale(float t, f3 a) { return { t * a.x, t * a.y, t * a.z }; }
inline f3 cross(f3 a, f3 b) { return { a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x }; }
inline f3 unit(f3 a) {
    float l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return { a.x / l, a.y / l, a.z / l };
}
inline void put(rt_vec3& d, f3 a) { d.e[0] = a.x; d.e[1] = a.y; d.e[2] = a.z; }

inline rt_sphere mk_sphere(float x, float y, float z, float r) {
    rt_sphere s; s.center.e[0] = x; s.center.e[1] = y; s.center.e[2] = z; s.radius = r; return s;
}
inline rt_material mk_mat(int type, float r, float g, float b, float param) {
    rt_material m; m.type = type; m.color.e[0] = r; m.color.e[1] = g; m.color.e[2] = b; m.param = param; m.texId = -1; return m;
}

}  // namespace

extern "C" {

// helper_structs.h:194-207: thin-lens camera from look-at parameters; vfov in degrees.
void rtMakeCamera(const float lookfrom_[3], const float lookat_[3], const float vup_[3], float vfov,
                  float aspect, float aperture, float focus_dist, rt_camera* out) {
    const f3 lookfrom = F(lookfrom_), lookat = F(lookat_), vup = F(vup_);
    const float theta = vfov * ((float)M_PI) / 180.0f;
    const float half_height = std::tan(theta / 2.0f);       // float overload, as in the reference
    const float half_width = aspect * half_height;
    const f3 w = unit(lookfrom - lookat);
    const f3 u = unit(cross(vup, w));
    const f3 v = cross(w, u);
    const f3 llc = ((lookfrom - scale(half_width * focus_dist, u)) - scale(half_height * focus_dist, v)) - scale(focus_dist, w);
    put(out->origin, lookfrom);
    put(out->lower_left_corner, llc);
    put(out->horizontal, scale(2.0f * half_width * focus_dist, u));
    put(out->vertical, scale(2.0f * half_height * focus_dist, v));
    put(out->u, u); put(out->v, v); put(out->w, w);
    out->lens_radius = aperture / 2.0f;
}

// main.cpp:17-20
float rtRandomFloat(uint32_t* state) {
    *state = (214013u * *state + 2531011u);
    return (float)((*state >> 16) & 0x7FFF) / 32767;
}

int rtSceneThreeSpheres(rt_sphere* spheres, rt_material* materials, int cap, int nx, int ny, rt_camera* cam) {
    if (cap < 3) return -3;
    spheres[0] = mk_sphere(0.0f, -100.5f, -1.0f, 100.0f);  materials[0] = mk_mat(RT_DIFFUSE, 0.8f, 0.8f, 0.0f, 0.0f);
    spheres[1] = mk_sphere(-1.0f, 0.0f, -1.0f, 0.5f);      materials[1] = mk_mat(RT_GLASS, 1.0f, 1.0f, 1.0f, 1.5f);
    spheres[2] = mk_sphere(1.0f, 0.0f, -1.0f, 0.5f);       materials[2] = mk_mat(RT_METAL, 0.8f, 0.6f, 0.2f, 0.0f);
    if (cam) {
        const float from[3] = { 0, 0, 1 }, at[3] = { 0, 0, -1 }, up[3] = { 0, 1, 0 };
        rtMakeCamera(from, at, up, 60.0f, float(nx) / float(ny), 0.0f, 2.0f, cam);
    }
    return 3;
}

int rtSceneRandomSpheres(uint32_t seed, rt_sphere* spheres, rt_material* materials, int cap,
                         int nx, int ny, rt_camera* cam) {
    const int needed = 22 * 22 + 1 + 3;
    if (cap < needed) return -needed;
    uint32_t st = seed;
    int i = 0;
    spheres[i] = mk_sphere(0.0f, -1000.0f, -1.0f, 1000.0f);
    materials[i++] = mk_mat(RT_DIFFUSE, 0.5f, 0.5f, 0.5f, 0.0f);
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            // every draw is its own statement: the order is part of the scene definition
            const float choose_mat = rtRandomFloat(&st);
            const float cx = a + rtRandomFloat(&st);
            const float cz = b + rtRandomFloat(&st);
            spheres[i] = mk_sphere(cx, 0.2f, cz, 0.2f);
            if (choose_mat < 0.8f) {
                const float r1 = rtRandomFloat(&st), r2 = rtRandomFloat(&st);
                const float g1 = rtRandomFloat(&st), g2 = rtRandomFloat(&st);
                const float b1 = rtRandomFloat(&st), b2 = rtRandomFloat(&st);
                materials[i] = mk_mat(RT_DIFFUSE, r1 * r2, g1 * g2, b1 * b2, 0.0f);
            } else if (choose_mat < 0.95f) {
                const float r = rtRandomFloat(&st);
                const float g = rtRandomFloat(&st);
                const float bl = rtRandomFloat(&st);
                const float fuzz = rtRandomFloat(&st);
                materials[i] = mk_mat(RT_METAL, 0.5f * (1.0f + r), 0.5f * (1.0f + g), 0.5f * (1.0f + bl), 0.5f * fuzz);
            } else {
                materials[i] = mk_mat(RT_GLASS, 1.0f, 1.0f, 1.0f, 1.5f);
            }
            i++;
        }
    }
    spheres[i] = mk_sphere(0.0f, 1.0f, 0.0f, 1.0f);   materials[i++] = mk_mat(RT_GLASS, 1.0f, 1.0f, 1.0f, 1.5f);
    spheres[i] = mk_sphere(-4.0f, 1.0f, 0.0f, 1.0f);  materials[i++] = mk_mat(RT_DIFFUSE, 0.4f, 0.2f, 0.1f, 0.0f);
    spheres[i] = mk_sphere(4.0f, 1.0f, 0.0f, 1.0f);   materials[i++] = mk_mat(RT_METAL, 0.7f, 0.6f, 0.5f, 0.0f);
    if (cam) {
        const float from[3] = { 13, 2, 3 }, at[3] = { 0, 0, 0 }, up[3] = { 0, 1, 0 };
        rtMakeCamera(from, at, up, 30.0f, float(nx) / float(ny), 0.1f, 10.0f, cam);
    }
    return i;
}

// staircase_scene.h:62-73 (lookfrom/lookat are double literals narrowed to float there)
void rtStaircaseCamera(int nx, int ny, rt_camera* cam) {
    const float from[3] = { (float)5.555139, (float)173.679901, (float)494.515045 };
    const float at[3] = { (float)5.555139, (float)173.679901, (float)493.515045 };
    const float up[3] = { 0, 1, 0 };
    const f3 d = F(from) - F(at);
    const float dist_to_focus = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
    rtMakeCamera(from, at, up, 42.0f, float(nx) / float(ny), 0.0f, dist_to_focus, cam);
}

}  // extern "C"
