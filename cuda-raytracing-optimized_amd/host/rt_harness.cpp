// rt_harness.cpp — output + verification harness (librt_host.so): the artefacts the reference's
// main.cpp produces and consumes around the render call.
#include "../../include/rt_host.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

extern "C" {

// staircase_scene.h:22-30: sRGB approximation, then x*255.9 truncated and clamped to 255.
uint32_t rtLinearToSRGB(float x) {
    x = fmaxf(x, 0.0f);
    x = fmaxf(1.055f * powf(x, 0.416666667f) - 0.055f, 0.0f);
    uint32_t u = (uint32_t)(x * 255.9f);
    return u < 255u ? u : 255u;
}

// staircase_scene.h:32-43: "P3\n<nx> <ny>\n255\n" then one "r g b\n" per pixel, top row first.
// (The reference writes to stdout; we write to `path`, "-" meaning stdout.)
int rtWritePPM(const char* path, int nx, int ny, const rt_vec3* colors) {
    FILE* f = (strcmp(path, "-") == 0) ? stdout : fopen(path, "w");
    if (!f) return -1;
    fprintf(f, "P3\n%d %d\n255\n", nx, ny);
    for (int j = ny - 1; j >= 0; j--)
        for (int i = 0; i < nx; i++) {
            const rt_vec3& c = colors[(size_t)j * nx + i];
            fprintf(f, "%u %u %u\n", rtLinearToSRGB(c.e[0]), rtLinearToSRGB(c.e[1]), rtLinearToSRGB(c.e[2]));
        }
    if (f != stdout) fclose(f);
    return 0;
}

static const char kRefHeader[] = "REF_00.01";    // main.cpp:27,39 — written with its NUL

// main.cpp:25-33
int rtSaveReference(const char* path, int nx, int ny, const rt_vec3* colors) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    bool ok = fwrite(kRefHeader, 1, sizeof kRefHeader, f) == sizeof kRefHeader;
    ok = ok && fwrite(&nx, sizeof(int), 1, f) == 1 && fwrite(&ny, sizeof(int), 1, f) == 1;
    ok = ok && fwrite(colors, sizeof(rt_vec3), (size_t)nx * ny, f) == (size_t)nx * ny;
    fclose(f);
    return ok ? 0 : -1;
}

// main.cpp:36-60
int rtLoadReference(const char* path, rt_vec3* reference, int nx, int ny) {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    char header[sizeof kRefHeader];
    if (fread(header, 1, sizeof header, f) != sizeof header || memcmp(header, kRefHeader, sizeof header) != 0) {
        fclose(f);
        return -1;
    }
    int inNx = 0, inNy = 0;
    if (fread(&inNx, sizeof(int), 1, f) != 1 || fread(&inNy, sizeof(int), 1, f) != 1 || inNx != nx || inNy != ny) {
        fclose(f);
        return -2;
    }
    const size_t n = (size_t)nx * ny;
    const bool ok = fread(reference, sizeof(rt_vec3), n, f) == n;
    fclose(f);
    return ok ? 0 : -1;
}

// main.cpp:117-125
double rtRmse(const rt_vec3* f, const rt_vec3* g, int nx, int ny) {
    double error = 0.0;
    const size_t n = (size_t)nx * ny;
    for (size_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) {
            const float d = f[i].e[c] - g[i].e[c];
            error += d * d / 3.0;
        }
    return sqrt(error / (double)n);
}

}  // extern "C"
