// rt_bvh.cpp — BVH builder, BVH_00.04 file I/O and the procedural staircase mesh (librt_host.so).
//
// The reference ships neither its BVH builder (a sibling project, cuda-raytracing-optimized.sln:8-12)
// nor the staircase asset (staircase_scene.h:122,162 point at the author's disk).  What the
// reference does fix is the LAYOUT its traversal consumes (kernels.cu:154-224,582-614):
//   * complete binary tree, heap indexed: root 1, children 2i and 2i+1, slot 0 unused;
//   * numBvhNodes = 2 * numLeaves, firstLeafIdx = numBvhNodes / 2  (so #leaves is a power of two);
//   * node k = {min.xyz, max.xyz} (24 B);
//   * leaf L owns tris[(L - firstLeafIdx) * nppl .. + nppl); a triangle whose v[0].x is +inf ends the leaf.
// This builder emits exactly that: a top-down build that, at every node, sorts the node's triangles
// by centroid along each axis and takes the SAH-cheapest cut among those that keep both halves
// within the capacity of their (fixed-shape) subtrees.
#include "../../include/rt_host.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

struct rt_host_mesh {
    std::vector<rt_triangle> tris;      // leaf-ordered, padded with sentinels: numLeaves * nppl entries
    std::vector<rt_bvh_node> bvh;       // 2 * numLeaves entries, [0] unused
    rt_bbox bounds;
    int nppl;
};

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; } }
    void grow(const float* p) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    void grow(const Box& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float area() const {
        if (hi[0] < lo[0]) return 0.0f;
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

Box tri_box(const rt_triangle& t) {
    Box b; b.reset();
    for (int k = 0; k < 3; k++) b.grow(t.v[k].e);
    return b;
}

struct Builder {
    const std::vector<rt_triangle>& in;
    std::vector<Box> boxes;
    std::vector<float> cent[3];
    std::vector<int> order;             // permutation being partitioned in place
    rt_host_mesh* out;
    int numLeaves;
    int nppl;

    explicit Builder(const std::vector<rt_triangle>& t) : in(t) {}

    void set_node(int idx, const Box& b) {
        rt_bvh_node& n = out->bvh[idx];
        for (int a = 0; a < 3; a++) { n.a.e[a] = b.lo[a]; n.b.e[a] = b.hi[a]; }
    }

    // node `idx` covers order[begin, end); `leaves` = number of leaf slots below it
    Box build(int idx, int begin, int end, int leaves) {
        Box nb; nb.reset();
        for (int i = begin; i < end; i++) nb.grow(boxes[order[i]]);
        set_node(idx, nb);
        if (leaves == 1) {
            const int leaf = idx - numLeaves;
            for (int i = begin; i < end; i++) out->tris[(size_t)leaf * nppl + (i - begin)] = in[order[i]];
            return nb;
        }
        const int n = end - begin;
        const int capHalf = (leaves / 2) * nppl;
        int nl;
        if (n == 0) {
            nl = 0;
        } else {
            const int lo = std::max(n - capHalf, (n > 1) ? 1 : 0);     // fewest triangles the left half may take
            const int hi = std::min(capHalf, (n > 1) ? n - 1 : n);     // most
            float bestCost = std::numeric_limits<float>::infinity();
            int bestAxis = 0, bestCut = (n + 1) / 2;
            std::vector<float> rightArea(n + 1);
            for (int axis = 0; axis < 3; axis++) {
                const std::vector<float>& c = cent[axis];
                std::sort(order.begin() + begin, order.begin() + end,
                          [&](int x, int y) { return c[x] < c[y] || (c[x] == c[y] && x < y); });
                Box rb; rb.reset();
                rightArea[n] = 0.0f;
                for (int i = n - 1; i >= 0; i--) { rb.grow(boxes[order[begin + i]]); rightArea[i] = rb.area(); }
                Box lb; lb.reset();
                for (int cut = 1; cut <= hi; cut++) {
                    lb.grow(boxes[order[begin + cut - 1]]);
                    if (cut < lo) continue;
                    const float cost = lb.area() * (float)cut + rightArea[cut] * (float)(n - cut);
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestCut = cut; }
                }
            }
            const std::vector<float>& c = cent[bestAxis];
            std::sort(order.begin() + begin, order.begin() + end,
                      [&](int x, int y) { return c[x] < c[y] || (c[x] == c[y] && x < y); });
            nl = std::min(std::max(bestCut, lo), hi);
        }
        build(2 * idx, begin, begin + nl, leaves / 2);
        build(2 * idx + 1, begin + nl, end, leaves / 2);
        return nb;
    }
};

rt_triangle sentinel_triangle() {
    rt_triangle t; memset(&t, 0, sizeof t);
    for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) t.v[k].e[a] = INFINITY;
    return t;
}

}  // namespace

extern "C" {

rt_host_mesh* rtBuildBvh(const rt_triangle* tris, int num_tris, int nppl) {
    return rtBuildBvhLevels(tris, num_tris, nppl, 1);
}

// extra_levels: tree levels beyond the smallest complete tree that holds the triangles at nppl per leaf.  Every extra level doubles
// the leaf slots, which the SAH cut uses: on the procedural staircase (36 k triangles, nppl 5) one extra level takes a path sample
// from 184 node visits + 43 triangle tests to 152 + 22, a second one to 147 + 15 (CPU oracle counters); the first is worth 5-7 % of
// frame time on the GPU, the second nothing - hence the default of 1.
rt_host_mesh* rtBuildBvhLevels(const rt_triangle* tris, int num_tris, int nppl, int extra_levels) {
    if (!tris || num_tris <= 0 || nppl <= 0 || extra_levels < 0 || extra_levels > 8) return nullptr;
    std::vector<rt_triangle> in(tris, tris + num_tris);
    int leaves = 2;                                   // at least one internal node: root 1 + leaves 2,3
    while ((long long)leaves * nppl < num_tris) leaves *= 2;
    for (int k = 0; k < extra_levels && leaves < (1 << 30); k++) leaves *= 2;
    if (leaves > (1 << 30)) return nullptr;           // bit-stack depth limit of the traversal (32 bits)

    rt_host_mesh* m = new rt_host_mesh();
    m->nppl = nppl;
    m->tris.assign((size_t)leaves * nppl, sentinel_triangle());
    m->bvh.resize((size_t)2 * leaves);
    memset(m->bvh.data(), 0, sizeof(rt_bvh_node));

    Builder b(in);
    b.out = m; b.numLeaves = leaves; b.nppl = nppl;
    b.boxes.resize(num_tris);
    for (int a = 0; a < 3; a++) b.cent[a].resize(num_tris);
    b.order.resize(num_tris);
    for (int i = 0; i < num_tris; i++) {
        b.boxes[i] = tri_box(in[i]);
        for (int a = 0; a < 3; a++) b.cent[a][i] = 0.5f * (b.boxes[i].lo[a] + b.boxes[i].hi[a]);
        b.order[i] = i;
    }
    const Box root = b.build(1, 0, num_tris, leaves);
    for (int a = 0; a < 3; a++) { m->bounds.min.e[a] = root.lo[a]; m->bounds.max.e[a] = root.hi[a]; }
    return m;
}

static const char kBvhHeader[] = "BVH_00.04";        // staircase_scene.h:78, written with its NUL

// staircase_scene.h:75-101
rt_host_mesh* rtLoadBvhFile(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return nullptr;
    rt_host_mesh* m = new rt_host_mesh();
    char header[sizeof kBvhHeader];
    int numTris = 0, numNodes = 0;
    bool ok = fread(header, 1, sizeof header, f) == sizeof header && memcmp(header, kBvhHeader, sizeof header) == 0;
    ok = ok && fread(&numTris, sizeof(int), 1, f) == 1 && numTris > 0;
    if (ok) { m->tris.resize(numTris); ok = fread(m->tris.data(), sizeof(rt_triangle), numTris, f) == (size_t)numTris; }
    ok = ok && fread(&numNodes, sizeof(int), 1, f) == 1 && numNodes >= 4;
    if (ok) { m->bvh.resize(numNodes); ok = fread(m->bvh.data(), sizeof(rt_bvh_node), numNodes, f) == (size_t)numNodes; }
    ok = ok && fread(&m->bounds.min, sizeof(rt_vec3), 1, f) == 1 && fread(&m->bounds.max, sizeof(rt_vec3), 1, f) == 1;
    ok = ok && fread(&m->nppl, sizeof(int), 1, f) == 1 && m->nppl > 0;
    fclose(f);
    // the traversal indexes tris[(leaf - numNodes/2) * nppl + i]: refuse files it would read out of bounds on
    if (ok && (long long)(numNodes / 2) * m->nppl > numTris) ok = false;
    if (!ok) { delete m; return nullptr; }
    return m;
}

int rtSaveBvhFile(const rt_host_mesh* m, const char* path) {
    if (!m) return -1;
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    const int numTris = (int)m->tris.size(), numNodes = (int)m->bvh.size();
    bool ok = fwrite(kBvhHeader, 1, sizeof kBvhHeader, f) == sizeof kBvhHeader;
    ok = ok && fwrite(&numTris, sizeof(int), 1, f) == 1;
    ok = ok && fwrite(m->tris.data(), sizeof(rt_triangle), numTris, f) == (size_t)numTris;
    ok = ok && fwrite(&numNodes, sizeof(int), 1, f) == 1;
    ok = ok && fwrite(m->bvh.data(), sizeof(rt_bvh_node), numNodes, f) == (size_t)numNodes;
    ok = ok && fwrite(&m->bounds.min, sizeof(rt_vec3), 1, f) == 1 && fwrite(&m->bounds.max, sizeof(rt_vec3), 1, f) == 1;
    ok = ok && fwrite(&m->nppl, sizeof(int), 1, f) == 1;
    fclose(f);
    return ok ? 0 : -1;
}

void rtFreeMesh(rt_host_mesh* m) { delete m; }

int rtMeshView(const rt_host_mesh* m, rt_mesh* out) {
    if (!m || !out) return 0;
    out->tris = const_cast<rt_triangle*>(m->tris.data());
    out->numTris = (uint32_t)m->tris.size();
    out->bvh = const_cast<rt_bvh_node*>(m->bvh.data());
    out->numBvhNodes = (int32_t)m->bvh.size();
    out->bounds = m->bounds;
    return m->nppl;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Procedural staircase
// ------------------------------------------------------------------------------------------------
namespace {

struct P3 { float x, y, z; };

struct MeshOut {
    rt_triangle* tris; int cap; int n;
    void tri(P3 a, P3 b, P3 c, unsigned char mat, float uvscale) {
        if (n < cap) {
            rt_triangle t; memset(&t, 0, sizeof t);
            const P3 p[3] = { a, b, c };
            // planar texture coordinates from the two dominant axes of the face
            const float ex1 = b.x - a.x, ey1 = b.y - a.y, ez1 = b.z - a.z, ex2 = c.x - a.x, ey2 = c.y - a.y, ez2 = c.z - a.z;
            const float nx = std::fabs(ey1 * ez2 - ez1 * ey2), ny = std::fabs(ez1 * ex2 - ex1 * ez2), nz = std::fabs(ex1 * ey2 - ey1 * ex2);
            for (int k = 0; k < 3; k++) {
                t.v[k].e[0] = p[k].x; t.v[k].e[1] = p[k].y; t.v[k].e[2] = p[k].z;
                float u, v;
                if (ny >= nx && ny >= nz) { u = p[k].x; v = p[k].z; }
                else if (nx >= nz)        { u = p[k].z; v = p[k].y; }
                else                      { u = p[k].x; v = p[k].y; }
                t.texCoords[2 * k + 0] = u * uvscale;
                t.texCoords[2 * k + 1] = v * uvscale;
            }
            t.meshID = mat;
            tris[n] = t;
        }
        n++;
    }
    // quad a-b-c-d (counter-clockwise) split into `div` x `div` cells
    void quad(P3 a, P3 b, P3 c, P3 d, unsigned char mat, int div, float uvscale) {
        auto lerp = [](P3 p, P3 q, float t) { return P3{ p.x + (q.x - p.x) * t, p.y + (q.y - p.y) * t, p.z + (q.z - p.z) * t }; };
        for (int i = 0; i < div; i++)
            for (int j = 0; j < div; j++) {
                const float s0 = (float)i / div, s1 = (float)(i + 1) / div, t0 = (float)j / div, t1 = (float)(j + 1) / div;
                const P3 p00 = lerp(lerp(a, b, s0), lerp(d, c, s0), t0), p10 = lerp(lerp(a, b, s1), lerp(d, c, s1), t0);
                const P3 p11 = lerp(lerp(a, b, s1), lerp(d, c, s1), t1), p01 = lerp(lerp(a, b, s0), lerp(d, c, s0), t1);
                tri(p00, p10, p11, mat, uvscale);
                tri(p00, p11, p01, mat, uvscale);
            }
    }
    void box(P3 lo, P3 hi, unsigned char mat, int div, float uvscale) {
        const P3 v000{ lo.x, lo.y, lo.z }, v100{ hi.x, lo.y, lo.z }, v010{ lo.x, hi.y, lo.z }, v110{ hi.x, hi.y, lo.z };
        const P3 v001{ lo.x, lo.y, hi.z }, v101{ hi.x, lo.y, hi.z }, v011{ lo.x, hi.y, hi.z }, v111{ hi.x, hi.y, hi.z };
        quad(v001, v101, v111, v011, mat, div, uvscale);   // +z
        quad(v100, v000, v010, v110, mat, div, uvscale);   // -z
        quad(v101, v100, v110, v111, mat, div, uvscale);   // +x
        quad(v000, v001, v011, v010, mat, div, uvscale);   // -x
        quad(v011, v111, v110, v010, mat, div, uvscale);   // +y
        quad(v000, v100, v101, v001, mat, div, uvscale);   // -y
    }
    void uv_sphere(P3 c, float r, unsigned char mat, int seg) {
        const int rings = seg, sectors = 2 * seg;
        auto pt = [&](int i, int j) {
            const float th = (float)M_PI * (float)i / rings, ph = 2.0f * (float)M_PI * (float)j / sectors;
            return P3{ c.x + r * std::sin(th) * std::cos(ph), c.y + r * std::cos(th), c.z + r * std::sin(th) * std::sin(ph) };
        };
        for (int i = 0; i < rings; i++)
            for (int j = 0; j < sectors; j++) {
                const P3 a = pt(i, j), b = pt(i + 1, j), cc = pt(i + 1, j + 1), d = pt(i, j + 1);
                if (i != 0) tri(a, cc, d, mat, 0.01f);
                if (i != rings - 1) tri(a, b, cc, mat, 0.01f);
            }
    }
};

}  // namespace

extern "C" {

// Materials as staircase_scene.h:139-158 (indices keep their meaning; texture ids dropped unless wanted).
static void staircase_materials(rt_material* m) {
    auto set = [&](int i, int type, float r, float g, float b, float param) {
        m[i].type = type; m[i].color.e[0] = r; m[i].color.e[1] = g; m[i].color.e[2] = b; m[i].param = param; m[i].texId = -1;
    };
    set(0, RT_DIFFUSE, 0.01f, 0.01f, 0.01f, 0);             // Black
    set(1, RT_METAL, 0.27f, 0.254f, 0.15f, 0.01f);          // Brass
    set(2, RT_METAL, 0.8f, 0.8f, 0.82f, 0);                 // BrushedAluminium (textured in the reference)
    set(3, RT_DIFFUSE, 1, 1, 1, 0);                         // Candles
    set(4, RT_DIFFUSE, 0.117647f, 0.054902f, 0.0666667f, 0);// ChairSeat
    set(5, RT_GLASS, 1, 1, 1, 1.45f);                       // Glass
    set(6, RT_METAL, 1.0f, 0.95f, 0.35f, 0.05f);            // Gold
    set(7, RT_DIFFUSE, 0.8f, 0.75f, 0.6f, 0);               // Lampshade
    set(8, RT_DIFFUSE, 0.578596f, 0.578596f, 0.578596f, 0); // MagnoliaPaint
    set(9, RT_DIFFUSE, 0.5f, 0.3f, 0.25f, 0);               // Painting1
    set(10, RT_DIFFUSE, 0.25f, 0.35f, 0.5f, 0);             // Painting2
    set(11, RT_DIFFUSE, 0.3f, 0.5f, 0.3f, 0);               // Painting3
    set(12, RT_METAL, 1.0f, 1.0f, 1.0f, 0.1f);              // StainlessSteel
    set(13, RT_DIFFUSE, 0.7f, 0.65f, 0.55f, 0);             // wallpaper
    set(14, RT_DIFFUSE, 0.578596f, 0.578596f, 0.578596f, 0);// whitePaint
    set(15, RT_DIFFUSE, 1, 1, 1, 0);                        // WhitePlastic
    set(16, RT_DIFFUSE, 0.4f, 0.25f, 0.15f, 0);             // WoodChair
    set(17, RT_DIFFUSE, 0.45f, 0.3f, 0.18f, 0);             // woodFloor
    set(18, RT_DIFFUSE, 0.4f, 0.25f, 0.15f, 0);             // WoodLamp
    set(19, RT_DIFFUSE, 0.35f, 0.22f, 0.12f, 0);            // woodstairs
}

int rtSceneStaircaseProcedural(int detail, rt_triangle* tris, int cap, rt_material* materials20) {
    if (detail < 1) detail = 1;
    if (materials20) staircase_materials(materials20);
    MeshOut o{ tris, cap, 0 };
    const int q = 2 * detail;              // cells per quad edge
    const int seg = 8 * detail;            // sphere rings
    // Stairwell open to the sky: camera sits at z=494.5 looking down -z (staircase_scene.h:63-64), the light is
    // high above and behind the back wall (kernels.cu:93) so it shines in through the open top.
    const float X0 = -160, X1 = 170, Z0 = -120, Z1 = 520, H = 420;
    o.quad({ X0, 0, Z1 }, { X1, 0, Z1 }, { X1, 0, Z0 }, { X0, 0, Z0 }, 17, 4 * q, 0.01f);          // floor
    o.quad({ X0, 0, Z0 }, { X1, 0, Z0 }, { X1, H, Z0 }, { X0, H, Z0 }, 13, 2 * q, 0.01f);          // back wall
    o.quad({ X0, 0, Z1 }, { X0, 0, Z0 }, { X0, H, Z0 }, { X0, H, Z1 }, 8, 2 * q, 0.01f);           // left wall
    o.quad({ X1, 0, Z0 }, { X1, 0, Z1 }, { X1, H, Z1 }, { X1, H, Z0 }, 14, 2 * q, 0.01f);          // right wall
    o.quad({ X1, 0, Z1 }, { X0, 0, Z1 }, { X0, H, Z1 }, { X1, H, Z1 }, 8, 2 * q, 0.01f);           // wall behind the camera
    // flight of 14 steps rising towards -z on the left half
    const int steps = 14;
    for (int s = 0; s < steps; s++) {
        const float z1 = 330.0f - 28.0f * s, z0 = z1 - 28.0f, y1 = 18.0f * (s + 1);
        o.box({ X0, 0, z0 }, { 10, y1, z1 }, 19, q, 0.02f);
    }
    // landing + banister posts
    o.box({ X0, 0, Z0 }, { 10, 18.0f * steps, 330.0f - 28.0f * steps }, 19, q, 0.02f);
    for (int s = 0; s < steps; s += 2) {
        const float z = 316.0f - 28.0f * s, y = 18.0f * (s + 1);
        o.box({ 4, y, z - 3 }, { 10, y + 70, z + 3 }, 16, 1, 0.05f);
    }
    // paintings on the right wall, furniture
    o.box({ X1 - 3, 150, 120 }, { X1 - 1, 260, 220 }, 9, 1, 0.01f);
    o.box({ X1 - 3, 150, 260 }, { X1 - 1, 260, 360 }, 10, 1, 0.01f);
    o.box({ X0 + 1, 260, 40 }, { X0 + 3, 360, 140 }, 11, 1, 0.01f);
    o.box({ 60, 0, 150 }, { 130, 60, 220 }, 4, q, 0.02f);                                            // chair seat block
    o.box({ 60, 60, 150 }, { 66, 140, 220 }, 16, 1, 0.02f);
    o.box({ 90, 0, 300 }, { 96, 110, 306 }, 18, 1, 0.05f);                                           // lamp post
    o.box({ 70, 110, 280 }, { 116, 150, 326 }, 7, 1, 0.02f);                                         // lampshade
    // spheres: glass, gold, steel, brass
    o.uv_sphere({ 40, 100, 360 }, 34, 5, seg);
    o.uv_sphere({ 95, 92, 185 }, 32, 6, seg);
    o.uv_sphere({ -60, 18.0f * 6 + 30, 330.0f - 28.0f * 5 - 14 }, 30, 12, seg);
    o.uv_sphere({ 120, 24, 400 }, 24, 1, seg);
    o.uv_sphere({ -20, 24, 420 }, 24, 15, seg);
    return o.n <= cap ? o.n : -o.n;
}

}  // extern "C"
