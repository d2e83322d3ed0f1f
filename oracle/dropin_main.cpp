// dropin_main.cpp — drop-in link test (TEST INFRASTRUCTURE, built only where /root/reference exists).
//
// This translation unit is compiled against the REFERENCE'S OWN kernels.h / helper_structs.h (its vec3, camera,
// kernel_scene, mesh classes) and linked against OUR librt_mi355x.so.  It is what a maintainer of the reference gets by
// replacing kernels.cu with `-lrt_mi355x`: the three extern "C" calls of main.cpp:94-101,138 resolve to our library,
// by-value class arguments and all.  The scene comes from librt_host.so (the staircase asset is not in the snapshot).
//
//   dropin_main <nx> <ny> <ns> <maxDepth> <out.raw>      writes the framebuffer (nx*ny vec3) as raw float32
#include <cstdint>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kernels.h"                    // the reference's declaration of initRenderer / runRenderer / cleanupRenderer
#include "../include/rt_host.h"        // our host-side scene helpers (C structs with the same layout)

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s nx ny ns maxDepth out.raw\n", argv[0]); return 2; }
    const int nx = atoi(argv[1]), ny = atoi(argv[2]), ns = atoi(argv[3]), maxDepth = atoi(argv[4]);

    material materials[20];
    const int need = -rtSceneStaircaseProcedural(1, nullptr, 0, reinterpret_cast<rt_material*>(materials));
    std::vector<triangle> tris(need);
    rtSceneStaircaseProcedural(1, reinterpret_cast<rt_triangle*>(tris.data()), need, reinterpret_cast<rt_material*>(materials));
    rt_host_mesh* hm = rtBuildBvh(reinterpret_cast<const rt_triangle*>(tris.data()), need, 5);
    rt_mesh view;
    const int nppl = rtMeshView(hm, &view);

    mesh* m = static_cast<mesh*>(malloc(sizeof(mesh)));      // not `new`: mesh::~mesh would delete[] memory it does not own
    m->tris = reinterpret_cast<triangle*>(view.tris);
    m->numTris = view.numTris;
    m->bvh = reinterpret_cast<bvh_node*>(view.bvh);
    m->numBvhNodes = view.numBvhNodes;
    memcpy(&m->bounds, &view.bounds, sizeof(bbox));

    kernel_scene ksc = { m, plane(), materials, 20, nullptr, 0, nppl };     // staircase_scene.h:181
    camera cam;
    rt_camera c;
    rtStaircaseCamera(nx, ny, &c);
    memcpy(&cam, &c, sizeof cam);

    vec3* fb = nullptr;
    initRenderer(ksc, cam, &fb, nx, ny, maxDepth);          // main.cpp:94
    runRenderer(ns, 8, 8);                                   // main.cpp:98
    FILE* f = fopen(argv[5], "wb");
    fwrite(fb, sizeof(vec3), (size_t)nx * ny, f);            // the host reads fb directly (main.cpp:105,119)
    fclose(f);
    cleanupRenderer();                                       // main.cpp:138
    free(m);
    rtFreeMesh(hm);
    return 0;
}
