#!/usr/bin/env python3
"""tools/sweep_scene_sizes.py — the sphere kernel's work orders by scene size: random small spheres on a ground sphere, n = 300 ... 4000
(the scene copy of a workgroup grows with n: two workgroups per CU up to ~80 KB of LDS, one beyond, the global-memory scene past 160 KB).
Prints Msamples/s of the default (cost-ordered two-dispatch) and the tile-major single dispatch, and the LDS bytes the scene needs."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_raytracing_optimized_amd as rt

NX, NY, SPP = 1200, 800, 50
_, _, cam = rt.scene_random_spheres(NX, NY)
for n in (300, 488, 700, 900, 1100, 1500, 2000, 2600, 4000):
    rng = np.random.default_rng(3)
    sp = np.zeros(n, rt.sphere_dtype); mt = np.zeros(n, rt.material_dtype)
    sp["center"] = rng.uniform(-12, 12, (n, 3)) * (1, 0, 1) + (0, 0.2, 0); sp["radius"] = 0.2
    sp["center"][0] = (0, -1000, 0); sp["radius"][0] = 1000
    mt["type"] = rng.choice([0, 0, 0, 0, 1, 2], n); mt["color"] = rng.uniform(0.2, 1, (n, 3)); mt["param"] = np.where(mt["type"] == 2, 1.5, 0.2); mt["texId"] = -1
    mt["type"][0] = 0
    res = {}
    for oname, v in (("default", 0), ("tile-major", 1 << 24), ("scattered", 2 << 24)):
        rt.initRendererSpheres(sp, mt, cam, NX, NY, 50)
        o = rt.getDefaultRenderOptions(True)
        rt.setRenderOptions(o, variant=v)
        rt.runRenderer(8, 8, 8)
        ms = []
        for _ in range(2):
            rt.runRenderer(SPP, 8, 8)
            ms.append(rt.getRenderStats().kernel_ms)
        rt.cleanupRenderer()
        res[oname] = round(NX * NY * SPP / min(ms) / 1e3)
    print(json.dumps({"spheres": n, **res}), flush=True)
