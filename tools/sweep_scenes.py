#!/usr/bin/env python3
"""tools/sweep_scenes.py — is the cost-ordered two-dispatch schedule of the sphere kernel (default) ever worse than a plain single
dispatch?  Scenes: C1 (3 spheres), C2 (488 random spheres), all-diffuse / all-glass / all-metal versions of C2, 1500 random spheres.
Work orders (rt_kernels_spheres.hip): 0 = default (two-phase, cost-ordered), 1<<24 tile-major, 2<<24 scattered, 3<<24 centre-ray classes.
Prints Msamples/s per (scene, order) and the ratio default / best."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_raytracing_optimized_amd as rt

NX, NY, SPP = 1200, 800, 100
ORDERS = {"default": 0, "tile-major": 1 << 24, "scattered": 2 << 24, "centre-ray": 3 << 24}


def scenes():
    sp, mt, cam = rt.scene_three_spheres(NX, NY)
    yield "C1 three spheres", sp, mt, cam
    sp, mt, cam = rt.scene_random_spheres(NX, NY)
    yield "C2 random spheres", sp, mt, cam
    for name, typ, par in (("all diffuse", rt.RT_DIFFUSE, 0.0), ("all glass", rt.RT_GLASS, 1.5), ("all metal", rt.RT_METAL, 0.1)):
        m2 = mt.copy(); m2["type"][1:] = typ; m2["param"][1:] = par
        yield "C2 " + name, sp, m2, cam
    rng = np.random.default_rng(3)
    n = 1500
    s3 = np.zeros(n, rt.sphere_dtype); m3 = np.zeros(n, rt.material_dtype)
    s3["center"] = rng.uniform(-12, 12, (n, 3)) * (1, 0, 1) + (0, 0.2, 0); s3["radius"] = 0.2
    s3["center"][0] = (0, -1000, 0); s3["radius"][0] = 1000
    m3["type"] = rng.choice([0, 0, 0, 0, 1, 2], n); m3["color"] = rng.uniform(0.2, 1, (n, 3)); m3["param"] = np.where(m3["type"] == 2, 1.5, 0.2); m3["texId"] = -1
    m3["type"][0] = 0
    yield "1500 random spheres", s3, m3, cam


for name, sp, mt, cam in scenes():
    res = {}
    for oname, v in ORDERS.items():
        rt.initRendererSpheres(sp, mt, cam, NX, NY, 50)
        o = rt.getDefaultRenderOptions(True)
        rt.setRenderOptions(o, variant=v)
        rt.runRenderer(16, 8, 8)
        ms = []
        for _ in range(3):
            rt.runRenderer(SPP, 8, 8)
            ms.append(rt.getRenderStats().kernel_ms)
        rt.cleanupRenderer()
        res[oname] = NX * NY * SPP / min(ms) / 1e3
    best = max(res.values())
    print(json.dumps({"scene": name, **{k: round(v) for k, v in res.items()}, "default_over_best": round(res["default"] / best, 3)}), flush=True)
