#!/usr/bin/env python3
"""tools/sweep_scene_sizes.py [plane|volume] — the sphere kernel by scene size: n = 300 ... 4000 random small spheres over a ground sphere, resting on it in a
24 x 24 field ("plane": the benchmark's shape, every group box shares its vertical extent) or scattered in the 24 x 7 x 24 volume above it ("volume": no
shared slab; the 3-axis cell tables).  The field does not grow with n, so the scene gets DENSER: a sample traces more rays - the table prints rays per
sample (device counters) and Mrays/s beside Msamples/s, which is what compares kernels across sizes.  Work orders: the default (cost-ordered two dispatches
where the launcher allows them) and the scattered single dispatch.  SWEEP_NX / SWEEP_NY / SWEEP_SPP / SWEEP_N override the frame and the sizes
(a 1200x800 frame at 50 spp ends with its slowest pixels, which get slower as the field gets denser; 2560x1440 at 16 spp is bound by throughput)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_raytracing_optimized_amd as rt

shape = sys.argv[1] if len(sys.argv) > 1 else "plane"
NX, NY, SPP = int(os.environ.get("SWEEP_NX", 1200)), int(os.environ.get("SWEEP_NY", 800)), int(os.environ.get("SWEEP_SPP", 50))
_, _, cam = rt.scene_random_spheres(NX, NY)
for n in [int(x) for x in os.environ.get("SWEEP_N", "300,488,700,900,1100,1500,2000,2600,4000").split(",")]:
    rng = np.random.default_rng(3)
    sp = np.zeros(n, rt.sphere_dtype); mt = np.zeros(n, rt.material_dtype)
    if shape == "plane":
        sp["center"] = rng.uniform(-12, 12, (n, 3)) * (1, 0, 1) + (0, 0.2, 0); sp["radius"] = 0.2
    else:
        c = rng.uniform(-12, 12, (n, 3)); c[:, 1] = rng.uniform(0.3, 7.0, n)
        sp["center"] = c; sp["radius"] = rng.uniform(0.12, 0.28, n)
    sp["center"][0] = (0, -1000, 0); sp["radius"][0] = 1000
    mt["type"] = rng.choice([0, 0, 0, 0, 1, 2], n); mt["color"] = rng.uniform(0.2, 1, (n, 3)); mt["param"] = np.where(mt["type"] == 2, 1.5, 0.2); mt["texId"] = -1
    mt["type"][0] = 0
    res = {}
    for oname, v in (("default", 0), ("scattered", 2 << 24)):
        rt.initRendererSpheres(sp, mt, cam, NX, NY, 50)
        o = rt.getDefaultRenderOptions(True)
        rt.setRenderOptions(o, variant=v)
        rt.runRenderer(8, 8, 8)
        ms = []
        for _ in range(2):
            rt.runRenderer(SPP, 8, 8)
            ms.append(rt.getRenderStats().kernel_ms)
        if oname == "default":
            rt.setRenderOptions(o, variant=v, counters=1)
            rt.runRenderer(4, 8, 8)
            st = rt.getRenderStats()
            res["rays_per_sample"] = round(st.rays / (NX * NY * 4), 3)
            res["tests_per_ray"] = round(st.exec_tests / max(st.rays, 1), 1)
            res["boxes_per_ray"] = round(getattr(st, "box_tests", 0) / max(st.rays, 1), 2)
        rt.cleanupRenderer()
        res[oname] = round(NX * NY * SPP / min(ms) / 1e3)
    res["Mrays_per_s"] = round(max(res["default"], res["scattered"]) * res["rays_per_sample"])
    print(json.dumps({"shape": shape, "spheres": n, **res}), flush=True)
