import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
import cuda_raytracing_optimized_amd as rt
NX, NY, SPP = 1200, 800, 20
_, _, cam = rt.scene_random_spheres(NX, NY)
for n in (488, 700, 1500):
    rng = np.random.default_rng(3)
    sp = np.zeros(n, rt.sphere_dtype); mt = np.zeros(n, rt.material_dtype)
    sp["center"] = rng.uniform(-12, 12, (n, 3)) * (1, 0, 1) + (0, 0.2, 0); sp["radius"] = 0.2
    sp["center"][0] = (0, -1000, 0); sp["radius"][0] = 1000
    mt["type"] = rng.choice([0, 0, 0, 0, 1, 2], n); mt["color"] = rng.uniform(0.2, 1, (n, 3)); mt["param"] = np.where(mt["type"] == 2, 1.5, 0.2); mt["texId"] = -1
    mt["type"][0] = 0
    rt.initRendererSpheres(sp, mt, cam, NX, NY, 50)
    o = rt.getDefaultRenderOptions(True)
    rt.setRenderOptions(o, counters=1)
    rt.runRenderer(SPP, 8, 8)
    st = rt.getRenderStats()
    rt.cleanupRenderer()
    print(json.dumps({"spheres": n, "rays_per_sample": st.rays / (NX * NY * SPP), "sphere_tests_per_ray": st.exec_tests / st.rays, "box_tests_per_ray": st.box_tests / st.rays}))
