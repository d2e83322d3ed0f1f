# tools/ab_spheres3d.sh reps lib... — A/B of builds on a scene WITHOUT a shared box axis: 488 spheres of mixed radii scattered in a slab of height 4
N=$1; shift
for i in $(seq $N); do for L in "$@"; do
  RT_LIB=$L python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import cuda_raytracing_optimized_amd as rt
rt.RENDERER_LIB = os.path.abspath(os.environ["RT_LIB"])
NX, NY, SPP = 1200, 800, 50
_, _, cam = rt.scene_random_spheres(NX, NY)
n = 488
rng = np.random.default_rng(5)
sp = np.zeros(n, rt.sphere_dtype); mt = np.zeros(n, rt.material_dtype)
sp["center"] = rng.uniform(-11, 11, (n, 3)) * (1, 0, 1) + np.stack([np.zeros(n), rng.uniform(0.2, 4.0, n), np.zeros(n)], 1); sp["radius"] = rng.uniform(0.1, 0.3, n)
sp["center"][0] = (0, -1000, 0); sp["radius"][0] = 1000
mt["type"] = rng.choice([0, 0, 0, 0, 1, 2], n); mt["color"] = rng.uniform(0.2, 1, (n, 3)); mt["param"] = np.where(mt["type"] == 2, 1.5, 0.2); mt["texId"] = -1
mt["type"][0] = 0
rt.initRendererSpheres(sp, mt, cam, NX, NY, 50)
o = rt.getDefaultRenderOptions(True); rt.setRenderOptions(o)
rt.runRenderer(8, 8, 8)
ms = []
for _ in range(4):
    rt.runRenderer(SPP, 8, 8); ms.append(rt.getRenderStats().kernel_ms)
rt.cleanupRenderer()
print(os.environ["RT_LIB"], round(NX * NY * SPP / (sum(ms) / len(ms)) / 1e3, 1))
PY
done; done
