#!/usr/bin/env python3
"""One-shot timings of the other BASELINE.json sphere configs on ONE GPU (C3: 1200x800x1000spp; C5 geometry
3840x2160 at a reduced and at the full 4096 spp) — for DESIGN.md; the contract bench is bench.py."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_raytracing_optimized_amd as rt

def run(nx, ny, spp, **kw):
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    rt.initRendererSpheres(sp, mt, cam, nx, ny, 50)
    o = rt.getDefaultRenderOptions(True)
    rt.setRenderOptions(o, **kw)
    rt.runRenderer(min(spp, 16), 8, 8)
    rt.runRenderer(spp, 8, 8)
    st = rt.getRenderStats()
    rt.cleanupRenderer()
    print(json.dumps({"config": f"{nx}x{ny}x{spp} {kw}", "kernel_ms": st.kernel_ms, "total_ms": st.total_ms,
                      "Msamples_per_s": nx * ny * spp / st.kernel_ms / 1e3}), flush=True)

run(1200, 800, 1000)
run(1200, 800, 1000, rng=rt.RT_RNG_COUNTER)
run(3840, 2160, 256)
if "--full" in sys.argv:
    run(3840, 2160, 4096)
