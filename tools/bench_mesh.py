#!/usr/bin/env python3
"""Config 4 (BASELINE.json): procedural staircase mesh, 1920x1080, 256 spp, maxDepth 64, NEE + RR (HEAD defaults).
Prints Msamples/s + the BVH-path roofline inputs of SURVEY.md §8d (node visits / triangle tests / rays per sample)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cuda_raytracing_optimized_amd as rt

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=1920); ap.add_argument("--ny", type=int, default=1080)
ap.add_argument("--spp", type=int, default=256); ap.add_argument("--depth", type=int, default=64)
ap.add_argument("--detail", type=int, default=4); ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--fp", default="parity"); ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--count-spp", type=int, default=4)
ap.add_argument("--lib", default=None, help="an experimental build of librt_mi355x.so (tools only)")
a = ap.parse_args()
if a.lib:
    rt.RENDERER_LIB = os.path.abspath(a.lib)
t0 = time.time()
tris, mats = rt.scene_staircase_procedural(a.detail)
hm = rt.HostMesh.build(tris, 5)
t_build = time.time() - t0
cam = rt.staircase_camera(a.nx, a.ny)
ks, keep = rt.make_kernel_scene(hm, mats)
fb = rt.initRenderer(ks, cam, a.nx, a.ny, a.depth, keepalive=keep)
o = rt.getDefaultRenderOptions(False)
rt.setRenderOptions(o, fp=rt.RT_FP_FAST if a.fp == "fast" else rt.RT_FP_PARITY, variant=a.variant, counters=1)
rt.runRenderer(a.count_spp, 8, 8)
st = rt.getRenderStats()
smp = a.nx * a.ny * a.count_spp
per = dict(rays=st.rays / smp, nodes=st.node_visits / smp, tris=st.prim_tests / smp)
rt.setRenderOptions(o, counters=0)
ms = []
for _ in range(a.steps):
    rt.runRenderer(a.spp, 8, 8)
    ms.append(rt.getRenderStats().kernel_ms)
rt.cleanupRenderer()
k = float(np.mean(ms))
samples = a.nx * a.ny * a.spp
bytes_per_sample = 48 * per["nodes"] + 64 * per["tris"] + 64 * per["rays"]
flops_per_sample = 30 * per["nodes"] + 51 * per["tris"] + 150 * per["rays"]
print(json.dumps({"config": f"staircase-procedural detail {a.detail} ({len(tris)} tris, {hm.view.numBvhNodes} nodes) {a.nx}x{a.ny}x{a.spp} depth {a.depth} fp {a.fp} variant {a.variant}",
                  "Msamples_per_s": samples / k / 1e3, "kernel_ms": k, "bvh_build_s": t_build, **per,
                  "algorithmic_bytes_per_sample": bytes_per_sample, "gather_GBps": bytes_per_sample * samples / k / 1e6,
                  "algorithmic_flops_per_sample": flops_per_sample, "TFLOPs": flops_per_sample * samples / k / 1e9}))
