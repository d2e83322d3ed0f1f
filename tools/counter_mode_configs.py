#!/usr/bin/env python3
"""tools/counter_mode_configs.py — the sphere configs C3 / C5 (at 1024 spp) with the per-sample counter RNG stream, under the library RT_LIB names (default: the built one)
(the number of work items per pixel is the renderer's choice: rt_renderer.hip, `samples_per_item`)."""
import os, sys
sys.path.insert(0, os.getcwd())
import cuda_raytracing_optimized_amd as rt
import bench
lib = os.environ.get("RT_LIB", rt.RENDERER_LIB)
rt.RENDERER_LIB = os.path.abspath(lib)
if True:
    for name, spp in (("C3", 1000), ("C5", 1024)):
        w = dict(bench.WORKLOADS[name]); w["spp"] = spp
        b = bench.HipBackend()
        b.open(w, 0, 1, None)
        b.rt.setRenderOptions(b.opt, rng=b.rt.RT_RNG_COUNTER)
        b.step(8)
        ms = [b.step() for _ in range(2)]
        b.close()
        print(lib, name, spp, "counter mode Msamples/s", round(w["nx"] * w["ny"] * spp / min(ms) / 1e3, 1), "ms", round(min(ms), 1), flush=True)
