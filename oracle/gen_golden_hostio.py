#!/usr/bin/env python3
"""Mints the host-I/O fixtures under tests/golden/ from the REFERENCE'S OWN CODE (oracle/_ref/libref.so: loadBVH, writePPM, setup_camera of
staircase_scene.h; oracle/_ref/libref_main.so: saveReference of main.cpp), see oracle/ref_driver.cpp and oracle/ref_main_shim.cpp.

Run only in the build container:   python oracle/gen_golden_hostio.py
Fixtures are data:
  hostio_small.bvh   a BVH_00.04 file (57 random triangles, 1 per leaf) that the reference's loadBVH accepts ...
  hostio.npz         ... and the arrays loadBVH returned for it; a seeded 16 x 12 framebuffer; the staircase camera of setup_camera at four sizes
  hostio_small.ppm   the bytes the reference's writePPM printed for that framebuffer
  hostio_small.ref   the REF_00.01 file the reference's saveReference wrote for it
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuda_raytracing_optimized_amd as rt  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    rng = np.random.default_rng(20261005)
    soup = np.zeros(57, rt.triangle_dtype)
    soup["v"] = rng.uniform(-3, 3, (57, 3, 3)).astype(np.float32)
    soup["texCoords"] = rng.uniform(0, 1, soup["texCoords"].shape).astype(np.float32)
    soup["meshID"] = rng.integers(0, 20, 57)
    hm = rt.HostMesh.build(soup, 1)
    bvh_path = os.path.join(OUT, "hostio_small.bvh")
    assert hm.save(bvh_path) == 0
    tris, bvh, bounds, nppl = O.ref_load_bvh(bvh_path)            # what the REFERENCE reads from that file
    fb = rng.uniform(-0.1, 1.3, (12, 16, 3)).astype(np.float32)
    fb[0, 0] = (0.0, 1.0, 0.5); fb[1, 0] = (1e-9, 1e9, 0.0031308)
    open(os.path.join(OUT, "hostio_small.ppm"), "wb").write(O.ref_write_ppm(fb))
    O.load_ref_main().ref_save_reference(os.path.join(OUT, "hostio_small.ref").encode(), 16, 12, fb.ctypes.data)
    sizes = np.array([(640, 800), (1920, 1080), (48, 60), (1234, 567)], np.int32)
    cams = np.stack([np.frombuffer(bytes(O.ref_setup_camera(int(nx), int(ny))), np.float32) for nx, ny in sizes])
    np.savez_compressed(os.path.join(OUT, "hostio.npz"), tris=tris.view(np.uint8), bvh=bvh.view(np.uint8), bounds=bounds, nppl=np.int32(nppl),
                        fb=fb, cam_sizes=sizes, cams=cams)
    for f in sorted(os.listdir(OUT)):
        if f.startswith("hostio"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
