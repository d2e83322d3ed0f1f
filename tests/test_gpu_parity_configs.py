"""GPU parity at the BASELINE.json workloads themselves, configured exactly as bench.py configures them (WORKLOADS["C3"|"C4"|"C5"]):
the frames the benchmark renders are the frames that are checked.  The reference's per-pixel stream runs across all samples of a pixel
(/root/reference/kernels.cu:541-542,548), so a 1000-spp or 4096-spp frame reaches stream positions, parked-state volumes and cost lists
that no low-spp frame reaches; the oracle renders crops and single pixels of the same frames (a whole frame would take it hours).

Every test also requires the WHOLE frame to be free of NaN: the framebuffer the kernel writes (the pinned host framebuffer on the default
sphere path) is poisoned with NaN before each frame, so a pixel the work distribution lost cannot hide behind an earlier frame's value."""
import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _open(rt, name):
    """bench.HipBackend.open: the very calls the benchmark makes for this workload."""
    w = bench.WORKLOADS[name]
    b = bench.HipBackend()
    b.open(w, 0, 1, None)
    return w, b


def _check_pixels(O, sc, cam, opt, w, got, pixels, spp, fb):
    for x, y in pixels:
        x, y = int(x), int(y)
        O.render(sc, cam, opt, w["nx"], w["ny"], spp, w["depth"], region=(x, y, x + 1, y + 1), fb=fb)
        assert np.array_equal(_bits(got[y, x]), _bits(fb[y, x])), (x, y, got[y, x], fb[y, x])


def _check_crop(O, sc, cam, opt, w, got, x0, y0, cw, ch, spp, fb):
    O.render(sc, cam, opt, w["nx"], w["ny"], spp, w["depth"], region=(x0, y0, x0 + cw, y0 + ch), fb=fb)
    assert np.array_equal(_bits(got[y0:y0 + ch, x0:x0 + cw]), _bits(fb[y0:y0 + ch, x0:x0 + cw])), (x0, y0)


def test_c3_1000spp_frame_bit_exact_on_crops_and_pixels(rt, O):
    """C3: 1200x800, 488 spheres, 1000 spp (bench.py's C3 job: two dispatches, 17 cost lists, chain waves).  Oracle: three 16x4 crops (one over
    the longest chains of the frame, ~36 k rays per pixel), 150 random pixels."""
    w, b = _open(rt, "C3")
    b.step()
    got = b.image()
    st = rt.getRenderStats()
    b.close()
    assert st.samples == w["nx"] * w["ny"] * w["spp"]
    assert not np.isnan(got).any()
    sp, mt, cam = rt.scene_random_spheres(w["nx"], w["ny"])
    sc, opt = O.sphere_scene(sp, mt), O.default_options(True)
    fb = np.zeros((w["ny"], w["nx"], 3), np.float32)
    for (x0, y0) in ((732, 305), (0, 0), (1184, 796)):
        _check_crop(O, sc, cam, opt, w, got, x0, y0, 16, 4, w["spp"], fb)
    rng = np.random.default_rng(3)
    _check_pixels(O, sc, cam, opt, w, got, zip(rng.integers(0, w["nx"], 150), rng.integers(0, w["ny"], 150)), w["spp"], fb)


@pytest.mark.parametrize("name,spp", [("C2", 100), ("C5", 16)])
def test_whole_frames_equal_with_and_without_the_box_prefilter(rt, name, spp, monkeypatch):
    """Size-independent check of the cell-table prefilter (group_needs_cells): it may only drop boxes the exact box test would drop too, so the
    WHOLE benchmark frame - every pixel of 1200x800 at 100 spp, of 3840x2160 at 16 spp - must come out bit-equal with it and without it
    (RT_BOX_CELLS=0: the uniform loop over all 31 boxes), and the reference's ray statistics must agree."""
    frames, rays = {}, {}
    for cells in ("1", "0"):
        monkeypatch.setenv("RT_BOX_CELLS", cells)
        w, b = _open(rt, name)
        b.step(spp)
        frames[cells] = b.image()
        rays[cells] = b.counted(min(spp, 8))["rays"]
        b.close()
    assert not np.isnan(frames["1"]).any()
    assert np.array_equal(_bits(frames["1"]), _bits(frames["0"])), np.count_nonzero(_bits(frames["1"]) != _bits(frames["0"]))
    assert rays["1"] == rays["0"] and rays["1"] > 0


def test_c5_frame_two_dispatch_and_full_4096spp_pixels(rt, O):
    """C5: 3840x2160, 488 spheres.  (a) 16 spp - the two-dispatch path with 8.3 M parked pixel states at this size - on four crops and 300
    random pixels; (b) the benchmark frame itself, 4096 spp (3.5 s on the GPU), on 40 single pixels: the glass rim with the longest chains of
    the frame (~140 k rays per pixel), the corners, random positions."""
    w, b = _open(rt, "C5")
    sp, mt, cam = rt.scene_random_spheres(w["nx"], w["ny"])
    sc, opt = O.sphere_scene(sp, mt), O.default_options(True)
    fb = np.zeros((w["ny"], w["nx"], 3), np.float32)
    b.step(16)
    got = b.image()
    assert rt.getRenderStats().samples == w["nx"] * w["ny"] * 16 and rt.getRenderStats().num_launches == 1
    assert not np.isnan(got).any()
    for (x0, y0) in ((0, 0), (2270, 824), (3816, 2144), (1900, 1000)):
        _check_crop(O, sc, cam, opt, w, got, x0, y0, 24, 16, 16, fb)
    rng = np.random.default_rng(5)
    _check_pixels(O, sc, cam, opt, w, got, zip(rng.integers(0, w["nx"], 300), rng.integers(0, w["ny"], 300)), 16, fb)
    b.step()                                                            # the full 4096 spp
    got = b.image()
    st = rt.getRenderStats()
    b.close()
    assert st.samples == w["nx"] * w["ny"] * w["spp"]
    assert not np.isnan(got).any()
    rim = [(2278, 829), (2293, 825), (2294, 828), (2307, 826), (2288, 828), (2285, 828), (2321, 828), (2290, 829)]
    corners = [(0, 0), (w["nx"] - 1, 0), (0, w["ny"] - 1), (w["nx"] - 1, w["ny"] - 1)]
    rnd = list(zip(rng.integers(0, w["nx"], 28), rng.integers(0, w["ny"], 28)))
    _check_pixels(O, sc, cam, opt, w, got, rim + corners + rnd, w["spp"], fb)


@pytest.fixture(scope="module")
def c4(rt):
    """bench.py's C4 scene: procedural staircase at detail 4 (36,380 triangles) under the default rtBuildBvh (32,768 nodes, 5 per leaf)."""
    w = bench.WORKLOADS["C4"]
    tris, mats = rt.scene_staircase_procedural(w["detail"])
    hm = rt.HostMesh.build(tris, 5)
    assert len(tris) == 36380 and hm.view.numBvhNodes == 32768
    return w, hm, mats


def _render_c4(rt, c4, spp, **opts):
    w, hm, mats = c4
    cam = rt.staircase_camera(w["nx"], w["ny"])
    ks, keep = rt.make_kernel_scene(hm, mats)
    fb = rt.initRenderer(ks, cam, w["nx"], w["ny"], w["depth"], keepalive=keep)
    o = rt.getDefaultRenderOptions(False)
    rt.setRenderOptions(o, **opts)
    rt.runRenderer(spp, 8, 8)
    got = np.array(fb, copy=True)
    st = rt.getRenderStats()
    rt.cleanupRenderer()
    return got, st, cam


def test_c4_detail4_tree_no_nee_bit_exact(rt, O, c4):
    """C4's scene, tree, image size and depth (1920x1080, maxDepth 64, Russian roulette) with next-event estimation off: integer + IEEE fp32
    on both sides.  8 spp; frame crops, 200 random pixels and, over a crop, the ray / node-visit / triangle-test counters are the oracle's."""
    w, hm, mats = c4
    got, st, cam = _render_c4(rt, c4, 8, nee=0)
    assert st.samples == w["nx"] * w["ny"] * 8 and not np.isnan(got).any()
    o = O.default_options(False)
    o.nee = 0
    sc = O.mesh_scene(hm, mats)
    fb = np.zeros((w["ny"], w["nx"], 3), np.float32)
    for (x0, y0) in ((0, 0), (944, 532), (1888, 1064), (400, 800), (1500, 300)):
        _check_crop(O, sc, cam, o, w, got, x0, y0, 32, 16, 8, fb)
    rng = np.random.default_rng(7)
    _check_pixels(O, sc, cam, o, w, got, zip(rng.integers(0, w["nx"], 200), rng.integers(0, w["ny"], 200)), 8, fb)


def test_c4_detail4_tree_counters_on_a_small_frame(rt, O, c4):
    """The detail-4 tree (one level deeper than the smallest complete tree, 32,768 nodes: the one that no longer fits an XCD's L2) on a frame the
    oracle can render whole: image and all counters equal, NEE off."""
    w, hm, mats = c4
    nx, ny, ns = 240, 135, 4
    cam = rt.staircase_camera(nx, ny)
    ks, keep = rt.make_kernel_scene(hm, mats)
    fb = rt.initRenderer(ks, cam, nx, ny, w["depth"], keepalive=keep)
    o = rt.getDefaultRenderOptions(False)
    rt.setRenderOptions(o, nee=0, counters=1)
    rt.runRenderer(ns, 8, 8)
    got = np.array(fb, copy=True)
    st = rt.getRenderStats()
    rt.cleanupRenderer()
    oo = O.default_options(False)
    oo.nee = 0
    ref, cnt = O.render(O.mesh_scene(hm, mats), cam, oo, nx, ny, ns, w["depth"], counters=True)
    assert np.array_equal(_bits(got), _bits(ref))
    assert (st.rays, st.node_visits, st.prim_tests) == (cnt.rays, cnt.node_visits, cnt.prim_tests)


def test_c4_headline_configuration_nee_rr(rt, O, c4):
    """C4 as bench.py renders it: NEE + Russian roulette (kernels.cu HEAD defaults), 8 spp of the 1920x1080 frame.  The light sample's
    cosf / sinf (kernels.cu:378-379) are glibc's algorithm restated on the device (rt_device.h glibc_sincosf): the frame is held to
    equality on crops and random pixels like every other frame."""
    w, hm, mats = c4
    got, st, cam = _render_c4(rt, c4, 8)
    assert st.samples == w["nx"] * w["ny"] * 8 and not np.isnan(got).any()
    o = O.default_options(False)
    sc = O.mesh_scene(hm, mats)
    fb = np.zeros((w["ny"], w["nx"], 3), np.float32)
    for (x0, y0) in ((0, 0), (944, 532), (1888, 1064), (400, 800), (1500, 300)):
        _check_crop(O, sc, cam, o, w, got, x0, y0, 32, 16, 8, fb)
    rng = np.random.default_rng(9)
    _check_pixels(O, sc, cam, o, w, got, zip(rng.integers(0, w["nx"], 200), rng.integers(0, w["ny"], 200)), 8, fb)


def test_render_after_cleanup_with_device_reset(rt):
    """RT_CLEANUP_DEVICE_RESET=1 makes cleanupRenderer end with hipDeviceReset like kernels.cu:679.  Everything the renderer holds on the device -
    the parameter-block copy of the sphere kernel included - belongs to the render context and is freed before the reset, so a second
    init + run in the same process works and gives the same image (run in a child process: a reset destroys the test process's own context)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r); import cuda_raytracing_optimized_amd as rt\n"
            "imgs = []\n"
            "for k in range(3):\n"
            "    sp, mt, cam = rt.scene_random_spheres(96, 64)\n"
            "    fb = rt.initRendererSpheres(sp, mt, cam, 96, 64, 50)\n"
            "    rt.runRenderer(10, 8, 8); rt.runRenderer(10, 8, 8)\n"
            "    imgs.append(np.array(fb, copy=True)); rt.cleanupRenderer()\n"
            "assert not np.isnan(imgs[0]).any() and all(np.array_equal(imgs[0].view(np.uint32), i.view(np.uint32)) for i in imgs)\n"
            "print('OK')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, RT_CLEANUP_DEVICE_RESET="1"))
    assert r.returncode == 0 and "OK" in r.stdout, (r.returncode, r.stderr[-500:])
