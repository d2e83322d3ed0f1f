"""CPU: our restatement (oracle/liboracle.so) against the reference's own header-only code compiled from
/root/reference (oracle/_ref/libref.so), function by function and on whole frames, on FRESH random inputs
(the committed golden vectors are the same comparison frozen).  Skipped where the shim was never built."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/libref.so absent (built only where /root/reference exists)")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def test_struct_layouts_match_reference(rt):
    ref = O.load_ref()
    sizes = (C.c_int * 32)()
    n = ref.ref_struct_sizes(sizes, 32)
    names = ["vec3", "ray", "camera", "sphere", "plane", "bbox", "triangle", "bvh_node", "material", "stexture", "mesh", "scene",
             "kernel_scene", "intersection", "scatter_info", "path", "tri_hit"]
    got = dict(zip(names, sizes[:n]))
    # SURVEY.md §8b table
    assert got == {"vec3": 12, "ray": 24, "camera": 88, "sphere": 16, "plane": 24, "bbox": 24, "triangle": 64, "bvh_node": 24,
                   "material": 24, "stexture": 16, "mesh": 56, "scene": 96, "kernel_scene": 64, "intersection": 56,
                   "scatter_info": 36, "path": 88, "tri_hit": 12}
    for nm in ("vec3", "camera", "sphere", "plane", "bbox", "triangle", "bvh_node", "material", "stexture", "mesh", "kernel_scene"):
        assert C.sizeof(getattr(rt, nm)) == got[nm], nm


def test_functions_random(rt):
    ref = O.load_ref(); orc = O.load_oracle()
    rng = np.random.default_rng(99)
    N = 1500
    o1 = (C.c_float * 3)(); o2 = (C.c_float * 3)()
    for k in range(N):
        s = int(rng.integers(1, 2 ** 32)) | 1
        a = C.c_uint32(s); b = C.c_uint32(s)
        orc.orc_random_in_unit_sphere(C.byref(a), o1); ref.ref_random_in_unit_sphere(C.byref(b), o2)
        assert o1[:] == o2[:] and a.value == b.value
        org = rng.uniform(-3, 3, 3); d = rng.normal(size=3)
        sp = rt.sphere(); sp.center.e[:] = rng.uniform(-2, 2, 3); sp.radius = rng.uniform(0.1, 2.5)
        t1 = orc.orc_sphere_hit(C.byref(sp), f3(org), f3(d), 0.001, 3.0e38); t2 = ref.ref_sphere_hit(C.byref(sp), f3(org), f3(d), 0.001, 3.0e38)
        assert np.float32(t1).view(np.uint32) == np.float32(t2).view(np.uint32)
        lo = rng.uniform(-2, 1, 3); hi = lo + rng.uniform(0, 2, 3)
        if k % 7 == 0:
            d[k % 3] = 0.0
        assert np.float32(orc.orc_hit_bbox_dist(f3(lo), f3(hi), f3(org), f3(d), 10.0)).view(np.uint32) == \
            np.float32(ref.ref_hit_bbox_dist(f3(lo), f3(hi), f3(org), f3(d), 10.0)).view(np.uint32)
        assert orc.orc_hit_bbox(f3(lo), f3(hi), f3(org), f3(d), 10.0) == ref.ref_hit_bbox(f3(lo), f3(hi), f3(org), f3(d), 10.0)
        tri = rt.triangle()
        for q in range(3):
            tri.v[q].e[:] = rng.uniform(-2, 2, 3)
        target = sum(np.array(tri.v[q].e[:]) * w for q, w in enumerate(rng.dirichlet([1, 1, 1])))
        dd = target - org
        u1 = C.c_float(); v1 = C.c_float(); u2 = C.c_float(); v2 = C.c_float()
        t1 = orc.orc_triangle_hit(C.byref(tri), f3(org), f3(dd), 0.01, 3.0e38, C.byref(u1), C.byref(v1))
        t2 = ref.ref_triangle_hit(C.byref(tri), f3(org), f3(dd), 0.01, 3.0e38, C.byref(u2), C.byref(v2))
        assert np.float32(t1).view(np.uint32) == np.float32(t2).view(np.uint32)
        if t1 < 1e30:
            assert u1.value == u2.value and v1.value == v2.value
        pl = rt.plane(); pl.norm.e[:] = (0.0, 1.0, 0.0); pl.point.e[:] = (0.0, float(rng.uniform(-1, 1)), 0.0)
        assert np.float32(orc.orc_plane_hit(C.byref(pl), f3(org), f3(d), 0.01, 3.0e38)).view(np.uint32) == \
            np.float32(ref.ref_plane_hit(C.byref(pl), f3(org), f3(d), 0.01, 3.0e38)).view(np.uint32)


def test_light_sampling_expressions(rt):
    """kernels.cu:378-387: the two vec3 expressions of generateShadowRay, restated in rt_oracle.c, against the
    same expressions evaluated with the reference's own vec3 operators (tests/golden/light.npz)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "light.npz"))
    f = np.float32
    for k in range(len(g["eps2"])):
        phi = f(2 * np.pi * np.float64(g["eps2"][k]))
        assert phi == g["phi"][k]
        su, sv, sw = g["su"][k], g["sv"][k], g["sw"][k]
        c, s = f(np.cos(phi, dtype=f)), f(np.sin(phi, dtype=f))
        # numpy's float32 cos/sin may differ from glibc's cosf/sinf in the last ulp; compare loosely here —
        # the oracle itself calls cosf/sinf like the reference and is frame-checked in test_oracle_golden
        l = g["sinA"][k] * (c * su) + g["sinA"][k] * (s * sv) + g["cosA"][k] * sw
        assert np.allclose(l, g["ldir"][k], rtol=1e-5, atol=1e-6)
        omega = f(2 * np.pi * np.float64(f(1.0) - g["cosAMax"][k]))
        con = (omega * (g["dotl"][k] * (g["att"][k] * f(20.0)))) / f(np.pi)
        assert np.array_equal(_bits(con.astype(f)), _bits(g["lcon"][k]))


@pytest.mark.parametrize("scene,nx,ny,ns,kw", [("c1", 160, 80, 3, {}), ("rs", 120, 80, 3, {}), ("rs", 64, 48, 4, {"rr": 1, "sky": 0}),
                                               ("rs", 64, 48, 4, {"rng": 1}), ("c1", 97, 33, 2, {"t_min": 0.01})])
def test_frames_random_configs(rt, scene, nx, ny, ns, kw):
    sp, mt, cam = rt.scene_three_spheres(nx, ny) if scene == "c1" else rt.scene_random_spheres(nx, ny)
    opt = O.default_options(True)
    for k, v in kw.items():
        setattr(opt, k, v)
    a, ca = O.render(O.sphere_scene(sp, mt), cam, opt, nx, ny, ns, 50, counters=True)
    b, cb = O.ref_render_spheres(sp, mt, cam, opt, nx, ny, ns, 50, counters=True)
    assert np.array_equal(_bits(a), _bits(b))
    assert (ca.rays, ca.prim_tests, ca.hits) == (cb.rays, cb.prim_tests, cb.hits)


# ---- the MESH path: oracle restatement vs the reference-arithmetic twin (oracle/ref_driver.cpp ref_render_mesh) ------------

def _triangle_soup(rt, rng, n, n_mats=6):
    tris = np.zeros(n, rt.triangle_dtype)
    c = rng.uniform(-2, 2, (n, 1, 3)).astype(np.float32)
    tris["v"] = c + rng.uniform(-0.7, 0.7, (n, 3, 3)).astype(np.float32)
    tris["texCoords"] = rng.uniform(-1, 2, (n, 6))
    tris["meshID"] = rng.integers(0, n_mats, n)
    mats = np.zeros(n_mats, rt.material_dtype)
    mats["type"] = [rt.RT_DIFFUSE, rt.RT_METAL, rt.RT_GLASS, rt.RT_DIFFUSE, rt.RT_METAL, rt.RT_MODEL_COAT][:n_mats]
    mats["color"] = rng.uniform(0.2, 1, (n_mats, 3))
    mats["param"] = [0.0, 0.1, 1.5, 0.0, 0.0, 0.0][:n_mats]
    mats["texId"] = -1
    return tris, mats


def _same_counters(a, b):
    return (a.samples, a.rays, a.shadow_rays, a.prim_tests, a.node_visits, a.hits) == \
           (b.samples, b.rays, b.shadow_rays, b.prim_tests, b.node_visits, b.hits) and list(a.ref_stats) == list(b.ref_stats)


@pytest.mark.parametrize("kw", [{}, {"nee": 0}, {"nee": 0, "rr": 0}, {"sky": 1}, {"floor": 1}, {"floor": 1, "nee": 0}, {"rng": 1},
                                {"light_radius": 900.0}])
def test_mesh_frames_oracle_equals_reference_twin_staircase(rt, kw):
    """Frames + every counter, NEE on and off, RR on and off, gradient sky, the floor call site re-enabled (kernels.cu:341-345),
    counter RNG, and a light so big that cosAMax is NaN for part of the scene (kernels.cu:371-372)."""
    tris, mats = rt.scene_staircase_procedural(1)
    hm = rt.HostMesh.build(tris, 5)
    nx, ny, ns = 64, 80, 2
    cam = rt.staircase_camera(nx, ny)
    opt = O.default_options(False)
    kw = dict(kw)
    if "light_radius" in kw:
        opt.light.radius = kw.pop("light_radius")
    for k, v in kw.items():
        setattr(opt, k, v)
    lo = np.array(hm.view.bounds.min.e[:])
    floor = (0.0, 1.0, 0.0, 0.0, float(lo[1]) - 5.0, 0.0)                 # plane(point, norm): norm first (helper_structs.h:165-171)
    sc = O.mesh_scene(hm, mats, floor=floor)
    a, ca = O.render(sc, cam, opt, nx, ny, ns, 64, counters=True)
    b, cb = O.ref_render_mesh(sc, cam, opt, nx, ny, ns, 64, counters=True)
    assert np.array_equal(_bits(a), _bits(b)), np.count_nonzero(_bits(a) != _bits(b))
    assert _same_counters(ca, cb)
    assert ca.ref_stats[rt.RT_STAT_PRIMARY] == nx * ny * ns and ca.ref_stats[rt.RT_STAT_PRIMARY] + ca.ref_stats[rt.RT_STAT_SECONDARY] == ca.rays
    if opt.nee:
        assert ca.ref_stats[rt.RT_STAT_SHADOWS] == ca.shadow_rays > 0


@pytest.mark.parametrize("n,nppl,nee,floor", [(1, 1, 1, 0), (7, 5, 1, 1), (65, 3, 0, 1), (1000, 5, 1, 0), (1000, 16, 0, 0), (300, 20, 1, 1)])
def test_mesh_frames_oracle_equals_reference_twin_soups(rt, n, nppl, nee, floor):
    rng = np.random.default_rng(777 + 13 * n + nppl)
    tris, mats = _triangle_soup(rt, rng, n)
    tex = [rng.uniform(0, 1, (9, 13, 3)).astype(np.float32)]
    mats["texId"][3] = 0
    hm = rt.HostMesh.build(tris, nppl)
    nx, ny, ns = 56, 40, 3
    cam = rt.make_camera((4.5, 2.5, 6.0), (0, 0, 0), (0, 1, 0), 40.0, nx / ny, 0.02, 8.0)
    opt = O.default_options(False)
    opt.nee = nee
    opt.light.center.e[:] = (3.0, 9.0, 2.0); opt.light.radius = 1.5       # a light the soup can actually shadow
    opt.floor = floor
    sc = O.mesh_scene(hm, mats, tex, floor=(0.0, 1.0, 0.0, 0.0, -3.0, 0.0))   # plane {norm, point} (helper_structs.h:165-171)
    a, ca = O.render(sc, cam, opt, nx, ny, ns, 12, counters=True)
    b, cb = O.ref_render_mesh(sc, cam, opt, nx, ny, ns, 12, counters=True)
    assert np.array_equal(_bits(a), _bits(b)), np.count_nonzero(_bits(a) != _bits(b))
    assert _same_counters(ca, cb)
    if floor:       # primary rays that hit only the floor count as "primary nohit" (kernels.cu:430); paths leaving the floor into the sky as "secondary no hit"
        assert ca.ref_stats[rt.RT_STAT_SECONDARY_NOHIT] > 0 and ca.ref_stats[rt.RT_STAT_PRIMARY_NOHITS] > 0
    else:
        assert ca.ref_stats[rt.RT_STAT_SECONDARY_NOHIT] == 0


def test_generate_shadow_ray_oracle_equals_reference_twin(rt):
    """generateShadowRay as a whole (kernels.cu:363-393) on tabulated (origin, attenuation, normal, rng): every output bit-equal,
    incl. the NaN early-out before any draw and the dotl <= 0 rejection after two draws."""
    rng = np.random.default_rng(5)
    opt = O.default_options(False)
    n_gen = n_nan = n_rej = 0
    for k in range(3000):
        if k % 10 == 9:                                   # inside the light's sphere: cosAMax = NaN
            org = np.array(opt.light.center.e[:]) + rng.normal(size=3) * 20.0
        else:
            org = rng.uniform(-300, 300, 3) + (0, 100, 0)
        att = rng.uniform(0, 1, 3)
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        if k % 3 == 0:                                    # facing the light more often than not
            nrm = np.array(opt.light.center.e[:]) - org; nrm /= np.linalg.norm(nrm)
        seed = int(rng.integers(1, 2 ** 32)) | 1
        ra = O.generate_shadow_ray(opt, org, att, nrm, seed, "orc")
        rb = O.generate_shadow_ray(opt, org, att, nrm, seed, "ref")
        assert ra[0] == rb[0] and ra[5] == rb[5] and ra[6] == rb[6], k
        assert np.array_equal(_bits(ra[4]), _bits(rb[4]))                                     # cosAMax (NaN included: same bits)
        if ra[0]:
            n_gen += 1
            assert np.array_equal(_bits(ra[1]), _bits(rb[1])) and np.array_equal(_bits(ra[2]), _bits(rb[2])) and _bits(ra[3]) == _bits(rb[3]), k
        elif ra[5] == 0:
            n_nan += 1
            assert np.isnan(ra[4])
        else:
            n_rej += 1
            assert ra[5] == 2
    assert n_gen > 500 and n_nan > 100 and n_rej > 100


# ---- the host I/O either side of the path, against the reference's own code -----------------------------------------

def test_bvh_file_written_here_loads_through_the_reference_loader(rt, tmp_path):
    """rtSaveBvhFile -> the reference's loadBVH (staircase_scene.h:75-101): triangles, nodes, bounds and the leaf size come back byte for byte;
    rtLoadBvhFile of the same file returns what loadBVH returns; both refuse a wrong header and a missing file."""
    cases = [(rt.scene_staircase_procedural(1)[0], 5, None), (rt.scene_staircase_procedural(1)[0][:333], 3, 2)]
    rng = np.random.default_rng(9)
    soup = np.zeros(57, rt.triangle_dtype)
    soup["v"] = rng.uniform(-3, 3, (57, 3, 3)).astype(np.float32)
    soup["meshID"] = rng.integers(0, 20, 57)
    cases.append((soup, 1, None))
    for k, (tris, nppl, levels) in enumerate(cases):
        hm = rt.HostMesh.build(tris, nppl, levels)
        path = str(tmp_path / f"m{k}.bvh")
        assert hm.save(path) == 0
        got = O.ref_load_bvh(path)
        assert got is not None
        rtris, rbvh, rbounds, rnppl = got
        assert rnppl == nppl == hm.nppl
        assert rtris.tobytes() == hm.tris.tobytes() and rbvh.tobytes() == hm.bvh.tobytes()
        assert rbounds.tobytes() == bytes(hm.view.bounds)
        hm2 = rt.HostMesh.load(path)                                   # our reader on the same file: what the reference's reader returned
        assert hm2.nppl == rnppl and hm2.tris.tobytes() == rtris.tobytes() and hm2.bvh.tobytes() == rbvh.tobytes()
        assert bytes(hm2.view.bounds) == rbounds.tobytes()
        if k == 0:
            raw = open(path, "rb").read()
            bad = str(tmp_path / "bad.bvh")
            open(bad, "wb").write(b"BVH_00.03\x00" + raw[10:])
            assert O.ref_load_bvh(bad) is None
            with pytest.raises(ValueError):
                rt.HostMesh.load(bad)
        hm.close(); hm2.close()
    assert O.ref_load_bvh(str(tmp_path / "missing.bvh")) is None


def test_ppm_bytes_equal_the_reference_writer(rt, tmp_path):
    """rtWritePPM's file = the bytes the reference's writePPM (staircase_scene.h:32-43) sends to std::cout: header, row order (top row first), sRGB
    quantisation of every channel, separators - on random frames incl. negative, > 1, NaN-free extremes."""
    rng = np.random.default_rng(11)
    for (ny, nx) in ((1, 1), (7, 5), (33, 64)):
        fb = rng.uniform(-0.2, 1.4, (ny, nx, 3)).astype(np.float32)
        fb[0, 0] = (0.0, 1.0, 0.5)
        if ny > 1:
            fb[1, 0] = (1e-9, 1e9, 0.0031308)
        path = str(tmp_path / "o.ppm")
        assert rt.write_ppm(path, fb) == 0
        assert open(path, "rb").read() == O.ref_write_ppm(fb)


def test_staircase_camera_equals_the_reference_setup_camera(rt):
    """rtStaircaseCamera = setup_camera (staircase_scene.h:62-73), every one of the 22 floats, for several image sizes."""
    for (nx, ny) in ((640, 800), (1920, 1080), (160, 200), (48, 60), (1, 1), (1234, 567)):
        assert bytes(rt.staircase_camera(nx, ny)) == bytes(O.ref_setup_camera(nx, ny)), (nx, ny)


@pytest.mark.skipif(not O.have_ref_main(), reason="oracle/_ref/libref_main.so absent (built only where /root/reference exists)")
def test_ref_files_against_the_reference_main_cpp(rt, tmp_path):
    """REF_00.01 images (main.cpp:25-60): a file rtSaveReference writes loads through the reference's loadReference into the same floats, the file the
    reference's saveReference writes is byte-identical to ours and loads through rtLoadReference; a size mismatch and a missing file are refused by both."""
    rm = O.load_ref_main()
    rng = np.random.default_rng(13)
    nx, ny = 9, 6
    fb = rng.uniform(0, 1.3, (ny, nx, 3)).astype(np.float32)
    ours, theirs = str(tmp_path / "ours.ref"), str(tmp_path / "theirs.ref")
    assert rt.save_reference(ours, fb) == 0
    rm.ref_save_reference(theirs.encode(), nx, ny, fb.ctypes.data)
    assert open(ours, "rb").read() == open(theirs, "rb").read()
    back = np.zeros_like(fb)
    assert rm.ref_load_reference(ours.encode(), back.ctypes.data, nx, ny) == 0 and back.tobytes() == fb.tobytes()
    rc, back2 = rt.load_reference(theirs, nx, ny)
    assert rc == 0 and back2.tobytes() == fb.tobytes()
    assert rm.ref_load_reference(ours.encode(), back.ctypes.data, ny, nx) != 0 and rt.load_reference(ours, ny, nx)[0] != 0
    assert rm.ref_load_reference(str(tmp_path / "none.ref").encode(), back.ctypes.data, nx, ny) != 0
    bad = str(tmp_path / "bad.ref")
    open(bad, "wb").write(b"REF_00.02\x00" + open(ours, "rb").read()[10:])
    assert rm.ref_load_reference(bad.encode(), back.ctypes.data, nx, ny) != 0 and rt.load_reference(bad, nx, ny)[0] != 0
