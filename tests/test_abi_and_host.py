"""CPU: the C-ABI libraries load and export every symbol the headers declare (no compute calls without a
GPU); host-side scene / BVH / harness code (librt_host.so)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    """Function names declared in a C header (after expanding the RT_PROBE_DECL macro by hand)."""
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text))
    return names


def test_renderer_exports_every_declared_symbol(rt):
    lib = C.CDLL(rt.RENDERER_LIB)
    declared = {n for n in _declared_functions("rt_api.h") if n[0].islower() or n.startswith("rt")}
    declared -= {"defined", "extern"}
    assert {"initRenderer", "runRenderer", "cleanupRenderer"} <= declared          # the reference's own three (kernels.h:6-8)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert set(rt.RENDERER_SYMBOLS) <= declared
    probe = re.findall(r"void (rtProbe[A-Za-z]+)##sfx", open(os.path.join(ROOT, "include", "rt_probe.h")).read())
    assert len(probe) == 12
    for base in probe:
        for sfx in ("_parity", "_fast"):
            assert hasattr(lib, base + sfx), base + sfx
    hdr = open(os.path.join(ROOT, "include", "rt_api.h")).read()
    assert lib.rtApiVersion() == rt.RT_API_VERSION == int(re.search(r"#define RT_API_VERSION (\d+)", hdr).group(1))
    assert lib.rtDeviceCount() >= 0             # the only entry point that is safe without a GPU


def test_abi_handshake_sizes_match_the_headers(rt, tmp_path):
    """rtStructSizes() (the library's own sizeof of every struct that crosses the C-ABI) == the ctypes mirror == what a C compiler makes of
    include/rt_types.h; load_renderer() runs the same comparison before any call that writes through a caller's struct pointer."""
    lib = C.CDLL(rt.RENDERER_LIB)
    rt.check_abi(lib)                                                   # the real mirror passes
    n = len(rt.ABI_STRUCTS)
    out = (C.c_int32 * n)()
    lib.rtStructSizes.argtypes = [C.POINTER(C.c_int32), C.c_int]
    assert lib.rtStructSizes(out, n) == n
    names = ["rt_render_options", "rt_render_stats", "rt_camera", "rt_sphere", "rt_material", "rt_triangle", "rt_bvh_node", "rt_mesh",
             "rt_kernel_scene", "rt_stexture", "rt_plane", "rt_bbox", "rt_vec3"]
    src = '#include <stdio.h>\n#include "rt_api.h"\nint main(void){' + "".join(f'printf("%d\\n", (int)sizeof({t}));' for t in names) + "return RT_SIZEOF_COUNT;}\n"
    exe = str(tmp_path / "sizes")
    r = subprocess.run(["gcc", "-std=c99", "-x", "c", "-I", os.path.join(ROOT, "include"), "-", "-o", exe], input=src.encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    r = subprocess.run([exe], capture_output=True)
    assert r.returncode == n
    c_sizes = [int(x) for x in r.stdout.split()]
    assert c_sizes == list(out) == [C.sizeof(t) for t in rt.ABI_STRUCTS]


def test_stale_mirror_is_refused_cleanly(rt, O):
    """The round-2 crash (gpurun_out/r02_t0_tests.log: a segmentation fault three tests into the mesh suite): rt_render_stats had grown by 160
    bytes and orc_counters by 144 in the C sources while one side of a binding was still of the previous generation, and getRenderStats /
    orc_render write sizeof(struct) bytes through the caller's pointer - a heap overrun that surfaces a few allocations later.  Now both
    bindings compare sizes at load time: a stale mirror is an ImportError with a message."""
    lib = C.CDLL(rt.RENDERER_LIB)

    class old_render_stats(C.Structure):                                # API 1000: before shadow_rays / box_tests / ref_stats[18]
        _fields_ = [("kernel_ms", C.c_double), ("total_ms", C.c_double), ("samples", C.c_int64), ("num_launches", C.c_int32), ("vgprs", C.c_int32),
                    ("rays", C.c_uint64), ("prim_tests", C.c_uint64), ("node_visits", C.c_uint64), ("exec_tests", C.c_uint64)]
    stale = [old_render_stats if t is rt.render_stats else t for t in rt.ABI_STRUCTS]
    with pytest.raises(ImportError, match="sizeof\\(old_render_stats\\) is 224 in the library and 64"):
        rt.check_abi(lib, structs=stale)
    old_version, rt.RT_API_VERSION = rt.RT_API_VERSION, 1001            # a mirror written against the previous API
    try:
        with pytest.raises(ImportError, match="API version 1002"):
            rt.check_abi(lib)
    finally:
        rt.RT_API_VERSION = old_version

    class old_counters(C.Structure):                                    # the oracle's counters before ref_stats[18]
        _fields_ = O.orc_counters._fields_[:-1]
    real, O.orc_counters = O.orc_counters, old_counters
    try:
        with pytest.raises(ImportError, match="differ from the Python mirror"):
            O._check_abi(C.CDLL(O.ORACLE_LIB), "orc_abi_sizes", O.ORACLE_LIB)
    finally:
        O.orc_counters = real
    O._check_abi(C.CDLL(O.ORACLE_LIB), "orc_abi_sizes", O.ORACLE_LIB)
    if O.have_ref():
        O._check_abi(C.CDLL(O.REF_LIB), "ref_abi_sizes", O.REF_LIB)


def test_host_library_exports_every_declared_symbol(rt):
    lib = C.CDLL(rt.HOST_LIB)
    declared = {n for n in _declared_functions("rt_host.h") if n.startswith("rt") and n != "rt_host_mesh"}
    assert declared == set(rt.HOST_SYMBOLS), declared ^ set(rt.HOST_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_headers_compile_as_c_and_cpp():
    for comp, std, lang in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "c++")):
        src = '#include "rt_api.h"\n#include "rt_host.h"\n#include "rt_probe.h"\nint main(void){return (int)sizeof(rt_kernel_scene) - 64;}\n'
        r = subprocess.run([comp, std, "-x", lang, "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-"], input=src.encode(),
                           capture_output=True)
        assert r.returncode == 0, r.stderr.decode()


def test_render_path_has_no_cpu_fallback(rt, tmp_path):
    """Without a GPU initRendererSpheres must terminate the process with the reference's error convention
    (message on stderr + exit code 99, kernels.cu:27-38) — never fall back to a CPU path."""
    if rt.device_count() > 0:
        pytest.skip("a GPU is present")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import cuda_raytracing_optimized_amd as rt\n"
            "sp, mt, cam = rt.scene_three_spheres(16, 8)\n"
            "rt.initRendererSpheres(sp, mt, cam, 16, 8, 5)\nprint('SHOULD NOT GET HERE')\n" % ROOT)
    r = subprocess.run(["python3", "-c", code], capture_output=True, timeout=120)
    assert r.returncode == 99, (r.returncode, r.stderr.decode()[-300:])
    assert b"HIP error" in r.stderr and b"SHOULD NOT" not in r.stdout


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cuda-raytracing-optimized_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for pat in ("import oracle", "from oracle", "liboracle", "libref.so", '#include "../../oracle', "rt_oracle.h"):
                    assert pat not in text, (f, pat)


def test_lcg_and_scene_generators(rt):
    h = rt.load_host()
    st = C.c_uint32(0)
    vals = [h.rtRandomFloat(C.byref(st)) for _ in range(3)]
    # main.cpp:17-20 by hand: state = 214013*state + 2531011 ; ((state >> 16) & 0x7FFF) / 32767
    s = 0
    exp = []
    for _ in range(3):
        s = (214013 * s + 2531011) & 0xFFFFFFFF
        exp.append(np.float32(np.float32((s >> 16) & 0x7FFF) / np.float32(32767)))
    assert [np.float32(v) for v in vals] == exp
    sp, mt, cam = rt.scene_three_spheres(400, 200)
    assert sorted(mt["type"]) == [0, 1, 2]
    sp2, mt2, _ = rt.scene_random_spheres(1200, 800, seed=0)
    sp3, mt3, _ = rt.scene_random_spheres(1200, 800, seed=0)
    assert sp2.tobytes() == sp3.tobytes() and mt2.tobytes() == mt3.tobytes()
    sp4, _, _ = rt.scene_random_spheres(1200, 800, seed=7)
    assert sp4.tobytes() != sp2.tobytes()
    small = sp2[1:485]
    assert np.all(small["radius"] == np.float32(0.2)) and np.all(small["center"][:, 1] == np.float32(0.2))
    assert small["center"][:, 0].min() >= -11 and small["center"][:, 0].max() < 11


def test_bvh_builder_layout_contract(rt):
    """kernels.cu:154-224,614: heap-indexed complete tree, numBvhNodes = 2*leaves, leaf L owns nppl triangles,
    +inf sentinels end a leaf, parents bound children, every input triangle appears exactly once."""
    tris, mats = rt.scene_staircase_procedural(1)
    assert len(mats) == 20 and len(tris) > 1000
    minimal = {}
    for nppl, extra in ((5, 0), (1, None), (5, None), (8, None), (5, 2)):
        hm = rt.HostMesh.build(tris, nppl, extra)
        v = hm.view
        leaves = v.numBvhNodes // 2
        assert v.numBvhNodes == 2 * leaves and leaves & (leaves - 1) == 0 and leaves >= 2
        if extra == 0:
            minimal[nppl] = leaves
            assert (leaves // 2) * nppl < len(tris) or leaves == 2          # the smallest complete tree that holds them
        if nppl == 5 and extra in (None, 2):                                # rtBuildBvh = one level more, rtBuildBvhLevels as asked
            assert leaves == minimal[5] << (1 if extra is None else extra)
        assert v.numTris == leaves * nppl and leaves * nppl >= len(tris)
        out = hm.tris
        real = ~np.isinf(out["v"][:, 0, 0])
        assert real.sum() == len(tris)
        a = np.sort(out[real].view(np.uint8).reshape(-1, 64), axis=0)
        b = np.sort(np.ascontiguousarray(tris).view(np.uint8).reshape(-1, 64), axis=0)
        assert np.array_equal(a, b)
        # within a leaf the real triangles come first (the traversal stops at the first sentinel)
        r = real.reshape(leaves, nppl)
        assert np.all(r[:, :-1] >= r[:, 1:])
        bvh = hm.bvh
        lo, hi = bvh["a"], bvh["b"]
        for idx in range(1, leaves):
            for ch in (2 * idx, 2 * idx + 1):
                if np.all(lo[ch] <= hi[ch]):            # non-empty child
                    assert np.all(lo[idx] <= lo[ch]) and np.all(hi[idx] >= hi[ch]), (idx, ch)
        for L in range(leaves, 2 * leaves):
            t = out[(L - leaves) * nppl:(L - leaves + 1) * nppl]
            t = t[~np.isinf(t["v"][:, 0, 0])]
            if len(t):
                assert np.all(lo[L] <= t["v"].min(axis=(0, 1))) and np.all(hi[L] >= t["v"].max(axis=(0, 1)))
        assert np.array_equal(np.array(v.bounds.min.e[:], np.float32), lo[1]) and np.array_equal(np.array(v.bounds.max.e[:], np.float32), hi[1])
        hm.close()


def test_an_extra_tree_level_saves_traversal_work(rt, O):
    """rtBuildBvhLevels: more leaf slots for the SAH cuts -> fewer node visits and triangle tests for the same rays (the oracle's counters)."""
    tris, mats = rt.scene_staircase_procedural(2)
    cam = rt.staircase_camera(64, 36)
    work = []
    for extra in (0, 1):
        hm = rt.HostMesh.build(tris, 5, extra)
        _, c = O.render(O.mesh_scene(hm, mats), cam, O.default_options(False), 64, 36, 1, 64, counters=True)
        work.append((c.node_visits, c.prim_tests))
        hm.close()
    assert work[1][0] < work[0][0] and work[1][1] < work[0][1], work


def test_bvh_file_round_trip(rt, tmp_path):
    """BVH_00.04 (staircase_scene.h:75-101): header with NUL, int numTris, triangles, int numBvhNodes, nodes, min, max, int nppl."""
    tris, _ = rt.scene_staircase_procedural(1)
    hm = rt.HostMesh.build(tris, 5)
    path = str(tmp_path / "s.bvh")
    assert hm.save(path) == 0
    raw = open(path, "rb").read()
    assert raw[:10] == b"BVH_00.04\x00"
    nt = int(np.frombuffer(raw[10:14], np.int32)[0])
    assert nt == hm.view.numTris
    off = 14 + nt * 64
    nn = int(np.frombuffer(raw[off:off + 4], np.int32)[0])
    assert nn == hm.view.numBvhNodes
    assert len(raw) == off + 4 + nn * 24 + 24 + 4
    assert int(np.frombuffer(raw[-4:], np.int32)[0]) == 5
    hm2 = rt.HostMesh.load(path)
    assert hm2.nppl == 5 and hm2.tris.tobytes() == hm.tris.tobytes() and hm2.bvh.tobytes() == hm.bvh.tobytes()
    open(str(tmp_path / "bad.bvh"), "wb").write(b"BVH_00.03\x00" + raw[10:])
    with pytest.raises(ValueError):
        rt.HostMesh.load(str(tmp_path / "bad.bvh"))
    with pytest.raises(ValueError):
        rt.HostMesh.load(str(tmp_path / "missing.bvh"))


def test_output_harness(rt, tmp_path):
    """PPM (staircase_scene.h:32-43), REF_00.01 (main.cpp:25-60), RMSE (main.cpp:117-125)."""
    rng = np.random.default_rng(5)
    fb = rng.uniform(0, 1.2, (6, 5, 3)).astype(np.float32)
    ppm = str(tmp_path / "o.ppm")
    assert rt.write_ppm(ppm, fb) == 0
    lines = open(ppm).read().split("\n")
    assert lines[0] == "P3" and lines[1] == "5 6" and lines[2] == "255"
    h = rt.load_host()
    first = [h.rtLinearToSRGB(float(x)) for x in fb[5, 0]]          # top row first: j = ny-1
    assert lines[3] == "%d %d %d" % tuple(first)
    assert len([ln for ln in lines[3:] if ln]) == 30
    ref = str(tmp_path / "f5-6.ref")
    assert rt.save_reference(ref, fb) == 0
    raw = open(ref, "rb").read()
    assert raw[:10] == b"REF_00.01\x00" and raw[10:18] == np.array([5, 6], np.int32).tobytes() and len(raw) == 18 + 5 * 6 * 12
    rc, back = rt.load_reference(ref, 5, 6)
    assert rc == 0 and back.tobytes() == fb.tobytes()
    assert rt.load_reference(ref, 6, 5)[0] == -2 and rt.load_reference(str(tmp_path / "nope.ref"), 5, 6)[0] == -1
    g = fb + np.float32(0.01)
    exp = np.sqrt(((fb.astype(np.float64) - g.astype(np.float64)) ** 2 / 3.0).sum() / 30)
    assert abs(rt.rmse(fb, g) - exp) < 1e-7
    assert rt.rmse(fb, fb) == 0.0


def test_bench_reports_traffic_only_for_the_kernel_sources_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic comes from profiles/traffic.json (rocprofv3 FETCH_SIZE / WRITE_SIZE passes, tools/measure_traffic.sh), which
    records the hash of the kernel sources it was taken on: a changed kernel must not inherit a stale figure."""
    import json
    import bench
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: h)
    assert bench.committed_traffic("C2", 100.0)["traffic"] is None                     # no file
    json.dump({"kernel_source_hash": h, "C2": {"bytes_per_frame": 300.0}}, open(prof / "traffic.json", "w"))
    t = bench.committed_traffic("C2", 100.0)
    assert t["traffic"] == 300.0 and t["traffic_over_algorithmic"] == 3.0 and "committed_profile" in t["traffic_source"]
    assert bench.committed_traffic("C4", 100.0)["traffic"] is None                     # no entry for that config
    json.dump({"kernel_source_hash": "0" * 16, "C2": {"bytes_per_frame": 300.0}}, open(prof / "traffic.json", "w"))
    assert bench.committed_traffic("C2", 100.0)["traffic"] is None                     # other kernel sources


def test_committed_traffic_matches_the_committed_kernel_sources():
    """profiles/traffic.json must have been measured on the kernel sources in this checkout (else bench.py prints traffic: null)."""
    import json
    import bench
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(p):
        pytest.skip("no committed traffic measurement")
    if json.load(open(p))["kernel_source_hash"] != bench.kernel_source_hash():
        # not an error of the tree: bench.py then prints `traffic: null` (test above); the measurement is re-taken with tools/measure_traffic.sh
        pytest.skip("profiles/traffic.json was measured on other kernel sources: bench.py reports traffic null until it is re-measured")


def test_pair_round_pretest_flags_every_positive_discriminant():
    """The sphere kernel's pair rounds FIND the spheres with a positive discriminant with fused multiply-adds and a slack (DESIGN.md 3.5, v9;
    rt_kernels_spheres.hip scan_pairs): v = (a - k) c_fma - (b_fma^2 + k0) must be negative whenever the reference's own evaluation
    (intersections.h:88-93, every product and sum rounded to fp32) has b*b - a*c > 0.  Emulated here in numpy - fp32 operations as float32
    arithmetic, a fused multiply-add as the float64 result rounded once (products of two fp32 numbers are exact in float64) - on random
    and on near-tangent rays, far and near, with the kernel's constants."""
    f32 = np.float32
    rng = np.random.default_rng(11)
    n = 400000
    k = f32(2.0 ** -18)
    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
    for dist, r_max in ((3.0, 0.3), (40.0, 0.3), (3000.0, 0.3), (1.0, 0.9)):
        r = rng.uniform(0.05, r_max, n).astype(f32)
        d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        d = d.astype(f32)
        # centres placed so that the ray (origin 0) passes at a distance of r (1 + tiny) from them: the discriminant is ~0 for half of the set
        along = rng.uniform(0.2, 1.0, n) * dist
        perp = rng.normal(size=(n, 3)); perp -= (perp * d).sum(1, keepdims=True) * d; perp /= np.linalg.norm(perp, axis=1, keepdims=True)
        tangent = rng.random(n) < 0.5
        miss = np.where(tangent, 1.0 + rng.normal(scale=3e-7, size=n), rng.uniform(0.0, 2.0, n))
        centre = (d * along[:, None] + perp * (r * miss)[:, None]).astype(f32)
        oc = (f32(0) - centre).astype(f32)                                              # o - centre, o = 0
        a = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(f32) + d[:, 2] * d[:, 2]).astype(f32)
        b = ((oc[:, 0] * d[:, 0] + oc[:, 1] * d[:, 1]).astype(f32) + oc[:, 2] * d[:, 2]).astype(f32)
        c = (((oc[:, 0] * oc[:, 0] + oc[:, 1] * oc[:, 1]).astype(f32) + oc[:, 2] * oc[:, 2]).astype(f32) - (r * r).astype(f32)).astype(f32)
        disc = ((b * b).astype(f32) - (a * c).astype(f32)).astype(f32)
        positive = disc > 0
        r2 = (r * r).astype(f32)
        bf = fma(oc[:, 2], d[:, 2], fma(oc[:, 1], d[:, 1], (oc[:, 0] * d[:, 0]).astype(f32)))
        cf = fma(oc[:, 2], oc[:, 2], fma(oc[:, 1], oc[:, 1], fma(oc[:, 0], oc[:, 0], -r2)))
        k0 = f32(2.0 * 3.814697265625e-6 * float(r_max) ** 2 * 1.0001)
        v = fma((a - k).astype(f32), cf, -fma(bf, bf, np.full(n, k0, f32)))
        assert positive.sum() > n // 10 and (~positive).sum() > n // 10
        assert np.all(v[positive] < 0), (dist, int(np.count_nonzero(v[positive] >= 0)))
        if dist <= 40.0:                                                                # (far away the slack, like the reference's own rounding noise, exceeds r^2:
            assert np.count_nonzero((v < 0) & ~positive & ~tangent) < n // 20           #  everything is flagged there) - near, away from tangency, little else is
