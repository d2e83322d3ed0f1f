"""ctypes binding of the CPU ORACLE (oracle/liboracle.so) and, where it was built, of the
reference-header shim (oracle/_ref/libref.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(_HERE, "liboracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libref.so")
REF_MAIN_LIB = os.path.join(_HERE, "_ref", "libref_main.so")


def _pkg():
    import cuda_raytracing_optimized_amd as rt
    return rt


class orc_scene(C.Structure):
    _fields_ = [("spheres", C.c_void_p), ("sphere_materials", C.c_void_p), ("num_spheres", C.c_int32),
                ("tris", C.c_void_p), ("num_tris", C.c_int32), ("bvh", C.c_void_p), ("num_bvh_nodes", C.c_int32),
                ("bounds", C.c_float * 6), ("nppl", C.c_int32), ("materials", C.c_void_p), ("num_materials", C.c_int32),
                ("textures", C.c_void_p), ("num_textures", C.c_int32), ("floor", C.c_float * 6)]


class orc_counters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("prim_tests", C.c_uint64),
                ("node_visits", C.c_uint64), ("hits", C.c_uint64), ("rng_draws", C.c_uint64),
                ("ref_stats", C.c_uint64 * 18)]       # the reference's STATS counters (kernels.cu:47-67), indices rt.RT_STAT_*


class orc_scatter(C.Structure):
    _fields_ = [("wi", C.c_float * 3), ("specular", C.c_int32), ("throughput", C.c_float * 3),
                ("refracted", C.c_int32), ("t", C.c_float)]


_fp = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


def _bind_common(lib, pre):
    rt = _pkg()
    g = lambda n: getattr(lib, pre + n)
    g("wang_hash").argtypes = [C.c_uint32]; g("wang_hash").restype = C.c_uint32
    g("pixel_seed").argtypes = [C.c_uint32]; g("pixel_seed").restype = C.c_uint32
    g("xor_shift_32").argtypes = [_u32p]; g("xor_shift_32").restype = C.c_uint32
    g("rnd").argtypes = [_u32p]; g("rnd").restype = C.c_float
    g("random_in_unit_disk").argtypes = [_u32p, _fp]; g("random_in_unit_disk").restype = None
    g("random_in_unit_sphere").argtypes = [_u32p, _fp]; g("random_in_unit_sphere").restype = None
    g("make_camera").argtypes = [_fp, _fp, _fp, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(rt.camera)]
    g("make_camera").restype = None
    g("get_ray").argtypes = [C.POINTER(rt.camera), C.c_float, C.c_float, _u32p, _fp, _fp]; g("get_ray").restype = None
    g("sphere_hit").argtypes = [C.POINTER(rt.sphere), _fp, _fp, C.c_float, C.c_float]; g("sphere_hit").restype = C.c_float
    g("triangle_hit").argtypes = [C.POINTER(rt.triangle), _fp, _fp, C.c_float, C.c_float, _fp, _fp]
    g("triangle_hit").restype = C.c_float
    g("hit_bbox").argtypes = [_fp, _fp, _fp, _fp, C.c_float]; g("hit_bbox").restype = C.c_int
    g("hit_bbox_dist").argtypes = [_fp, _fp, _fp, _fp, C.c_float]; g("hit_bbox_dist").restype = C.c_float
    g("plane_hit").argtypes = [C.POINTER(rt.plane), _fp, _fp, C.c_float, C.c_float]; g("plane_hit").restype = C.c_float
    g("schlick").argtypes = [C.c_float, C.c_float]; g("schlick").restype = C.c_float
    g("reflect").argtypes = [_fp, _fp, _fp]; g("reflect").restype = None
    g("refract").argtypes = [_fp, _fp, C.c_float, _fp]; g("refract").restype = None
    g("material_scatter").argtypes = [C.c_float, _fp, C.c_int, _fp, C.POINTER(rt.material), _fp, _u32p, C.POINTER(orc_scatter)]
    g("material_scatter").restype = None
    g("linear_to_srgb").argtypes = [C.c_float]; g("linear_to_srgb").restype = C.c_uint32
    if pre == "orc_":
        lib.orc_material_scatter_p.argtypes = [C.c_float, _fp, _fp, C.c_int, _fp, C.POINTER(rt.material), _fp, _u32p, C.POINTER(orc_scatter)]
        lib.orc_material_scatter_p.restype = None
    else:
        lib.ref_preset_scatter.argtypes = [C.c_int, C.c_float, _fp, _fp, C.c_int, _fp, _u32p, C.POINTER(orc_scatter)]
        lib.ref_preset_scatter.restype = None


_oracle = None
_ref = None


def _check_abi(lib, fn, path):
    """Refuses a library whose structs differ in size from the ctypes mirrors above (a rebuilt .so beside a stale mirror, or the reverse,
    would otherwise overrun the caller's buffer inside orc_render / ref_render_*)."""
    rt = _pkg()
    if not hasattr(lib, fn):
        raise ImportError(f"{path} predates the ABI handshake ({fn} missing): rebuild it (make -C oracle)")
    out = (C.c_int32 * 4)()
    getattr(lib, fn).argtypes = [C.POINTER(C.c_int32), C.c_int]
    getattr(lib, fn).restype = C.c_int
    n = getattr(lib, fn)(out, 4)
    mine = [C.sizeof(orc_scene), C.sizeof(orc_counters), C.sizeof(orc_scatter), C.sizeof(rt.render_options)]
    if n != 4 or list(out) != mine:
        raise ImportError(f"{path}: struct sizes {list(out)} (orc_scene, orc_counters, orc_scatter, rt_render_options) differ from the Python "
                          f"mirror's {mine}: library and oracle/oracle.py are of different generations - rebuild (make -C oracle)")


def load_oracle():
    global _oracle
    if _oracle is None:
        rt = _pkg()
        lib = C.CDLL(ORACLE_LIB)
        _check_abi(lib, "orc_abi_sizes", ORACLE_LIB)
        _bind_common(lib, "orc_")
        lib.orc_render.argtypes = [C.POINTER(orc_scene), C.POINTER(rt.camera), C.POINTER(rt.render_options),
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.POINTER(orc_counters)]
        lib.orc_render.restype = None
        lib.orc_hit_bvh.argtypes = [C.POINTER(orc_scene), _fp, _fp, C.c_float, C.c_float, C.c_int, _u32p, _fp, _fp,
                                    C.POINTER(orc_counters)]
        lib.orc_hit_bvh.restype = C.c_float
        lib.orc_generate_shadow_ray.argtypes = [C.POINTER(rt.render_options), _fp, _fp, _fp, _u32p, _fp]
        lib.orc_generate_shadow_ray.restype = C.c_int
        lib.orc_rmse.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.orc_rmse.restype = C.c_double
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(REF_LIB)


def load_ref():
    """The reference's own headers behind a C shim; only where oracle/_ref/libref.so was built."""
    global _ref
    if _ref is None:
        rt = _pkg()
        lib = C.CDLL(REF_LIB)
        _check_abi(lib, "ref_abi_sizes", REF_LIB)
        _bind_common(lib, "ref_")
        lib.ref_struct_sizes.argtypes = [C.POINTER(C.c_int), C.c_int]; lib.ref_struct_sizes.restype = C.c_int
        lib.ref_probe_light_dir.argtypes = [_fp, _fp, _fp, C.c_float, C.c_float, C.c_float, _fp]
        lib.ref_probe_light_dir.restype = None
        lib.ref_probe_light_contribution.argtypes = [_fp, _fp, C.c_float, C.c_float, _fp]
        lib.ref_probe_light_contribution.restype = None
        lib.ref_probe_phi.argtypes = [C.c_float]; lib.ref_probe_phi.restype = C.c_float
        lib.ref_render_spheres.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(rt.camera),
                                           C.c_int, C.c_int, C.c_float, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(orc_counters)]
        lib.ref_render_spheres.restype = None
        lib.ref_generate_shadow_ray.argtypes = [C.POINTER(rt.render_options), _fp, _fp, _fp, _u32p, _fp]
        lib.ref_generate_shadow_ray.restype = C.c_int
        lib.ref_render_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _fp, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(rt.camera), C.POINTER(rt.render_options),
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.POINTER(orc_counters)]
        lib.ref_render_mesh.restype = None
        # the reference's host I/O (staircase_scene.h: loadBVH, writePPM, setup_camera)
        lib.ref_load_bvh.argtypes = [C.c_char_p, C.POINTER(C.c_int)]; lib.ref_load_bvh.restype = C.c_void_p
        lib.ref_bvh_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, _fp]; lib.ref_bvh_copy.restype = None
        lib.ref_bvh_free.argtypes = [C.c_void_p]; lib.ref_bvh_free.restype = None
        lib.ref_write_ppm.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_long]; lib.ref_write_ppm.restype = C.c_long
        lib.ref_setup_camera.argtypes = [C.c_int, C.c_int, C.POINTER(rt.camera)]; lib.ref_setup_camera.restype = None
        _ref = lib
    return _ref


def ref_load_bvh(path):
    """The reference's loadBVH (staircase_scene.h:75-101) on `path`: (triangles, nodes, bounds6, nppl), or None when it refuses the file."""
    rt = _pkg()
    ref = load_ref()
    counts = (C.c_int * 3)()
    h = ref.ref_load_bvh(os.fsencode(path), counts)
    if not h:
        return None
    tris = np.zeros(counts[0], rt.triangle_dtype); bvh = np.zeros(counts[1], rt.bvh_node_dtype); bounds = (C.c_float * 6)()
    ref.ref_bvh_copy(h, tris.ctypes.data, bvh.ctypes.data, bounds)
    ref.ref_bvh_free(h)
    return tris, bvh, np.array(bounds[:], np.float32), counts[2]


def ref_write_ppm(fb):
    """The bytes the reference's writePPM (staircase_scene.h:32-43) sends to std::cout for the framebuffer fb[ny][nx][3]."""
    fb = np.ascontiguousarray(fb, dtype=np.float32)
    ref = load_ref()
    n = ref.ref_write_ppm(fb.shape[1], fb.shape[0], fb.ctypes.data, None, 0)
    buf = C.create_string_buffer(n)
    ref.ref_write_ppm(fb.shape[1], fb.shape[0], fb.ctypes.data, buf, n)
    return buf.raw[:n]


def ref_setup_camera(nx, ny):
    cam = _pkg().camera()
    load_ref().ref_setup_camera(nx, ny, C.byref(cam))
    return cam


_ref_main = None


def have_ref_main():
    return os.path.exists(REF_MAIN_LIB)


def load_ref_main():
    """The reference's main.cpp (its REF_00.01 reader / writer) behind a C shim; only where oracle/_ref/libref_main.so was built."""
    global _ref_main
    if _ref_main is None:
        lib = C.CDLL(REF_MAIN_LIB)
        lib.ref_load_reference.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]; lib.ref_load_reference.restype = C.c_int
        lib.ref_save_reference.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]; lib.ref_save_reference.restype = None
        _ref_main = lib
    return _ref_main


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def default_options(is_sphere_scene):
    """Same defaults as getDefaultRenderOptions of the renderer, without loading the HIP library."""
    rt = _pkg()
    o = rt.render_options()
    o.sky = rt.RT_SKY_GRADIENT if is_sphere_scene else rt.RT_SKY_CONST_GREY
    o.nee = 0 if is_sphere_scene else 1
    o.rr = 0 if is_sphere_scene else 1
    o.t_min = 0.001 if is_sphere_scene else 0.01
    o.rng = rt.RT_RNG_REFERENCE_STREAM
    o.fp = rt.RT_FP_PARITY
    o.light.center.e[0] = 52.514355
    o.light.center.e[1] = 715.686951
    o.light.center.e[2] = -272.620972
    o.light.radius = 50.0
    o.lightColor.e[0] = o.lightColor.e[1] = o.lightColor.e[2] = 20.0
    o.stripe_rows = 8
    o.num_devices = 0
    o.part_rank = 0
    o.part_world = 1
    return o


def sphere_scene(spheres, materials):
    rt = _pkg()
    spheres = np.ascontiguousarray(spheres, dtype=rt.sphere_dtype)
    materials = np.ascontiguousarray(materials, dtype=rt.material_dtype)
    sc = orc_scene()
    sc.spheres = spheres.ctypes.data
    sc.sphere_materials = materials.ctypes.data
    sc.num_spheres = len(spheres)
    sc._keep = (spheres, materials)
    return sc


def mesh_scene(host_mesh, materials, textures=(), floor=None):
    """floor = (norm xyz, point xyz) of kernel_scene.floor (only read with rt_render_options.floor = 1)."""
    rt = _pkg()
    materials = np.ascontiguousarray(materials, dtype=rt.material_dtype)
    sc = orc_scene()
    v = host_mesh.view
    sc.tris = C.cast(v.tris, C.c_void_p)
    sc.num_tris = v.numTris
    sc.bvh = C.cast(v.bvh, C.c_void_p)
    sc.num_bvh_nodes = v.numBvhNodes
    for a in range(3):
        sc.bounds[a] = v.bounds.min.e[a]
        sc.bounds[3 + a] = v.bounds.max.e[a]
    sc.nppl = host_mesh.nppl
    sc.materials = materials.ctypes.data
    sc.num_materials = len(materials)
    tex_arr = (rt.stexture * max(1, len(textures)))()
    keep = [materials, host_mesh, tex_arr]
    for k, t in enumerate(textures):
        t = np.ascontiguousarray(t, dtype=np.float32)
        keep.append(t)
        tex_arr[k].data = t.ctypes.data_as(C.POINTER(C.c_float))
        tex_arr[k].height, tex_arr[k].width = t.shape[0], t.shape[1]
    sc.textures = C.cast(tex_arr, C.c_void_p) if textures else None
    sc.num_textures = len(textures)
    if floor is not None:
        for a in range(6):
            sc.floor[a] = float(floor[a])
    sc._keep = keep
    return sc


def render(scene, cam, opt, nx, ny, ns, max_depth, region=None, counters=False, fb=None):
    """orc_render over the pixel rectangle region=(x0,y0,x1,y1) (default: whole image).
    Returns (fb[ny,nx,3] float32, counters or None)."""
    lib = load_oracle()
    if fb is None:
        fb = np.zeros((ny, nx, 3), np.float32)
    x0, y0, x1, y1 = region if region else (0, 0, nx, ny)
    cnt = orc_counters() if counters else None
    lib.orc_render(C.byref(scene), C.byref(cam), C.byref(opt), nx, ny, ns, max_depth, x0, y0, x1, y1,
                   fb.ctypes.data, C.byref(cnt) if counters else None)
    return fb, cnt


def ref_render_spheres(spheres, materials, cam, opt, nx, ny, ns, max_depth, region=None, counters=False, fb=None):
    """The reference-header host loop (oracle/ref_driver.cpp). Same return as render()."""
    rt = _pkg()
    lib = load_ref()
    spheres = np.ascontiguousarray(spheres, dtype=rt.sphere_dtype)
    materials = np.ascontiguousarray(materials, dtype=rt.material_dtype)
    if fb is None:
        fb = np.zeros((ny, nx, 3), np.float32)
    x0, y0, x1, y1 = region if region else (0, 0, nx, ny)
    cnt = orc_counters() if counters else None
    lib.ref_render_spheres(spheres.ctypes.data, materials.ctypes.data, len(spheres), C.byref(cam),
                           1 if opt.sky == rt.RT_SKY_GRADIENT else 0, opt.rr, opt.t_min,
                           1 if opt.rng == rt.RT_RNG_COUNTER else 0,
                           nx, ny, ns, max_depth, x0, y0, x1, y1, fb.ctypes.data, C.byref(cnt) if counters else None)
    return fb, cnt


def ref_render_mesh(scene, cam, opt, nx, ny, ns, max_depth, region=None, counters=False, fb=None):
    """The reference-arithmetic twin of the MESH path (oracle/ref_driver.cpp ref_render_mesh) on an orc_scene from mesh_scene().
    Same return as render()."""
    lib = load_ref()
    if fb is None:
        fb = np.zeros((ny, nx, 3), np.float32)
    x0, y0, x1, y1 = region if region else (0, 0, nx, ny)
    cnt = orc_counters() if counters else None
    lib.ref_render_mesh(scene.tris, scene.bvh, scene.num_bvh_nodes, scene.bounds, scene.nppl,
                        C.addressof(scene) + orc_scene.floor.offset, scene.materials, scene.textures,
                        C.byref(cam), C.byref(opt), nx, ny, ns, max_depth, x0, y0, x1, y1,
                        fb.ctypes.data, C.byref(cnt) if counters else None)
    return fb, cnt


def generate_shadow_ray(opt, origin, attenuation, normal, rng, which="orc"):
    """generateShadowRay (kernels.cu:363-393) on one input by the oracle ("orc") or the reference-header twin ("ref").
    Returns (generated, shadowDir[3], lightContribution[3], lightDist, cosAMax, draws, rng_after)."""
    lib = load_oracle() if which == "orc" else load_ref()
    fn = lib.orc_generate_shadow_ray if which == "orc" else lib.ref_generate_shadow_ray
    st = C.c_uint32(int(rng))
    out = (C.c_float * 9)()
    ok = fn(C.byref(opt), f3(origin), f3(attenuation), f3(normal), C.byref(st), out)
    o = np.array(out[:], np.float32)
    return ok, o[0:3], o[3:6], o[6], o[7], int(o[8]), st.value
