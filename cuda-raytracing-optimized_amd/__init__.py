"""Host-side mirror of the reference's renderer API, bound to the MI355X-native shared libraries.

The reference's host interface for the render hot path is three ``extern "C"`` functions
(/root/reference/kernels.h:6-8) called from ``main()`` (/root/reference/main.cpp:94-101,138):

    initRenderer(ksc, cam, &fb, nx, ny, maxDepth); runRenderer(ns, tx, ty); cleanupRenderer();

This module binds exactly those symbols (plus the additive ones of include/rt_api.h) from
``librt_mi355x.so`` with ctypes, and the host-side scene / BVH / PPM / .ref helpers of
include/rt_host.h from ``librt_host.so``.  Same names, same argument meaning, same error
behaviour (a HIP failure prints and ``exit(99)``s the process, /root/reference/kernels.cu:27-38).

There is NO CPU fallback: if ``librt_mi355x.so`` is missing the import of the renderer fails
loudly (``load_renderer``), and without a GPU ``initRenderer*`` terminates the process the way
the reference does.  The CPU oracle lives in ``oracle/`` and is never imported from here.

Directory name contains '-' (it mirrors the reference repo's name), so import it through the
``cuda_raytracing_optimized_amd`` alias module at the repo root.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
RENDERER_LIB = os.path.join(_HERE, "librt_mi355x.so")
HOST_LIB = os.path.join(_HERE, "librt_host.so")

RT_DIFFUSE, RT_METAL, RT_GLASS = 0, 1, 2
# additive: the reference's dormant look presets (scene_materials.h:22-93), see include/rt_types.h
(RT_FLOOR_COAT, RT_FLOOR_DIFFUSE, RT_FLOOR_CHECKER, RT_MODEL_COAT, RT_MODEL_DIFFUSE, RT_MODEL_GLOSSY, RT_MODEL_GLASS,
 RT_MODEL_TINTEDGLASS, RT_MODEL_SSS) = range(3, 12)
RT_SKY_CONST_GREY, RT_SKY_GRADIENT = 0, 1
RT_RNG_REFERENCE_STREAM, RT_RNG_COUNTER = 0, 1
RT_FP_PARITY, RT_FP_FAST = 0, 1
RT_MAX_DEVICES = 8

# ---------------------------------------------------------------------------------------------
# ctypes mirrors of include/rt_types.h (layouts identical to /root/reference/helper_structs.h)
# ---------------------------------------------------------------------------------------------


class vec3(C.Structure):
    _fields_ = [("e", C.c_float * 3)]


class camera(C.Structure):
    _fields_ = [("origin", vec3), ("lower_left_corner", vec3), ("horizontal", vec3), ("vertical", vec3),
                ("u", vec3), ("v", vec3), ("w", vec3), ("lens_radius", C.c_float)]


class sphere(C.Structure):
    _fields_ = [("center", vec3), ("radius", C.c_float)]


class plane(C.Structure):
    _fields_ = [("norm", vec3), ("point", vec3)]


class bbox(C.Structure):
    _fields_ = [("min", vec3), ("max", vec3)]


class triangle(C.Structure):
    _fields_ = [("v", vec3 * 3), ("texCoords", C.c_float * 6), ("meshID", C.c_ubyte), ("_pad", C.c_ubyte * 3)]


class bvh_node(C.Structure):
    _fields_ = [("a", vec3), ("b", vec3)]


class material(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", vec3), ("param", C.c_float), ("texId", C.c_int32)]


class stexture(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_float)), ("width", C.c_int32), ("height", C.c_int32)]


class mesh(C.Structure):
    _fields_ = [("tris", C.POINTER(triangle)), ("numTris", C.c_uint32), ("bvh", C.POINTER(bvh_node)),
                ("numBvhNodes", C.c_int32), ("bounds", bbox)]


class kernel_scene(C.Structure):
    _fields_ = [("m", C.POINTER(mesh)), ("floor", plane), ("materials", C.POINTER(material)),
                ("numMaterials", C.c_int32), ("textures", C.POINTER(stexture)), ("numTextures", C.c_int32),
                ("numPrimitivesPerLeaf", C.c_int32)]


class render_options(C.Structure):
    _fields_ = [("sky", C.c_int32), ("nee", C.c_int32), ("rr", C.c_int32), ("t_min", C.c_float),
                ("rng", C.c_int32), ("fp", C.c_int32), ("light", sphere), ("lightColor", vec3),
                ("stripe_rows", C.c_int32), ("num_devices", C.c_int32), ("devices", C.c_int32 * RT_MAX_DEVICES),
                ("part_rank", C.c_int32), ("part_world", C.c_int32), ("variant", C.c_int32), ("counters", C.c_int32),
                ("samples_per_item", C.c_int32), ("floor", C.c_int32)]


class render_stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("total_ms", C.c_double), ("samples", C.c_int64),
                ("num_launches", C.c_int32), ("vgprs", C.c_int32), ("rays", C.c_uint64),
                ("prim_tests", C.c_uint64), ("node_visits", C.c_uint64), ("exec_tests", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("box_tests", C.c_uint64), ("ref_stats", C.c_uint64 * 18)]


# indices into render_stats.ref_stats: the reference's STATS counters, /root/reference/kernels.cu:47-67
(RT_STAT_PRIMARY, RT_STAT_PRIMARY_HIT_MESH, RT_STAT_PRIMARY_NOHITS, RT_STAT_PRIMARY_BBOX_NOHITS, RT_STAT_SECONDARY,
 RT_STAT_SECONDARY_MESH, RT_STAT_SECONDARY_NOHIT, RT_STAT_SECONDARY_MESH_NOHIT, RT_STAT_SECONDARY_BBOX_NOHIT, RT_STAT_SHADOWS,
 RT_STAT_SHADOWS_BBOX_NOHITS, RT_STAT_SHADOWS_NOHITS, RT_STAT_LOW_POWER, RT_STAT_EXCEED_MAX_BOUNCE, RT_STAT_RUSSIAN_KILL,
 RT_STAT_NAN, RT_STAT_NODES_BOTH, RT_STAT_NODES_SINGLE) = range(18)
RT_STAT_NAMES = ["primary", "primary hit mesh", "primary nohit", "primary bb nohit", "secondary", "secondary mesh", "secondary no hit",
                 "secondary mesh nohit", "secondary bb nohit", "shadows", "shadows bb nohit", "shadows nohit", "power < 0.01",
                 "exceeded max bounce", "russian roulette", "NaNs", "both nodes hit", "single node hit"]


_SIZES = {vec3: 12, camera: 88, sphere: 16, plane: 24, bbox: 24, triangle: 64, bvh_node: 24, material: 24,
          stexture: 16, mesh: 56, kernel_scene: 64}
for _t, _s in _SIZES.items():
    assert C.sizeof(_t) == _s, (_t, C.sizeof(_t), _s)

# numpy views of the array-of-struct types
sphere_dtype = np.dtype([("center", np.float32, 3), ("radius", np.float32)])
material_dtype = np.dtype([("type", np.int32), ("color", np.float32, 3), ("param", np.float32), ("texId", np.int32)])
triangle_dtype = np.dtype([("v", np.float32, (3, 3)), ("texCoords", np.float32, 6), ("meshID", np.uint8), ("_pad", np.uint8, 3)])
bvh_node_dtype = np.dtype([("a", np.float32, 3), ("b", np.float32, 3)])
assert sphere_dtype.itemsize == 16 and material_dtype.itemsize == 24
assert triangle_dtype.itemsize == 64 and bvh_node_dtype.itemsize == 24

# symbols every library must export (tests check the .so against the headers with these)
RENDERER_SYMBOLS = ["initRenderer", "runRenderer", "cleanupRenderer", "initRendererSpheres",
                    "getDefaultRenderOptions", "setRenderOptions", "setExternalFramebuffer", "getRenderStats",
                    "rtDeviceCount", "rtApiVersion", "rtStructSizes"]
RT_API_VERSION = 1002       # include/rt_api.h: the version this mirror was written against
# the structs that cross the C-ABI, in the order of the RT_SIZEOF_* indices of include/rt_api.h
ABI_STRUCTS = [render_options, render_stats, camera, sphere, material, triangle, bvh_node, mesh, kernel_scene, stexture, plane, bbox, vec3]
HOST_SYMBOLS = ["rtMakeCamera", "rtRandomFloat", "rtSceneThreeSpheres", "rtSceneRandomSpheres", "rtStaircaseCamera",
                "rtBuildBvh", "rtBuildBvhLevels", "rtLoadBvhFile", "rtSaveBvhFile", "rtFreeMesh", "rtMeshView",
                "rtSceneStaircaseProcedural", "rtLinearToSRGB", "rtWritePPM", "rtSaveReference", "rtLoadReference", "rtRmse"]

_renderer = None
_host = None


def load_host():
    """librt_host.so (plain C++; works without a GPU)."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise ImportError(f"{HOST_LIB} is not built: run `make` (or __graft_entry__.build())")
        h = C.CDLL(HOST_LIB)
        fp = C.POINTER(C.c_float)
        h.rtMakeCamera.argtypes = [fp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(camera)]
        h.rtMakeCamera.restype = None
        h.rtRandomFloat.argtypes = [C.POINTER(C.c_uint32)]
        h.rtRandomFloat.restype = C.c_float
        h.rtSceneThreeSpheres.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(camera)]
        h.rtSceneThreeSpheres.restype = C.c_int
        h.rtSceneRandomSpheres.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(camera)]
        h.rtSceneRandomSpheres.restype = C.c_int
        h.rtStaircaseCamera.argtypes = [C.c_int, C.c_int, C.POINTER(camera)]
        h.rtStaircaseCamera.restype = None
        h.rtBuildBvh.argtypes = [C.c_void_p, C.c_int, C.c_int]
        h.rtBuildBvh.restype = C.c_void_p
        h.rtBuildBvhLevels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        h.rtBuildBvhLevels.restype = C.c_void_p
        h.rtLoadBvhFile.argtypes = [C.c_char_p]
        h.rtLoadBvhFile.restype = C.c_void_p
        h.rtSaveBvhFile.argtypes = [C.c_void_p, C.c_char_p]
        h.rtSaveBvhFile.restype = C.c_int
        h.rtFreeMesh.argtypes = [C.c_void_p]
        h.rtFreeMesh.restype = None
        h.rtMeshView.argtypes = [C.c_void_p, C.POINTER(mesh)]
        h.rtMeshView.restype = C.c_int
        h.rtSceneStaircaseProcedural.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        h.rtSceneStaircaseProcedural.restype = C.c_int
        h.rtLinearToSRGB.argtypes = [C.c_float]
        h.rtLinearToSRGB.restype = C.c_uint32
        h.rtWritePPM.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        h.rtWritePPM.restype = C.c_int
        h.rtSaveReference.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        h.rtSaveReference.restype = C.c_int
        h.rtLoadReference.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        h.rtLoadReference.restype = C.c_int
        h.rtRmse.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        h.rtRmse.restype = C.c_double
        _host = h
    return _host


def check_abi(lib, path=RENDERER_LIB, structs=None):
    """Version + struct-size handshake with a loaded librt_mi355x.so.  getDefaultRenderOptions / getRenderStats write sizeof(struct) bytes
    through the caller's pointer: a mirror of another generation than the library must be an ImportError here, not a heap overrun there
    (what the round-2 crash was: a rebuilt library beside a stale mirror)."""
    structs = ABI_STRUCTS if structs is None else structs
    if not hasattr(lib, "rtStructSizes"):
        raise ImportError(f"{path} predates the ABI handshake (no rtStructSizes): rebuild it with `make`")
    lib.rtApiVersion.argtypes = []
    lib.rtApiVersion.restype = C.c_int
    lib.rtStructSizes.argtypes = [C.POINTER(C.c_int32), C.c_int]
    lib.rtStructSizes.restype = C.c_int
    ver = lib.rtApiVersion()
    if ver != RT_API_VERSION:
        raise ImportError(f"{path} has API version {ver}, this binding was written against {RT_API_VERSION}: rebuild the library or update the mirror")
    out = (C.c_int32 * len(structs))()
    n = lib.rtStructSizes(out, len(structs))
    if n != len(structs):
        raise ImportError(f"{path} reports {n} ABI structs, the mirror has {len(structs)}")
    for t, sz in zip(structs, out):
        if C.sizeof(t) != sz:
            raise ImportError(f"{path}: sizeof({t.__name__}) is {sz} in the library and {C.sizeof(t)} in the Python mirror - "
                              "library and mirror are of different generations; refusing to call into it")


def load_renderer():
    """librt_mi355x.so — the HIP renderer.  Raises ImportError if it is not built: there is no fallback."""
    global _renderer
    if _renderer is None:
        if not os.path.exists(RENDERER_LIB):
            raise ImportError(f"{RENDERER_LIB} is not built: run `make` (or __graft_entry__.build()); "
                              "there is no CPU fallback for the render path")
        r = C.CDLL(RENDERER_LIB)
        check_abi(r)
        r.initRenderer.argtypes = [kernel_scene, camera, C.POINTER(C.POINTER(vec3)), C.c_int, C.c_int, C.c_int]
        r.initRenderer.restype = None
        r.runRenderer.argtypes = [C.c_int, C.c_int, C.c_int]
        r.runRenderer.restype = None
        r.cleanupRenderer.argtypes = []
        r.cleanupRenderer.restype = None
        r.initRendererSpheres.argtypes = [C.c_void_p, C.c_void_p, C.c_int, camera, C.POINTER(C.POINTER(vec3)),
                                          C.c_int, C.c_int, C.c_int]
        r.initRendererSpheres.restype = None
        r.getDefaultRenderOptions.argtypes = [C.POINTER(render_options), C.c_int]
        r.getDefaultRenderOptions.restype = None
        r.setRenderOptions.argtypes = [C.POINTER(render_options)]
        r.setRenderOptions.restype = None
        r.setExternalFramebuffer.argtypes = [C.c_void_p]
        r.setExternalFramebuffer.restype = None
        r.getRenderStats.argtypes = [C.POINTER(render_stats)]
        r.getRenderStats.restype = None
        r.rtDeviceCount.argtypes = []
        r.rtDeviceCount.restype = C.c_int
        r.rtApiVersion.argtypes = []
        r.rtApiVersion.restype = C.c_int
        _renderer = r
    return _renderer


# ---------------------------------------------------------------------------------------------
# host-side helpers (scene set-up; what main.cpp / staircase_scene.h do before initRenderer)
# ---------------------------------------------------------------------------------------------

def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def make_camera(lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist):
    """camera::camera, /root/reference/helper_structs.h:194-207."""
    cam = camera()
    load_host().rtMakeCamera(_f3(lookfrom), _f3(lookat), _f3(vup), vfov, aspect, aperture, focus_dist, C.byref(cam))
    return cam


def scene_three_spheres(nx, ny):
    """C1 of SURVEY.md §8d. Returns (spheres, materials, camera)."""
    sp = np.zeros(3, sphere_dtype)
    mt = np.zeros(3, material_dtype)
    cam = camera()
    n = load_host().rtSceneThreeSpheres(sp.ctypes.data, mt.ctypes.data, 3, nx, ny, C.byref(cam))
    assert n == 3
    return sp, mt, cam


def scene_random_spheres(nx, ny, seed=0):
    """C2/C3/C5 of SURVEY.md §8d: 488 spheres. Returns (spheres, materials, camera)."""
    cap = 488
    sp = np.zeros(cap, sphere_dtype)
    mt = np.zeros(cap, material_dtype)
    cam = camera()
    n = load_host().rtSceneRandomSpheres(seed, sp.ctypes.data, mt.ctypes.data, cap, nx, ny, C.byref(cam))
    assert n == cap, n
    return sp, mt, cam


def staircase_camera(nx, ny):
    cam = camera()
    load_host().rtStaircaseCamera(nx, ny, C.byref(cam))
    return cam


class HostMesh:
    """Owns a BVH'd mesh built/loaded by librt_host.so; `.view` is an rt_mesh for kernel_scene.m."""

    def __init__(self, handle):
        if not handle:
            raise ValueError("mesh build/load failed")
        self._h = C.c_void_p(handle)
        self.view = mesh()
        self.nppl = load_host().rtMeshView(self._h, C.byref(self.view))

    @classmethod
    def build(cls, tris, nppl=5, extra_levels=None):
        """rtBuildBvh; extra_levels: rtBuildBvhLevels (None = the builder's default)."""
        tris = np.ascontiguousarray(tris, dtype=triangle_dtype)
        if extra_levels is None:
            return cls(load_host().rtBuildBvh(tris.ctypes.data, len(tris), nppl))
        return cls(load_host().rtBuildBvhLevels(tris.ctypes.data, len(tris), nppl, extra_levels))

    @classmethod
    def load(cls, path):
        return cls(load_host().rtLoadBvhFile(os.fsencode(path)))

    def save(self, path):
        return load_host().rtSaveBvhFile(self._h, os.fsencode(path))

    @property
    def tris(self):
        return np.ctypeslib.as_array(C.cast(self.view.tris, C.POINTER(C.c_ubyte)), (self.view.numTris * 64,)).view(triangle_dtype)

    @property
    def bvh(self):
        return np.ctypeslib.as_array(C.cast(self.view.bvh, C.POINTER(C.c_ubyte)), (self.view.numBvhNodes * 24,)).view(bvh_node_dtype)

    def close(self):
        if self._h:
            load_host().rtFreeMesh(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def scene_staircase_procedural(detail=1):
    """Procedural stand-in for the absent staircase asset. Returns (triangles, materials[20])."""
    h = load_host()
    mats = np.zeros(20, material_dtype)
    need = h.rtSceneStaircaseProcedural(detail, None, 0, mats.ctypes.data)
    n = -need if need < 0 else need
    tris = np.zeros(n, triangle_dtype)
    got = h.rtSceneStaircaseProcedural(detail, tris.ctypes.data, n, mats.ctypes.data)
    assert got == n, (got, n)
    return tris, mats


def make_kernel_scene(host_mesh, materials, textures=(), floor=None):
    """setup_kernel_scene, /root/reference/staircase_scene.h:166-184. Returns (kernel_scene, keepalive).
    floor = (norm xyz, point xyz) of kernel_scene.floor (read by the renderer only with rt_render_options.floor = 1)."""
    materials = np.ascontiguousarray(materials, dtype=material_dtype)
    ks = kernel_scene()
    ks.m = C.pointer(host_mesh.view)
    ks.materials = C.cast(materials.ctypes.data, C.POINTER(material))
    ks.numMaterials = len(materials)
    tex_arr = (stexture * max(1, len(textures)))()
    keep = [materials, tex_arr, host_mesh]
    for k, t in enumerate(textures):
        t = np.ascontiguousarray(t, dtype=np.float32)       # (height, width, 3)
        keep.append(t)
        tex_arr[k].data = t.ctypes.data_as(C.POINTER(C.c_float))
        tex_arr[k].height, tex_arr[k].width = t.shape[0], t.shape[1]
    ks.textures = tex_arr if textures else None
    ks.numTextures = len(textures)
    ks.numPrimitivesPerLeaf = host_mesh.nppl
    if floor is not None:
        for a in range(3):
            ks.floor.norm.e[a] = float(floor[a])
            ks.floor.point.e[a] = float(floor[3 + a])
    return ks, keep


# ---------------------------------------------------------------------------------------------
# renderer API (same names as the C symbols)
# ---------------------------------------------------------------------------------------------

_state = {"fb": None, "nx": 0, "ny": 0, "keep": None}


def _fb_view(fbp, nx, ny):
    arr = np.ctypeslib.as_array(C.cast(fbp, C.POINTER(C.c_float)), (ny, nx, 3))
    return arr


def initRenderer(ksc, cam, nx, ny, maxDepth, keepalive=None):
    """extern "C" initRenderer (/root/reference/kernels.h:6). Returns the framebuffer as a (ny, nx, 3) float32 view."""
    r = load_renderer()
    fbp = C.POINTER(vec3)()
    r.initRenderer(ksc, cam, C.byref(fbp), nx, ny, maxDepth)
    _state.update(fb=_fb_view(fbp, nx, ny), nx=nx, ny=ny, keep=keepalive)
    return _state["fb"]


def initRendererSpheres(spheres, materials, cam, nx, ny, maxDepth):
    """Additive: sphere-scene initialiser (include/rt_api.h). Returns the framebuffer view."""
    r = load_renderer()
    spheres = np.ascontiguousarray(spheres, dtype=sphere_dtype)
    materials = np.ascontiguousarray(materials, dtype=material_dtype)
    if len(spheres) != len(materials):
        raise ValueError("one material per sphere")
    fbp = C.POINTER(vec3)()
    r.initRendererSpheres(spheres.ctypes.data, materials.ctypes.data, len(spheres), cam, C.byref(fbp), nx, ny, maxDepth)
    _state.update(fb=_fb_view(fbp, nx, ny), nx=nx, ny=ny, keep=None)
    return _state["fb"]


def runRenderer(ns, tx=8, ty=8):
    """extern "C" runRenderer (/root/reference/kernels.h:7); blocking."""
    load_renderer().runRenderer(ns, tx, ty)


def cleanupRenderer():
    """extern "C" cleanupRenderer (/root/reference/kernels.h:8). The framebuffer view is dead afterwards."""
    load_renderer().cleanupRenderer()
    _state.update(fb=None, keep=None)


def getDefaultRenderOptions(is_sphere_scene):
    o = render_options()
    load_renderer().getDefaultRenderOptions(C.byref(o), 1 if is_sphere_scene else 0)
    return o


def setRenderOptions(opt=None, **kw):
    """Set options for the following runRenderer calls; keyword arguments override fields of `opt`."""
    if opt is None:
        raise ValueError("pass the options struct from getDefaultRenderOptions()")
    for k, v in kw.items():
        if k == "devices":
            opt.num_devices = len(v)
            for idx, d in enumerate(v):
                opt.devices[idx] = d
        else:
            setattr(opt, k, v)
    load_renderer().setRenderOptions(C.byref(opt))
    return opt


def setExternalFramebuffer(array):
    """Deliver the following renders into `array` ((ny, nx, 3) float32, C-contiguous, e.g. a shared memmap); None reverts."""
    if array is None:
        load_renderer().setExternalFramebuffer(None)
        _state["ext"] = None
        return
    assert array.dtype == np.float32 and array.flags["C_CONTIGUOUS"] and array.shape == (_state["ny"], _state["nx"], 3)
    load_renderer().setExternalFramebuffer(array.ctypes.data)
    _state["ext"] = array


def getRenderStats():
    s = render_stats()
    load_renderer().getRenderStats(C.byref(s))
    return s


def device_count():
    return load_renderer().rtDeviceCount()


# ---------------------------------------------------------------------------------------------
# output / verification harness (main.cpp:25-60,105-128; staircase_scene.h:22-43)
# ---------------------------------------------------------------------------------------------

def write_ppm(path, fb):
    fb = np.ascontiguousarray(fb, dtype=np.float32)
    return load_host().rtWritePPM(os.fsencode(path), fb.shape[1], fb.shape[0], fb.ctypes.data)


def save_reference(path, fb):
    fb = np.ascontiguousarray(fb, dtype=np.float32)
    return load_host().rtSaveReference(os.fsencode(path), fb.shape[1], fb.shape[0], fb.ctypes.data)


def load_reference(path, nx, ny):
    out = np.zeros((ny, nx, 3), np.float32)
    rc = load_host().rtLoadReference(os.fsencode(path), out.ctypes.data, nx, ny)
    return rc, out


def rmse(f, g):
    f = np.ascontiguousarray(f, dtype=np.float32)
    g = np.ascontiguousarray(g, dtype=np.float32)
    assert f.shape == g.shape
    return load_host().rtRmse(f.ctypes.data, g.ctypes.data, f.shape[1], f.shape[0])


# ---------------------------------------------------------------------------------------------
# per-function device probes (include/rt_probe.h)
# ---------------------------------------------------------------------------------------------

PROBE_SYMBOLS = [f"rtProbe{n}_{m}" for n in ("Rng", "DiskSphere", "GetRay", "SphereHit", "TriangleHit", "Bbox", "Scatter", "Math", "ShadowRay", "PlaneHit", "SinCos", "Schlick")
                 for m in ("parity", "fast")]


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Probe:
    """probe = Probe('parity'); probe.sphere_hit(...)  — every method returns numpy arrays."""

    def __init__(self, mode="parity"):
        assert mode in ("parity", "fast")
        self.lib = load_renderer()
        self.sfx = "_" + mode

    def _fn(self, name):
        f = getattr(self.lib, "rtProbe" + name + self.sfx)
        f.restype = None
        return f

    def rng(self, pixel_ids):
        ids = _u32(pixel_ids); n = len(ids)
        seed = np.zeros(n, np.uint32); draws = np.zeros((n, 4), np.float32); state = np.zeros(n, np.uint32)
        self._fn("Rng")(_p(ids), C.c_int(n), _p(seed), _p(draws), _p(state))
        return seed, draws, state

    def disk_sphere(self, states):
        st = _u32(states); n = len(st)
        disk = np.zeros((n, 3), np.float32); sd = np.zeros(n, np.uint32)
        sph = np.zeros((n, 3), np.float32); ss = np.zeros(n, np.uint32)
        self._fn("DiskSphere")(_p(st), C.c_int(n), _p(disk), _p(sd), _p(sph), _p(ss))
        return disk, sd, sph, ss

    def get_ray(self, cam, s, t, states):
        s = _f32(s); t = _f32(t); st = _u32(states); n = len(s)
        org = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32); sa = np.zeros(n, np.uint32)
        self._fn("GetRay")(C.byref(cam), _p(s), _p(t), _p(st), C.c_int(n), _p(org), _p(d), _p(sa))
        return org, d, sa

    def sphere_hit(self, spheres, org, dirs, tmin, tmax):
        sp = np.ascontiguousarray(spheres, dtype=sphere_dtype); n = len(sp)
        org = _f32(org); dirs = _f32(dirs); tmin = _f32(tmin); tmax = _f32(tmax)
        out = np.zeros(n, np.float32)
        self._fn("SphereHit")(_p(sp), _p(org), _p(dirs), _p(tmin), _p(tmax), C.c_int(n), _p(out))
        return out

    def triangle_hit(self, tris, org, dirs, tmin, tmax):
        tr = np.ascontiguousarray(tris, dtype=triangle_dtype); n = len(tr)
        org = _f32(org); dirs = _f32(dirs); tmin = _f32(tmin); tmax = _f32(tmax)
        t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        self._fn("TriangleHit")(_p(tr), _p(org), _p(dirs), _p(tmin), _p(tmax), C.c_int(n), _p(t), _p(u), _p(v))
        return t, u, v

    def bbox(self, bmin, bmax, org, dirs, tmax):
        bmin = _f32(bmin); bmax = _f32(bmax); org = _f32(org); dirs = _f32(dirs); tmax = _f32(tmax); n = len(tmax)
        dist = np.zeros(n, np.float32); hit = np.zeros(n, np.int32)
        self._fn("Bbox")(_p(bmin), _p(bmax), _p(org), _p(dirs), _p(tmax), C.c_int(n), _p(dist), _p(hit))
        return dist, hit

    def scatter(self, t, normal, inside, wo, mats, color, states, hit_point=None):
        t = _f32(t); normal = _f32(normal); inside = np.ascontiguousarray(inside, dtype=np.int32); wo = _f32(wo)
        hp = _f32(hit_point if hit_point is not None else np.zeros((len(t), 3), np.float32))
        mats = np.ascontiguousarray(mats, dtype=material_dtype); color = _f32(color); st = _u32(states); n = len(t)
        wi = np.zeros((n, 3), np.float32); thr = np.zeros((n, 3), np.float32); flags = np.zeros(n, np.int32)
        tout = np.zeros(n, np.float32); sa = np.zeros(n, np.uint32)
        self._fn("Scatter")(_p(t), _p(hp), _p(normal), _p(inside), _p(wo), _p(mats), _p(color), _p(st), C.c_int(n),
                            _p(wi), _p(thr), _p(flags), _p(tout), _p(sa))
        return wi, thr, flags, tout, sa

    def shadow_ray(self, light, light_color, org, atten, normal, states):
        """generateShadowRay (kernels.cu:363-393). Returns (generated, shadowDir, lightContribution, lightDist, cosAMax, draws, state_after)."""
        org = _f32(org); atten = _f32(atten); normal = _f32(normal); st = _u32(states); n = len(st)
        out = np.zeros((n, 9), np.float32); ok = np.zeros(n, np.int32); sa = np.zeros(n, np.uint32)
        self._fn("ShadowRay")(C.byref(light), C.byref(light_color), _p(org), _p(atten), _p(normal), _p(st), C.c_int(n), _p(out), _p(ok), _p(sa))
        return ok, out[:, 0:3], out[:, 3:6], out[:, 6], out[:, 7], out[:, 8].astype(np.int32), sa

    def plane_hit(self, planes, org, dirs, tmin, tmax):
        pl = np.ascontiguousarray(planes, dtype=np.float32).reshape(-1, 6); n = len(pl)
        org = _f32(org); dirs = _f32(dirs); tmin = _f32(tmin); tmax = _f32(tmax)
        out = np.zeros(n, np.float32)
        self._fn("PlaneHit")(_p(pl), _p(org), _p(dirs), _p(tmin), _p(tmax), C.c_int(n), _p(out))
        return out

    def sincos(self, y):
        """sinf / cosf as generateShadowRay computes them on the device (csrc/rt_glibc_sincosf.h)."""
        y = _f32(y); n = len(y)
        s = np.zeros(n, np.float32); c = np.zeros(n, np.float32)
        self._fn("SinCos")(_p(y), C.c_int(n), _p(s), _p(c))
        return s, c

    def schlick(self, cosine, ref_idx, u):
        """schlick (material.h:9-13) with the device's powf(x, 5) (csrc/rt_glibc_powf.h), and `u < schlick` as the path decides it."""
        c = _f32(cosine); r = _f32(ref_idx); u = _f32(u); n = len(c)
        out = np.zeros(n, np.float32); above = np.zeros(n, np.int32)
        self._fn("Schlick")(_p(c), _p(r), _p(u), C.c_int(n), _p(out), _p(above))
        return out, above

    def math(self, a, b):
        a = _f32(a); b = _f32(b); n = len(a)
        q = np.zeros(n, np.float32); r = np.zeros(n, np.float32); p5 = np.zeros(n, np.float32); u3 = np.zeros((n, 3), np.float32)
        self._fn("Math")(_p(a), _p(b), C.c_int(n), _p(q), _p(r), _p(p5), _p(u3))
        return q, r, p5, u3
