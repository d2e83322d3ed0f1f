"""GPU parity per FUNCTION of the hot path (SURVEY.md §8a rows a-6 .. a-13): each device function is
run by itself through the probe C-ABI (include/rt_probe.h) on seeded random + edge-case inputs and
compared with the CPU oracle.  PARITY build: bit-exact (integer work and IEEE fp32 +,-,*,/,sqrt; the libm calls of the path - powf(x, 5), sinf,
cosf - are glibc's algorithms restated for the device: csrc/rt_glibc_powf.h, csrc/rt_glibc_sincosf.h)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 3000


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


@pytest.fixture(scope="module")
def probe(rt):
    return rt.Probe("parity")


@pytest.fixture(scope="module")
def lib(O):
    return O.load_oracle()


def test_basic_math_is_ieee(probe, lib):
    rng = np.random.default_rng(1)
    a = rng.uniform(-3, 3, N).astype(np.float32)
    b = rng.uniform(0.01, 5, N).astype(np.float32) * rng.choice([-1, 1], N).astype(np.float32)
    q, r, p5, u3 = probe.math(a, b)
    assert np.array_equal(_bits(q), _bits(a / b))
    assert np.array_equal(_bits(r), _bits(np.sqrt(np.abs(a))))
    v = np.stack([a, b, a - b], 1)
    ln = np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2])
    assert np.array_equal(_bits(u3), _bits(v / ln[:, None]))
    lp = np.zeros(N, np.float32)                                          # powf(a, 5.0f) of this machine's libm (material.h:12 calls it)
    lib.orc_libm_powf5.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.orc_libm_powf5(a.ctypes.data, N, lp.ctypes.data)
    assert np.array_equal(_bits(p5), _bits(lp))


def test_division_and_unit_vector_through_fp64_are_ieee(probe):
    """vec3 / float and unit_vector on the device go through an fp64 reciprocal / square-root iteration (csrc/rt_div64.h; the host twin of the same text is held to
    IEEE on 2^29 cases in test_div64_twin_is_ieee).  Here the DEVICE's bits - v_rcp_f32 / v_rsq_f32 seeds included - on 2^20 operand pairs: random ones, quotients beside
    rounding boundaries, unit vectors and nearly-unit vectors (the render path's case), tiny and huge lengths (the plain-operator fall-backs)."""
    rng = np.random.default_rng(11)
    n = 1 << 20
    b = (rng.uniform(0.001, 1000, n) * rng.choice([-1, 1], n)).astype(np.float32)
    m = ((rng.integers(1 << 23, 1 << 24, n) * 2 + 1).astype(np.float64) * 2.0 ** -25)
    a = np.where(np.arange(n) % 2 == 0, rng.uniform(-1000, 1000, n), b.astype(np.float64) * m * 2.0 ** rng.integers(-8, 8, n)).astype(np.float32)
    a[:8] = (0.0, -0.0, 1e-30, -1e-30, 1e30, 3.0, 1.0, 1e-38); b[:8] = (2.0, 3.0, 1e10, 1e12, 1e-20, 3.0, 1e-25, 7.0)
    q, r, _, u3 = probe.math(a, b)
    with np.errstate(all="ignore"):
        assert np.array_equal(_bits(q), _bits(a / b))
        assert np.array_equal(_bits(r), _bits(np.sqrt(np.abs(a))))
        v = np.stack([a, b, a - b], 1)
        ln = np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2])
        assert np.array_equal(_bits(u3), _bits(v / ln[:, None]))
    # nearly-unit vectors (what hit() renormalises) and tiny / huge ones
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    scale = np.where(np.arange(n) % 16 == 0, 10.0 ** rng.uniform(-30, 30, n), 1.0)
    d = (d * scale[:, None]).astype(np.float32)
    _, _, _, u3 = probe.math(d[:, 0], d[:, 1])               # unit(F3(a, b, a - b)): two free components
    with np.errstate(all="ignore"):
        v = np.stack([d[:, 0], d[:, 1], d[:, 0] - d[:, 1]], 1)
        ln = np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2])
        want = v / ln[:, None]
    ok = np.isfinite(want).all(axis=1)
    assert ok.mean() > 0.9 and np.array_equal(_bits(u3[ok]), _bits(want[ok]))


def test_powf5_and_schlick_are_the_cpu_libm(probe, lib):
    """material.h:9-13: schlick calls pow(1 - cosine, 5.0f) - powf, glibc's on the CPU side.  The device runs that algorithm restated
    (csrc/rt_glibc_powf.h; on the host the same text equals libm on every float: test_glibc_powf5_twin_is_libm).  Here the DEVICE's bits: powf(x, 5) on
    2^20 arguments of [0, 2] (every 1 - min(cos, 1)), on 2^18 spread over all floats (negative, subnormal, huge, inf, NaN) and on the boundary cases;
    schlick on 2^20 (cosine, ref_idx) pairs - glass and its reciprocal, the presets' indices, random ones - against orc_schlick.  Every bit equal."""
    rng = np.random.default_rng(5)
    xb = np.concatenate([rng.integers(0, 0x40000001, 1 << 20, dtype=np.uint64), rng.integers(0, 1 << 32, 1 << 18, dtype=np.uint64),
                         np.array([0, 0x80000000, 1, 0x007FFFFF, 0x00800000, 0x3F800000, 0x3F7FFFFF, 0x3F800001, 0x40000000, 0x7F800000, 0xFF800000,
                                   0x7FC00000, 0x4C000000, 0x4B800000, 0x20000000, 0x21000000, 0x21800000, 0xBF800000, 0xC0000000], np.uint64)]).astype(np.uint32)
    x = xb.view(np.float32)
    pad = np.ones(len(x), np.float32)
    _, _, p5, _ = probe.math(x, pad)
    lp = np.zeros(len(x), np.float32)
    lib.orc_libm_powf5.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.orc_libm_powf5(x.ctypes.data, len(x), lp.ctypes.data)
    nan = np.isnan(lp)
    assert np.array_equal(np.isnan(p5), nan)
    assert np.array_equal(_bits(p5)[~nan], _bits(lp)[~nan])
    n = 1 << 20
    cosine = np.concatenate([rng.uniform(0.0, 1.0, n - 4096), 1.0 - rng.uniform(0, 1, 2048) ** 6, rng.uniform(-1e-3, 0.0, 2040),
                             np.array([0.0, 1.0, 0.5, 1e-8, 1.0 - 2.0 ** -24, 2.0 ** -24, -0.0, 0.99999994], np.float64)]).astype(np.float32)
    idx = rng.choice(np.array([1.5, 1.0 / 1.5, 1.1, 1.0 / 1.1, 1.333, 1.0 / 1.333], np.float32), n)
    idx[::7] = rng.uniform(0.3, 3.0, len(idx[::7])).astype(np.float32)
    want = np.zeros(n, np.float32)
    lib.orc_schlick_array.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.orc_schlick_array(cosine.ctypes.data, idx.ctypes.data, n, want.ctypes.data)
    # the path only ever asks `rnd < schlick` (material.h:58), and decides it from a bracket around a cheap x^5 unless the draw is within a few ulps of
    # the threshold (rt_device.h schlick_above): draws AT the threshold, one and two steps of the 2^-24 grid and of the float grid beside it, and random ones
    u = rng.integers(0, 1 << 24, n).astype(np.float32) / np.float32(16777216.0)
    near = np.floor(want.astype(np.float64) * 16777216.0)
    sel = np.arange(n) % 4
    u = np.where(sel == 1, ((near + rng.integers(-2, 3, n)) / 16777216.0).astype(np.float32), u)
    u = np.where(sel == 2, (_bits(want).astype(np.int64) + rng.integers(-3, 4, n)).clip(0, 0x7F7FFFFF).astype(np.uint32).view(np.float32), u).astype(np.float32)
    got, above = probe.schlick(cosine, idx, u)
    assert np.array_equal(_bits(got), _bits(want))
    assert np.array_equal(above != 0, u < want)
    assert (u == want).sum() > 1000 and (above != 0).sum() > n // 8 and (above == 0).sum() > n // 8


def test_rng_known_answers(probe, lib):
    ids = np.array([0, 12345, 1, 959999, 2 ** 31 + 7] + list(range(100, 160)), np.uint32)
    seed, draws, state = probe.rng(ids)
    # known answers minted from the reference's rnd.h
    assert seed[0] == 3202281099 and seed[1] == 2055759875
    assert np.array_equal(_bits(draws[0]), _bits(np.array([0.538071394, 0.20499754, 0.405853808, 0.0468345881], np.float32)))
    for k, pid in enumerate(ids):
        st = C.c_uint32(lib.orc_pixel_seed(int(pid)))
        assert st.value == seed[k]
        for q in range(4):
            assert np.float32(lib.orc_rnd(C.byref(st))) == draws[k, q]
        assert st.value == state[k]


def test_disk_and_sphere_sampling(probe, lib):
    rng = np.random.default_rng(2)
    states = (rng.integers(1, 2 ** 32, N, dtype=np.uint64).astype(np.uint32)) | 1
    states[0] = 2055759875      # pixel 12345 seed
    disk, sd, sph, ss = probe.disk_sphere(states)
    out = (C.c_float * 3)()
    for k in range(N):
        st = C.c_uint32(int(states[k]))
        lib.orc_random_in_unit_disk(C.byref(st), out)
        assert np.array_equal(_bits(disk[k]), _bits(np.array(out[:], np.float32))) and st.value == sd[k]
        st = C.c_uint32(int(states[k]))
        lib.orc_random_in_unit_sphere(C.byref(st), out)
        assert np.array_equal(_bits(sph[k]), _bits(np.array(out[:], np.float32))) and st.value == ss[k]


@pytest.mark.parametrize("scene", ["c1", "c2"])
def test_get_ray(rt, probe, lib, scene):
    nx, ny = (400, 200) if scene == "c1" else (1200, 800)
    _, _, cam = rt.scene_three_spheres(nx, ny) if scene == "c1" else rt.scene_random_spheres(nx, ny)
    rng = np.random.default_rng(3)
    s = rng.uniform(0, 1, N).astype(np.float32)
    t = rng.uniform(0, 1, N).astype(np.float32)
    states = (rng.integers(1, 2 ** 32, N, dtype=np.uint64).astype(np.uint32)) | 1
    org, d, sa = probe.get_ray(cam, s, t, states)
    o_ = (C.c_float * 3)(); d_ = (C.c_float * 3)()
    for k in range(N):
        st = C.c_uint32(int(states[k]))
        lib.orc_get_ray(C.byref(cam), float(s[k]), float(t[k]), C.byref(st), o_, d_)
        assert np.array_equal(_bits(org[k]), _bits(np.array(o_[:], np.float32))), k
        assert np.array_equal(_bits(d[k]), _bits(np.array(d_[:], np.float32))), k
        assert st.value == sa[k]


def _rays(rng, n):
    org = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    return org, d


def test_sphere_hit(rt, probe, lib):
    rng = np.random.default_rng(4)
    sp = np.zeros(N, rt.sphere_dtype)
    sp["center"] = rng.uniform(-2, 2, (N, 3))
    sp["radius"] = rng.uniform(0.1, 2.5, N)
    org, d = _rays(rng, N)
    # edge cases: origin inside the sphere, grazing rays, huge ground sphere, t window that cuts the near root
    org[:300] = sp["center"][:300] + rng.uniform(-0.05, 0.05, (300, 3)).astype(np.float32)
    sp["center"][300:400] = (0, -1000, -1); sp["radius"][300:400] = 1000
    org[300:400, 1] = np.abs(org[300:400, 1]) * 0.01
    for k in range(400, 700):           # aim exactly at the limb
        c = sp["center"][k]; r = sp["radius"][k]
        to = c - org[k]; perp = np.cross(to, [0.3, 0.9, 0.1]); perp /= np.linalg.norm(perp)
        d[k] = (to + perp * r).astype(np.float32)
    tmin = np.full(N, 0.001, np.float32); tmin[700:900] = 0.01
    tmax = np.full(N, np.finfo(np.float32).max, np.float32); tmax[900:1400] = rng.uniform(0.5, 4, 500)
    got = probe.sphere_hit(sp, org, d, tmin, tmax)
    exp = np.array([lib.orc_sphere_hit(C.byref(rt.sphere.from_buffer_copy(sp[k].tobytes())), _f3(org[k]), _f3(d[k]),
                                       float(tmin[k]), float(tmax[k])) for k in range(N)], np.float32)
    assert np.array_equal(_bits(got), _bits(exp)), np.count_nonzero(_bits(got) != _bits(exp))
    assert (exp < 1e30).sum() > 300        # the table does exercise hits


def test_triangle_hit(rt, probe, lib):
    rng = np.random.default_rng(5)
    tr = np.zeros(N, rt.triangle_dtype)
    tr["v"] = rng.uniform(-2, 2, (N, 3, 3))
    org, d = _rays(rng, N)
    for k in range(0, 1500):            # aim at a point inside / on an edge / at a vertex of the triangle
        w = rng.dirichlet([1, 1, 1]) if k < 1000 else (np.array([0.5, 0.5, 0.0]) if k < 1250 else np.array([1.0, 0.0, 0.0]))
        target = (tr["v"][k] * w[:, None]).sum(0)
        d[k] = (target - org[k]).astype(np.float32)
    d[1500:1600] = tr["v"][1500:1600, 1] - tr["v"][1500:1600, 0]     # parallel to the plane
    tmin = np.full(N, 0.01, np.float32)
    tmax = np.full(N, np.finfo(np.float32).max, np.float32); tmax[2000:2500] = rng.uniform(0.5, 4, 500)
    t, u, v = probe.triangle_hit(tr, org, d, tmin, tmax)
    hu = C.c_float(); hv = C.c_float()
    for k in range(N):
        hu.value = 0.0; hv.value = 0.0
        e = lib.orc_triangle_hit(C.byref(rt.triangle.from_buffer_copy(tr[k].tobytes())), _f3(org[k]), _f3(d[k]),
                                 float(tmin[k]), float(tmax[k]), C.byref(hu), C.byref(hv))
        assert np.float32(e).view(np.uint32) == t[k].view(np.uint32), k
        if e < 1e30:
            assert np.float32(hu.value) == u[k] and np.float32(hv.value) == v[k], k
    assert (t < 1e30).sum() > 500


def test_bbox_slab_tests(probe, lib):
    rng = np.random.default_rng(6)
    lo = rng.uniform(-2, 1, (N, 3)).astype(np.float32)
    hi = lo + rng.uniform(0, 2, (N, 3)).astype(np.float32)
    hi[:200, 1] = lo[:200, 1]                    # flat boxes (axis-aligned quads)
    org, d = _rays(rng, N)
    d[200:400, 0] = 0.0                          # axis-parallel rays: 1/0 = inf, 0*inf = NaN inside the slab test
    d[400:500, 2] = -0.0
    org[500:600] = (lo[500:600] + hi[500:600]) / 2   # origin inside the box
    org[600:700, 0] = lo[600:700, 0]             # origin exactly on a slab plane with dir.x = 0 -> NaN path
    d[600:700, 0] = 0.0
    tmax = np.full(N, np.finfo(np.float32).max, np.float32); tmax[1000:2000] = rng.uniform(0.1, 5, 1000)
    dist, hit = probe.bbox(lo, hi, org, d, tmax)
    for k in range(N):
        e = lib.orc_hit_bbox_dist(_f3(lo[k]), _f3(hi[k]), _f3(org[k]), _f3(d[k]), float(tmax[k]))
        h = lib.orc_hit_bbox(_f3(lo[k]), _f3(hi[k]), _f3(org[k]), _f3(d[k]), float(tmax[k]))
        assert np.float32(e).view(np.uint32) == dist[k].view(np.uint32), k
        assert h == hit[k], k
    assert 100 < hit.sum() < N - 100, hit.sum()


def test_material_scatter(rt, O, probe, lib):
    rng = np.random.default_rng(7)
    normal = rng.normal(size=(N, 3)); normal /= np.linalg.norm(normal, axis=1)[:, None]
    wo = rng.normal(size=(N, 3)); wo /= np.linalg.norm(wo, axis=1)[:, None]
    flip = (wo * normal).sum(1) > 0
    normal[flip] *= -1                            # the normal always faces the ray (kernels.cu:354)
    wo[2000:2300] = -normal[2000:2300] * 0.999 + 0.04 * rng.normal(size=(300, 3))     # near-normal incidence
    wo[2300:2600] -= normal[2300:2600] * (wo[2300:2600] * normal[2300:2600]).sum(1)[:, None] * 0.98  # grazing -> TIR inside
    mats = np.zeros(N, rt.material_dtype)
    mats["type"] = np.arange(N) % 3
    mats["color"] = rng.uniform(0, 1, (N, 3))
    mats["param"] = np.where(mats["type"] == rt.RT_GLASS, 1.5, np.where(np.arange(N) % 2 == 0, 0.0, rng.uniform(0, 0.5, N)))
    mats["texId"] = -1
    inside = (rng.uniform(size=N) < 0.5).astype(np.int32)
    t = rng.uniform(0.01, 10, N).astype(np.float32)
    color = mats["color"].astype(np.float32)
    states = (rng.integers(1, 2 ** 32, N, dtype=np.uint64).astype(np.uint32)) | 1
    wi, thr, flags, tout, sa = probe.scatter(t, normal, inside, wo, mats, color, states)
    sc = O.orc_scatter()
    n32 = normal.astype(np.float32); w32 = wo.astype(np.float32)
    kinds = set()
    for k in range(N):
        st = C.c_uint32(int(states[k]))
        lib.orc_material_scatter(float(t[k]), _f3(n32[k]), int(inside[k]), _f3(w32[k]),
                                 C.byref(rt.material.from_buffer_copy(mats[k].tobytes())), _f3(color[k]), C.byref(st), C.byref(sc))
        assert st.value == sa[k], (k, "rng state")
        assert flags[k] == (sc.specular | (sc.refracted << 1)), k
        assert np.array_equal(_bits(wi[k]), _bits(np.array(sc.wi[:], np.float32))), (k, int(mats["type"][k]))
        assert np.array_equal(_bits(thr[k]), _bits(np.array(sc.throughput[:], np.float32))), k
        assert np.float32(sc.t) == tout[k]
        kinds.add((int(mats["type"][k]), int(flags[k])))
    assert (rt.RT_GLASS, 3) in kinds and (rt.RT_GLASS, 1) in kinds and (rt.RT_METAL, 1) in kinds and (rt.RT_DIFFUSE, 0) in kinds


def test_dormant_presets_on_device(rt, O, probe):
    """SURVEY.md §8 f-4 on the GPU, against the golden vectors minted from the reference's own preset functions.
    Presets without libm calls (coat / diffuse / glossy / glass: kinds 3,4,6,7,8,9) are bit-exact.  The checker (sinf),
    tinted glass (logf, expf) and subsurface (logf, expf) presets go through OCML on the device and glibc on the CPU:
    same decisions and RNG consumption, values within 4e-6 relative."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "presets.npz"))
    n = len(g["t"])
    mats = np.zeros(n, rt.material_dtype)
    mats["type"] = g["kind"]; mats["color"] = 0.5; mats["param"] = 1.5; mats["texId"] = -1
    color = np.full((n, 3), 0.5, np.float32)
    wi, thr, flags, tout, sa = probe.scatter(g["t"], g["normal"], g["inside"], g["wo"], mats, color, g["st_in"], hit_point=g["p"])
    exact = np.isin(g["kind"], [3, 4, 6, 7, 8, 9])
    assert np.array_equal(sa, g["st_out"]) and np.array_equal(flags, g["flags"])
    assert np.array_equal(_bits(wi[exact]), _bits(g["wi"][exact]))
    assert np.array_equal(_bits(thr[exact]), _bits(g["throughput"][exact]))
    assert np.array_equal(_bits(tout[exact]), _bits(g["t_out"][exact]))
    loose = ~exact
    assert np.allclose(wi[loose], g["wi"][loose], rtol=4e-6, atol=1e-7)
    assert np.allclose(thr[loose], g["throughput"][loose], rtol=4e-6, atol=1e-7)
    assert np.allclose(tout[loose], g["t_out"][loose], rtol=4e-6, atol=1e-7)


def _ulp_diff(a, b):
    """distance in units in the last place between float32 arrays of equal sign (monotone integer view)"""
    ia = _bits(a).astype(np.int64); ib = _bits(b).astype(np.int64)
    ia = np.where(ia & 0x80000000, 0x80000000 - ia, ia); ib = np.where(ib & 0x80000000, 0x80000000 - ib, ib)
    return np.abs(ia - ib)


def test_sincos_is_the_cpu_libm(probe, lib):
    """The device's sinf / cosf (rt_glibc_sincosf.h: glibc's fp64-polynomial algorithm restated) against the libm of the machine the test
    runs on, bit for bit: 2^20 of the 2^24 arguments phi = (float)(2 pi k 2^-24) of generateShadowRay (kernels.cu:378; a stride through all
    of them - tests/test_oracle_golden.py sweeps every one on the host), 2^18 arguments over (-120, 120), the tiny and the boundary cases."""
    k = (np.arange(1 << 20, dtype=np.int64) * 16 + np.arange(1 << 20) % 16)
    phi = (2.0 * np.pi * (k.astype(np.float32) / np.float32(16777216.0)).astype(np.float64)).astype(np.float32)
    rng = np.random.default_rng(21)
    wide = rng.uniform(-119.99, 119.99, 1 << 18).astype(np.float32)
    edge = np.array([0.0, -0.0, 1e-30, 2.0 ** -12, np.nextafter(np.float32(2.0 ** -12), np.float32(0)), np.pi / 4, np.nextafter(np.float32(np.pi / 4), np.float32(0)),
                     np.pi / 2, np.pi, 2 * np.pi, 119.99, -3.0], np.float32)
    y = np.concatenate([phi, wide, edge])
    s, c = probe.sincos(y)
    ls, lc = np.zeros_like(y), np.zeros_like(y)
    lib.orc_libm_sincosf.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.orc_libm_sincosf(y.ctypes.data, len(y), ls.ctypes.data, lc.ctypes.data)
    assert np.array_equal(_bits(s), _bits(ls)) and np.array_equal(_bits(c), _bits(lc)), (np.count_nonzero(_bits(s) != _bits(ls)), np.count_nonzero(_bits(c) != _bits(lc)))


def test_generate_shadow_ray(rt, probe, O):
    """generateShadowRay as a whole (kernels.cu:363-393) on tabulated (origin, attenuation, normal, rng): every output EXACT - cosAMax, the
    generated / rejected decision, the number of draws, the RNG state, lightDist, shadowDir and lightContribution (round 2 held the two
    vectors to 2 ulp: cosf / sinf were OCML's on the device; they are glibc's algorithm now, rt_glibc_sincosf.h)."""
    rng = np.random.default_rng(15)
    opt = O.default_options(False)
    lc = np.array(opt.light.center.e[:], np.float64)
    org = (rng.uniform(-300, 300, (N, 3)) + (0, 100, 0)).astype(np.float32)
    org[::10] = (lc + rng.normal(size=(N // 10, 3)) * 20.0).astype(np.float32)        # inside the light: cosAMax = NaN, no draws
    att = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    nrm = rng.normal(size=(N, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    to_l = lc - org; to_l /= np.linalg.norm(to_l, axis=1, keepdims=True)
    nrm[::3] = to_l[::3]
    nrm = nrm.astype(np.float32)
    seeds = (rng.integers(1, 2 ** 32, N, dtype=np.uint64) | 1).astype(np.uint32)
    ok, sdir, lcon, dist, cam, draws, sa = probe.shadow_ray(opt.light, opt.lightColor, org, att, nrm, seeds)
    n_gen = n_nan = n_rej = 0
    for k in range(N):
        e_ok, e_dir, e_con, e_dist, e_cam, e_draws, e_st = O.generate_shadow_ray(opt, org[k], att[k], nrm[k], int(seeds[k]), "orc")
        assert _bits(cam[k]) == _bits(e_cam), k                       # cosAMax exact (NaN included)
        assert draws[k] == e_draws and sa[k] == e_st, k              # draw count and stream position exact
        assert ok[k] == e_ok, k                                       # the dotl > 0 decision
        if e_draws == 0:
            n_nan += 1
            assert ok[k] == 0
        elif e_ok:
            n_gen += 1
            assert _bits(dist[k]) == _bits(e_dist), k
            assert np.array_equal(_bits(sdir[k]), _bits(e_dir)) and np.array_equal(_bits(lcon[k]), _bits(e_con)), k
        else:
            n_rej += 1
    assert n_gen > 500 and n_nan > 100 and n_rej > 100, (n_gen, n_nan, n_rej)


def test_plane_hit(rt, probe, lib):
    rng = np.random.default_rng(16)
    pl = np.zeros((N, 6), np.float32)
    nrm = rng.normal(size=(N, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm[:500] = (0, 1, 0)
    pl[:, 0:3] = nrm; pl[:, 3:6] = rng.uniform(-2, 2, (N, 3))
    org, d = _rays(rng, N)
    d[:100, 1] = 0.0                                                  # parallel to the y = const planes: denom = 0 -> miss
    tmin = np.full(N, 0.01, np.float32); tmax = np.full(N, np.finfo(np.float32).max, np.float32); tmax[1000:1500] = rng.uniform(0.1, 3, 500)
    t = probe.plane_hit(pl, org, d, tmin, tmax)
    for k in range(N):
        p = rt.plane(); p.norm.e[:] = pl[k, 0:3]; p.point.e[:] = pl[k, 3:6]
        e = lib.orc_plane_hit(C.byref(p), _f3(org[k]), _f3(d[k]), float(tmin[k]), float(tmax[k]))
        assert np.float32(e).view(np.uint32) == t[k].view(np.uint32), k
    assert 300 < (t < 1e30).sum() < N - 300
