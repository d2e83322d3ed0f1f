#!/usr/bin/env python3
"""Mints the golden vectors under tests/golden/ from the REFERENCE'S OWN CODE (oracle/_ref/libref.so =
the reference headers compiled where they lie under /root/reference, see oracle/ref_driver.cpp).

Run only in the build container (needs oracle/_ref/libref.so):   python oracle/gen_golden.py
Fixtures are data: seeded inputs and the reference's outputs, as float32/uint32 arrays in .npz files.
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuda_raytracing_optimized_amd as rt  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
N = 400


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def fnv1a64(b):
    h = 0xcbf29ce484222325
    for x in hashlib.sha256(b).digest():      # FNV over the sha256 digest: short, stable, no big loop in Python
        h = ((h ^ x) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def main():
    ref = O.load_ref()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261004)

    # ---- rnd.h -------------------------------------------------------------------------------------------------
    ids = np.concatenate([[0, 1, 12345, 959999, 2 ** 31 + 7], rng.integers(0, 2 ** 32, 60, dtype=np.uint64)]).astype(np.uint32)
    seeds = np.array([ref.ref_pixel_seed(int(i)) for i in ids], np.uint32)
    draws = np.zeros((len(ids), 8), np.float32)
    states = np.zeros((len(ids), 8), np.uint32)
    for k, s in enumerate(seeds):
        st = C.c_uint32(int(s))
        for q in range(8):
            draws[k, q] = ref.ref_rnd(C.byref(st))
            states[k, q] = st.value
    st_in = (rng.integers(1, 2 ** 32, N, dtype=np.uint64).astype(np.uint32)) | 1
    st_in[0] = 2055759875
    disk = np.zeros((N, 3), np.float32); sd = np.zeros(N, np.uint32)
    sph = np.zeros((N, 3), np.float32); ss = np.zeros(N, np.uint32)
    o = (C.c_float * 3)()
    for k in range(N):
        st = C.c_uint32(int(st_in[k])); ref.ref_random_in_unit_disk(C.byref(st), o); disk[k] = o[:]; sd[k] = st.value
        st = C.c_uint32(int(st_in[k])); ref.ref_random_in_unit_sphere(C.byref(st), o); sph[k] = o[:]; ss[k] = st.value
    np.savez_compressed(os.path.join(OUT, "rng.npz"), pixel_ids=ids, seeds=seeds, draws=draws, states=states,
                        st_in=st_in, disk=disk, st_disk=sd, sphere=sph, st_sphere=ss)

    # ---- camera + get_ray ----------------------------------------------------------------------------------------
    cams = [((0, 0, 1), (0, 0, -1), (0, 1, 0), 60.0, 2.0, 0.0, 2.0),
            ((13, 2, 3), (0, 0, 0), (0, 1, 0), 30.0, 1.5, 0.1, 10.0),
            ((5.555139, 173.679901, 494.515045), (5.555139, 173.679901, 493.515045), (0, 1, 0), 42.0, 640.0 / 800.0, 0.0, 1.0),
            ((-2, 2, 1), (0, 0, -1), (0, 1, 0), 20.0, 16.0 / 9.0, 2.0, 3.4641016)]
    cam_in = np.array([list(c[0]) + list(c[1]) + list(c[2]) + list(c[3:]) for c in cams], np.float32)
    cam_out = np.zeros((len(cams), 22), np.float32)
    rays = []
    for k, c in enumerate(cams):
        cam = rt.camera()
        ref.ref_make_camera(f3(c[0]), f3(c[1]), f3(c[2]), c[3], c[4], c[5], c[6], C.byref(cam))
        cam_out[k] = np.frombuffer(bytes(cam), np.float32)
        s = rng.uniform(0, 1, 100).astype(np.float32); t = rng.uniform(0, 1, 100).astype(np.float32)
        sti = (rng.integers(1, 2 ** 32, 100, dtype=np.uint64).astype(np.uint32)) | 1
        org = np.zeros((100, 3), np.float32); d = np.zeros((100, 3), np.float32); sta = np.zeros(100, np.uint32)
        oo = (C.c_float * 3)(); dd = (C.c_float * 3)()
        for q in range(100):
            st = C.c_uint32(int(sti[q]))
            ref.ref_get_ray(C.byref(cam), float(s[q]), float(t[q]), C.byref(st), oo, dd)
            org[q] = oo[:]; d[q] = dd[:]; sta[q] = st.value
        rays.append((s, t, sti, org, d, sta))
    np.savez_compressed(os.path.join(OUT, "camera.npz"), cam_in=cam_in, cam_out=cam_out,
                        s=np.stack([r[0] for r in rays]), t=np.stack([r[1] for r in rays]), st_in=np.stack([r[2] for r in rays]),
                        org=np.stack([r[3] for r in rays]), dir=np.stack([r[4] for r in rays]), st_out=np.stack([r[5] for r in rays]))

    # ---- intersections.h -----------------------------------------------------------------------------------------
    sp = np.zeros(N, rt.sphere_dtype)
    sp["center"] = rng.uniform(-2, 2, (N, 3)); sp["radius"] = rng.uniform(0.1, 2.5, N)
    org = rng.uniform(-3, 3, (N, 3)).astype(np.float32); d = rng.normal(size=(N, 3)).astype(np.float32)
    org[:60] = sp["center"][:60] + rng.uniform(-0.05, 0.05, (60, 3)).astype(np.float32)
    sp["center"][60:90] = (0, -1000, -1); sp["radius"][60:90] = 1000; org[60:90, 1] = np.abs(org[60:90, 1]) * 0.01
    for k in range(90, 160):
        c = sp["center"][k]; r = sp["radius"][k]; to = c - org[k]
        perp = np.cross(to, [0.3, 0.9, 0.1]); perp /= np.linalg.norm(perp); d[k] = (to + perp * r).astype(np.float32)
    tmin = np.full(N, 0.001, np.float32); tmin[200:260] = 0.01
    tmax = np.full(N, np.finfo(np.float32).max, np.float32); tmax[260:340] = rng.uniform(0.5, 4, 80)
    sph_t = np.array([ref.ref_sphere_hit(C.byref(rt.sphere.from_buffer_copy(sp[k].tobytes())), f3(org[k]), f3(d[k]),
                                         float(tmin[k]), float(tmax[k])) for k in range(N)], np.float32)
    tr = np.zeros(N, rt.triangle_dtype); tr["v"] = rng.uniform(-2, 2, (N, 3, 3))
    torg = rng.uniform(-3, 3, (N, 3)).astype(np.float32); td = rng.normal(size=(N, 3)).astype(np.float32)
    for k in range(0, 250):
        w = rng.dirichlet([1, 1, 1]) if k < 180 else (np.array([0.5, 0.5, 0.0]) if k < 215 else np.array([1.0, 0.0, 0.0]))
        td[k] = ((tr["v"][k] * w[:, None]).sum(0) - torg[k]).astype(np.float32)
    td[250:270] = tr["v"][250:270, 1] - tr["v"][250:270, 0]
    ttmax = np.full(N, np.finfo(np.float32).max, np.float32); ttmax[300:360] = rng.uniform(0.5, 4, 60)
    tri_t = np.zeros(N, np.float32); tri_u = np.zeros(N, np.float32); tri_v = np.zeros(N, np.float32)
    hu = C.c_float(); hv = C.c_float()
    for k in range(N):
        hu.value = 0; hv.value = 0
        tri_t[k] = ref.ref_triangle_hit(C.byref(rt.triangle.from_buffer_copy(tr[k].tobytes())), f3(torg[k]), f3(td[k]), 0.01,
                                        float(ttmax[k]), C.byref(hu), C.byref(hv))
        tri_u[k] = hu.value; tri_v[k] = hv.value
    lo = rng.uniform(-2, 1, (N, 3)).astype(np.float32); hi = lo + rng.uniform(0, 2, (N, 3)).astype(np.float32)
    hi[:40, 1] = lo[:40, 1]
    borg = rng.uniform(-3, 3, (N, 3)).astype(np.float32); bd = rng.normal(size=(N, 3)).astype(np.float32)
    bd[40:80, 0] = 0.0; bd[80:100, 2] = -0.0
    borg[100:130] = (lo[100:130] + hi[100:130]) / 2
    borg[130:160, 0] = lo[130:160, 0]; bd[130:160, 0] = 0.0
    btmax = np.full(N, np.finfo(np.float32).max, np.float32); btmax[200:330] = rng.uniform(0.1, 5, 130)
    bdist = np.array([ref.ref_hit_bbox_dist(f3(lo[k]), f3(hi[k]), f3(borg[k]), f3(bd[k]), float(btmax[k])) for k in range(N)], np.float32)
    bhit = np.array([ref.ref_hit_bbox(f3(lo[k]), f3(hi[k]), f3(borg[k]), f3(bd[k]), float(btmax[k])) for k in range(N)], np.int32)
    np.savez_compressed(os.path.join(OUT, "intersections.npz"), spheres=sp, s_org=org, s_dir=d, s_tmin=tmin, s_tmax=tmax, s_t=sph_t,
                        tris=tr, t_org=torg, t_dir=td, t_tmax=ttmax, t_t=tri_t, t_u=tri_u, t_v=tri_v,
                        b_lo=lo, b_hi=hi, b_org=borg, b_dir=bd, b_tmax=btmax, b_dist=bdist, b_hit=bhit)

    # ---- material.h / scene_materials.h ----------------------------------------------------------------------------
    normal = rng.normal(size=(N, 3)); normal /= np.linalg.norm(normal, axis=1)[:, None]
    wo = rng.normal(size=(N, 3)); wo /= np.linalg.norm(wo, axis=1)[:, None]
    flip = (wo * normal).sum(1) > 0; normal[flip] *= -1
    wo[300:340] = -normal[300:340] * 0.999 + 0.04 * rng.normal(size=(40, 3))
    wo[340:400] -= normal[340:400] * (wo[340:400] * normal[340:400]).sum(1)[:, None] * 0.98
    normal = normal.astype(np.float32); wo = wo.astype(np.float32)
    mats = np.zeros(N, rt.material_dtype)
    mats["type"] = np.arange(N) % 3; mats["color"] = rng.uniform(0, 1, (N, 3))
    mats["param"] = np.where(mats["type"] == rt.RT_GLASS, 1.5, np.where(np.arange(N) % 2 == 0, 0.0, rng.uniform(0, 0.5, N)))
    mats["texId"] = -1
    inside = (rng.uniform(size=N) < 0.5).astype(np.int32)
    ht = rng.uniform(0.01, 10, N).astype(np.float32)
    sti = (rng.integers(1, 2 ** 32, N, dtype=np.uint64).astype(np.uint32)) | 1
    wi = np.zeros((N, 3), np.float32); thr = np.zeros((N, 3), np.float32); flags = np.zeros(N, np.int32)
    tout = np.zeros(N, np.float32); sta = np.zeros(N, np.uint32)
    sc = O.orc_scatter()
    for k in range(N):
        st = C.c_uint32(int(sti[k]))
        ref.ref_material_scatter(float(ht[k]), f3(normal[k]), int(inside[k]), f3(wo[k]),
                                 C.byref(rt.material.from_buffer_copy(mats[k].tobytes())), f3(mats["color"][k]), C.byref(st), C.byref(sc))
        wi[k] = sc.wi[:]; thr[k] = sc.throughput[:]; flags[k] = sc.specular | (sc.refracted << 1); tout[k] = sc.t; sta[k] = st.value
    cosv = rng.uniform(0, 1, N).astype(np.float32); idx = np.where(np.arange(N) % 2 == 0, 1.5, 1 / 1.5).astype(np.float32)
    schl = np.array([ref.ref_schlick(float(cosv[k]), float(idx[k])) for k in range(N)], np.float32)
    refl = np.zeros((N, 3), np.float32); refr = np.zeros((N, 3), np.float32)
    for k in range(N):
        ref.ref_reflect(f3(wo[k]), f3(normal[k]), o); refl[k] = o[:]
        ref.ref_refract(f3(wo[k]), f3(normal[k]), float(idx[k]), o); refr[k] = o[:]
    srgb_in = np.concatenate([np.linspace(-0.1, 1.2, 300), rng.uniform(0, 1, 100)]).astype(np.float32)
    srgb = np.array([ref.ref_linear_to_srgb(float(x)) for x in srgb_in], np.uint32)
    np.savez_compressed(os.path.join(OUT, "materials.npz"), t=ht, normal=normal, inside=inside, wo=wo, mats=mats, st_in=sti,
                        wi=wi, throughput=thr, flags=flags, t_out=tout, st_out=sta,
                        cos=cosv, idx=idx, schlick=schl, reflect=refl, refract=refr, srgb_in=srgb_in, srgb=srgb)

    # ---- dormant look presets (scene_materials.h:22-93), called through the reference's own preset functions -----------
    M = 540
    pn = rng.normal(size=(M, 3)); pn /= np.linalg.norm(pn, axis=1)[:, None]
    pw = rng.normal(size=(M, 3)); pw /= np.linalg.norm(pw, axis=1)[:, None]
    fl = (pw * pn).sum(1) > 0; pn[fl] *= -1
    pn = pn.astype(np.float32); pw = pw.astype(np.float32)
    pp = rng.uniform(-30, 30, (M, 3)).astype(np.float32)
    pt = rng.uniform(0.01, 6, M).astype(np.float32)
    pin = (rng.uniform(size=M) < 0.5).astype(np.int32)
    pkind = (3 + np.arange(M) % 9).astype(np.int32)
    pst = (rng.integers(1, 2 ** 32, M, dtype=np.uint64).astype(np.uint32)) | 1
    pwi = np.zeros((M, 3), np.float32); pthr = np.zeros((M, 3), np.float32); pfl = np.zeros(M, np.int32)
    pto = np.zeros(M, np.float32); psa = np.zeros(M, np.uint32)
    for k in range(M):
        st = C.c_uint32(int(pst[k]))
        ref.ref_preset_scatter(int(pkind[k]), float(pt[k]), f3(pp[k]), f3(pn[k]), int(pin[k]), f3(pw[k]), C.byref(st), C.byref(sc))
        pwi[k] = sc.wi[:]; pthr[k] = sc.throughput[:]; pfl[k] = sc.specular | (sc.refracted << 1); pto[k] = sc.t; psa[k] = st.value
    np.savez_compressed(os.path.join(OUT, "presets.npz"), kind=pkind, t=pt, p=pp, normal=pn, inside=pin, wo=pw, st_in=pst,
                        wi=pwi, throughput=pthr, flags=pfl, t_out=pto, st_out=psa)

    # ---- light-sampling expression probes (kernels.cu:378-387) -------------------------------------------------------
    M = 200
    su = rng.normal(size=(M, 3)).astype(np.float32); sv = rng.normal(size=(M, 3)).astype(np.float32); sw = rng.normal(size=(M, 3)).astype(np.float32)
    eps2 = rng.uniform(0, 1, M).astype(np.float32); sinA = rng.uniform(0, 1, M).astype(np.float32); cosA = rng.uniform(0, 1, M).astype(np.float32)
    att = rng.uniform(0, 1, (M, 3)).astype(np.float32); dotl = rng.uniform(0, 1, M).astype(np.float32); cam_ = rng.uniform(0.9, 1, M).astype(np.float32)
    phi = np.array([ref.ref_probe_phi(float(e)) for e in eps2], np.float32)
    ldir = np.zeros((M, 3), np.float32); lcon = np.zeros((M, 3), np.float32)
    for k in range(M):
        ref.ref_probe_light_dir(f3(su[k]), f3(sv[k]), f3(sw[k]), float(phi[k]), float(sinA[k]), float(cosA[k]), o); ldir[k] = o[:]
        ref.ref_probe_light_contribution(f3(att[k]), f3((20, 20, 20)), float(dotl[k]), float(cam_[k]), o); lcon[k] = o[:]
    np.savez_compressed(os.path.join(OUT, "light.npz"), su=su, sv=sv, sw=sw, eps2=eps2, phi=phi, sinA=sinA, cosA=cosA, ldir=ldir,
                        att=att, dotl=dotl, cosAMax=cam_, lcon=lcon)

    # ---- frames: the reference-header host loop ------------------------------------------------------------------------
    frames = {}
    for name, (nx, ny, ns, kw) in {"c1_400x200x1": (400, 200, 1, {}), "c1_200x100x4_rr": (200, 100, 4, {"rr": 1}),
                                   "rs_300x200x2": (300, 200, 2, {}), "rs_96x64x8_counter": (96, 64, 8, {"rng": rt.RT_RNG_COUNTER})}.items():
        sp_, mt_, cam = rt.scene_three_spheres(nx, ny) if name.startswith("c1") else rt.scene_random_spheres(nx, ny)
        opt = O.default_options(True)
        for k, v in kw.items():
            setattr(opt, k, v)
        fb, cnt = O.ref_render_spheres(sp_, mt_, cam, opt, nx, ny, ns, 50, counters=True)
        frames[name + "_sha256"] = np.frombuffer(hashlib.sha256(fb.tobytes()).digest(), np.uint8)
        frames[name + "_crop"] = fb[ny // 2 - 16:ny // 2 + 16, nx // 2 - 24:nx // 2 + 24].copy()
        frames[name + "_strided"] = fb[::9, ::7].copy()
        frames[name + "_counts"] = np.array([cnt.samples, cnt.rays, cnt.prim_tests, cnt.hits], np.uint64)
        frames[name + "_mean"] = fb.mean(axis=(0, 1)).astype(np.float64)
    # the benchmark scene itself (488 spheres + materials + camera at 1200x800) as data
    sp_, mt_, cam = rt.scene_random_spheres(1200, 800)
    frames["rs_scene_spheres"] = sp_; frames["rs_scene_materials"] = mt_
    frames["rs_scene_camera_1200x800"] = np.frombuffer(bytes(cam), np.float32)
    np.savez_compressed(os.path.join(OUT, "frames.npz"), **frames)
    sizes = (C.c_int * 32)()
    n = ref.ref_struct_sizes(sizes, 32)
    np.savez_compressed(os.path.join(OUT, "abi.npz"), struct_sizes=np.array(sizes[:n], np.int32))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
