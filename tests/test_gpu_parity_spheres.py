"""GPU parity, sphere path: the HIP renderer (through the C-ABI) against the CPU oracle on the same
seeded inputs.  PARITY fp mode must be BIT-EXACT (integer RNG + IEEE fp32 with no contraction on
both sides); FAST mode (FMA contraction) is held to a stated tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render_gpu(rt, sp, mt, cam, nx, ny, ns, depth, **opts):
    fb = rt.initRendererSpheres(sp, mt, cam, nx, ny, depth)
    o = rt.getDefaultRenderOptions(True)
    rt.setRenderOptions(o, **opts)
    rt.runRenderer(ns, 8, 8)
    out = np.array(fb, copy=True)
    st = rt.getRenderStats()
    rt.cleanupRenderer()
    return out, st


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("nx,ny,ns", [(400, 200, 1), (400, 200, 4), (67, 45, 3)])
def test_c1_three_spheres_bit_exact(rt, O, nx, ny, ns):
    sp, mt, cam = rt.scene_three_spheres(nx, ny)
    ref, cnt = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 50, counters=True)
    got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50, counters=1)
    assert np.array_equal(_bits(got), _bits(ref)), f"{np.count_nonzero(_bits(got) != _bits(ref))} differing words"
    assert st.rays == cnt.rays and st.prim_tests == cnt.prim_tests


# kernel variants (rt_kernels_spheres.hip): 0 = persistent waves + pixel queue (default), 1 = one tile per wave;
# bits 8..15 workgroups per CU, bits 16..23 cooperative-scan threshold, bits 24..25 work order, bit 26 = culling off
@pytest.mark.parametrize("variant", [0, 1, (1 << 8), (8 << 8), (1 << 24), (2 << 24), (3 << 24), (24 << 16), (65 << 16), (1 << 16) + 1, (1 << 26), (1 << 26) + 1, (1 << 26) + (65 << 16)])
@pytest.mark.parametrize("nx,ny,ns", [(300, 200, 2), (120, 80, 8), (61, 37, 5)])
def test_random_spheres_bit_exact(rt, O, nx, ny, ns, variant):
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    ref, cnt = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 50, counters=True)
    got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50, counters=1, variant=variant)
    assert np.array_equal(_bits(got), _bits(ref)), f"{np.count_nonzero(_bits(got) != _bits(ref))} differing words"
    assert st.rays == cnt.rays


def test_two_dispatch_cost_ordered_frame_bit_exact(rt, O):
    """ns >= 8 with the reference stream renders a frame in two dispatches (2 samples, cost ordering, resume: DESIGN.md 3.2)
    with chain waves, 17 cost lists and boosts.  Same bits as the oracle and as the single-dispatch work orders, every pixel
    written (the device framebuffer is NaN-poisoned before each frame), ray count equal to the oracle's."""
    nx, ny, ns = 480, 320, 12
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    ref, cnt = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 50, counters=True)
    for variant in (0, 1 << 24, 3 << 24, 7 << 27):     # two dispatches; tile-major and centre-ray single dispatch; boost off
        got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50, counters=1, variant=variant)
        assert not np.isnan(got).any(), variant
        assert np.array_equal(_bits(got), _bits(ref)), (variant, np.count_nonzero(_bits(got) != _bits(ref)))
        assert st.rays == cnt.rays, variant


@pytest.mark.parametrize("nx,ny,ns,world", [(480, 320, 12, 1), (61, 37, 9, 1), (96, 64, 10, 1), (333, 200, 8, 3), (1200, 96, 10, 1)])
def test_traffic_forms_of_the_two_dispatch_frame_bit_exact(rt, O, nx, ny, ns, world, monkeypatch):
    """The traffic forms of the two-dispatch frame (RtSphereParams::ord_rec / xcd_queues / p1_tile_major): 32-byte parked records in queue
    order, one set of cost lists and queue counters per XCD (a wave serves the queue of the XCD it runs on, then steals from the others), a tile-major
    first dispatch, and the finished pixels stored into the compact DEVICE framebuffer (RT_FB_DIRECT=0: whole lines leave the L2, the copy engine delivers
    rows) - each alone and all together: the same bits as the oracle, every pixel written (NaN poison), the oracle's ray count.  61x37 launches three
    workgroups and 96x64 six: at most that many XCDs have a wave, the queues of the others - chain lists included (96x64 has a chain pixel on XCD 6's) - are
    emptied by the other XCDs' waves alone."""
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    ref, cnt = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 50, counters=True)
    off = {"RT_ORD_PACKED": "0", "RT_XCD_QUEUES": "0", "RT_P1_TILE": "0", "RT_FB_DIRECT": "1"}
    singles = [{"RT_ORD_PACKED": "1"}, {"RT_XCD_QUEUES": "1"}, {"RT_P1_TILE": "1"}, {"RT_P1_TILE": "2"}, {"RT_FB_DIRECT": "0"}]
    combos = [{}] + singles + [{"RT_XCD_QUEUES": "1", "RT_FB_DIRECT": "0"}, {"RT_ORD_PACKED": "1", "RT_XCD_QUEUES": "1"},
                               {"RT_ORD_PACKED": "1", "RT_XCD_QUEUES": "1", "RT_P1_TILE": "1", "RT_FB_DIRECT": "0"},
                               {"RT_ORD_PACKED": "1", "RT_XCD_QUEUES": "1", "RT_P1_TILE": "2", "RT_FB_DIRECT": "0"},
                               {"RT_ORD_PACKED": "1", "RT_XCD_QUEUES": "1", "RT_P1_TILE": "2", "RT_FB_DIRECT": "1"}]
    for combo in combos:
        for n in off:
            monkeypatch.setenv(n, combo.get(n, off[n]))
        fb = rt.initRendererSpheres(sp, mt, cam, nx, ny, 50)
        o = rt.getDefaultRenderOptions(True)
        rays = 0
        for frame in range(2):                                       # (twice: the queue words and records of the first frame are behind the second)
            fb[:] = 0
            rays = 0
            for r in range(world):
                rt.setRenderOptions(o, counters=1, part_rank=r, part_world=world)
                rt.runRenderer(ns, 8, 8)
                rays += rt.getRenderStats().rays
            got = np.array(fb, copy=True)
            assert not np.isnan(got).any(), (combo, frame)
            assert np.array_equal(_bits(got), _bits(ref)), (combo, frame, np.count_nonzero(_bits(got) != _bits(ref)))
        rt.cleanupRenderer()
        assert rays == cnt.rays, combo
        # the production kernels (no counters: the lean instantiations), same forms
        fb = rt.initRendererSpheres(sp, mt, cam, nx, ny, 50)
        for r in range(world):
            rt.setRenderOptions(o, counters=0, part_rank=r, part_world=world)
            rt.runRenderer(ns, 8, 8)
        got = np.array(fb, copy=True)
        rt.cleanupRenderer()
        assert np.array_equal(_bits(got), _bits(ref)), (combo, "production", np.count_nonzero(_bits(got) != _bits(ref)))


def test_random_spheres_options_bit_exact(rt, O):
    """Russian roulette, constant sky, counter RNG, shallow depth: each option against the oracle."""
    nx, ny, ns = 96, 64, 4
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    for kw in (dict(rr=1), dict(sky=rt.RT_SKY_CONST_GREY), dict(rng=rt.RT_RNG_COUNTER), dict(t_min=0.01)):
        o = O.default_options(True)
        for k, v in kw.items():
            setattr(o, k, v)
        ref, _ = O.render(O.sphere_scene(sp, mt), cam, o, nx, ny, ns, 50)
        got, _ = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50, **kw)
        assert np.array_equal(_bits(got), _bits(ref)), kw
    for depth in (1, 2, 5):
        ref, _ = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, depth)
        got, _ = _render_gpu(rt, sp, mt, cam, nx, ny, ns, depth)
        assert np.array_equal(_bits(got), _bits(ref)), depth


def test_full_size_crop_bit_exact(rt, O):
    """BASELINE config-2 geometry (1200x800, 488 spheres) at 2 spp on the GPU; the oracle renders
    three 32x16 crops of the same frame (it cannot do the whole frame in seconds)."""
    nx, ny, ns = 1200, 800, 2
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50)
    sc = O.sphere_scene(sp, mt)
    for (x0, y0) in ((0, 0), (584, 392), (1168, 784)):
        ref, _ = O.render(sc, cam, O.default_options(True), nx, ny, ns, 50, region=(x0, y0, x0 + 32, y0 + 16))
        assert np.array_equal(_bits(got[y0:y0 + 16, x0:x0 + 32]), _bits(ref[y0:y0 + 16, x0:x0 + 32])), (x0, y0)
    assert st.samples == nx * ny * ns


def test_headline_frame_bit_exact_on_crops_and_random_pixels(rt, O):
    """THE benchmark frame (BASELINE config 2: 1200x800, 488 spheres, 100 spp, depth 50) exactly as bench.py renders it -
    two dispatches, cost lists, chain waves, boosts - against the oracle on five 32x16 crops (two of them over the pixels
    with the longest chains of the frame, 3500-3700 rays each) and on 300 random single pixels."""
    nx, ny, ns = 1200, 800, 100
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50)
    assert not np.isnan(got).any()
    sc = O.sphere_scene(sp, mt)
    opt = O.default_options(True)
    for (x0, y0) in ((728, 300), (1040, 344), (0, 0), (584, 392), (1168, 784)):
        ref, _ = O.render(sc, cam, opt, nx, ny, ns, 50, region=(x0, y0, x0 + 32, y0 + 16))
        assert np.array_equal(_bits(got[y0:y0 + 16, x0:x0 + 32]), _bits(ref[y0:y0 + 16, x0:x0 + 32])), (x0, y0)
    rng = np.random.default_rng(2026)
    for x, y in zip(rng.integers(0, nx, 300), rng.integers(0, ny, 300)):
        ref, _ = O.render(sc, cam, opt, nx, ny, ns, 50, region=(int(x), int(y), int(x) + 1, int(y) + 1))
        assert np.array_equal(_bits(got[y, x]), _bits(ref[y, x])), (x, y)
    assert st.samples == nx * ny * ns


def test_tinted_glass_and_fuzzy_metal_bit_exact(rt, O):
    """Materials the benchmark scene does not contain: tinted glass (throughput * tint on the reflected
    branch only, material.h:80-82) and a strongly fuzzed metal."""
    nx, ny, ns = 160, 96, 4
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    glass = mt["type"] == rt.RT_GLASS
    mt["color"][glass] = (0.9, 0.5, 0.2)
    metal = mt["type"] == rt.RT_METAL
    mt["param"][metal] = 0.45
    ref, _ = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 50)
    got, _ = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50)
    assert np.array_equal(_bits(got), _bits(ref)), f"{np.count_nonzero(_bits(got) != _bits(ref))} differing words"


def test_fast_mode_within_tolerance(rt, O):
    """FAST fp mode (FMA contraction) draws the same RNG stream, but a path tracer is a chaotic map: a
    1-ulp change of a hit point on a radius-0.2 sphere grows by ~distance/radius per bounce, so paths
    that bounce between small spheres decorrelate and, because a pixel owns ONE stream, so do that
    pixel's later samples.  What must hold: most pixels are untouched to rounding noise, and the image
    is the same estimate (error far below the Monte-Carlo noise of the frame).
    Stated tolerance at 300x200x4spp: >= 97 % of channels within 1e-4 absolute of the oracle,
    mean |diff| <= 2e-3, RMSE (main.cpp:117-125) <= 0.03 (the frame's own MC noise is ~0.15)."""
    nx, ny, ns = 300, 200, 4
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    ref, _ = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 50)
    got, _ = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50, fp=rt.RT_FP_FAST)
    close = np.abs(got - ref) <= 1e-4
    print("fast-mode: close fraction", close.mean(), "mean abs", np.abs(got - ref).mean(), "rmse", rt.rmse(got, ref))
    assert close.mean() >= 0.97, close.mean()
    assert np.abs(got - ref).mean() <= 2e-3
    assert rt.rmse(got, ref) <= 0.03


def test_stripe_partition_is_invisible(rt, O):
    """Interleaved row stripes (multi-GPU partition) must not change a single bit: render the image as
    the members of a 2- and a 3-way partition in turn on the one GPU and interleave.  ns = 10 takes the two-dispatch
    cost-ordered path inside every member (whose cost windows then straddle stripe boundaries: scheduling only)."""
    nx, ny = 200, 120
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    for ns, world, rows in ((2, 2, 16), (10, 2, 16), (10, 3, 8)):
        whole, _ = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50)
        parts = [_render_gpu(rt, sp, mt, cam, nx, ny, ns, 50, part_rank=r, part_world=world, stripe_rows=rows)[0] for r in range(world)]
        merged = np.zeros_like(whole)
        for r in range(world):
            for k in range(r, (ny + rows - 1) // rows, world):
                merged[k * rows:(k + 1) * rows] = parts[r][k * rows:(k + 1) * rows]
        assert np.array_equal(_bits(merged), _bits(whole)), (ns, world, rows)
        # rows a member does not own stay untouched (zero)
        assert not parts[0][rows:2 * rows].any() and not parts[1][0:rows].any()
    # more members than stripes: members 1..3 own nothing, render nothing and return (no hang, no fault)
    nx, ny, ns = 40, 8, 9
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    whole, _ = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50)
    parts = [_render_gpu(rt, sp, mt, cam, nx, ny, ns, 50, part_rank=r, part_world=4, stripe_rows=8)[0] for r in range(4)]
    assert np.array_equal(_bits(parts[0]), _bits(whole)) and not any(p.any() for p in parts[1:])


def test_rerun_is_deterministic(rt):
    nx, ny = 128, 64
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    fb = rt.initRendererSpheres(sp, mt, cam, nx, ny, 50)
    rt.runRenderer(3, 8, 8)
    a = np.array(fb, copy=True)
    rt.runRenderer(3, 8, 8)
    b = np.array(fb, copy=True)
    rt.cleanupRenderer()
    assert np.array_equal(_bits(a), _bits(b))


def _random_scene(rt, rng, n, big=1, dup=0):
    sp = np.zeros(n, rt.sphere_dtype)
    mt = np.zeros(n, rt.material_dtype)
    sp["center"] = rng.uniform(-6, 6, (n, 3)) * (1, 0.3, 1)
    sp["radius"] = rng.uniform(0.1, 0.45, n)
    for k in range(min(big, n)):
        sp["center"][k] = (0, -500.5 - k, 0) if k == 0 else rng.uniform(-3, 3, 3)
        sp["radius"][k] = 500 if k == 0 else 1.5
    for k in range(dup):                                   # exact duplicates LATER in the list: the first index must win
        src = big + k
        dst = n - 1 - k
        if dst > src:
            sp[dst] = sp[src]
    mt["type"] = rng.integers(0, 3, n)
    mt["color"] = rng.uniform(0.1, 1, (n, 3))
    mt["param"] = np.where(mt["type"] == rt.RT_GLASS, 1.5, rng.uniform(0, 0.3, n))
    mt["texId"] = -1
    return sp, mt


@pytest.mark.parametrize("n,big,dup", [(1, 1, 0), (2, 0, 0), (17, 1, 3), (33, 0, 5), (64, 2, 0), (65, 1, 8), (300, 3, 20), (1500, 1, 0), (500, 40, 0), (40, 40, 0)])
def test_arbitrary_sphere_scenes_bit_exact(rt, O, n, big, dup):
    """Scene shapes the benchmark does not have: 1 sphere, no big spheres, only a few small ones, group counts
    that are not multiples of 4, > 1024 spheres, and EXACT duplicate spheres with different materials (an exact
    tie in t: the reference's strict `<` keeps the lower index — the Morton re-ordering must not change that)."""
    rng = np.random.default_rng(1000 + n)
    sp, mt = _random_scene(rt, rng, n, big, dup)
    nx, ny, ns = 96, 64, 3
    cam = rt.make_camera((9, 3, 7), (0, 0, 0), (0, 1, 0), 35.0, nx / ny, 0.05, 10.0)
    ref, cnt = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 20, counters=True)
    for variant in (0, 1 << 26, (24 << 16)):
        got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 20, counters=1, variant=variant)
        assert np.array_equal(_bits(got), _bits(ref)), (variant, np.count_nonzero(_bits(got) != _bits(ref)))
        assert st.rays == cnt.rays
    # without the counters the launcher takes the PRODUCTION instantiations (the counting one is the general kernel): the lean kernels where the scene
    # allows them (basic materials, one pass of <= 32 groups behind the 3-axis cell tables)
    got, _ = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 20)
    assert np.array_equal(_bits(got), _bits(ref)), np.count_nonzero(_bits(got) != _bits(ref))


@pytest.mark.parametrize("n,shape", [(488, "volume"), (700, "volume"), (900, "volume"), (1500, "volume"), (488, "plane"), (700, "plane"), (900, "plane"),
                                     (1500, "plane"), (300, "wall"), (200, "column"), (520, "cluster")])
def test_scene_shapes_3d_and_plane_production_kernels(rt, O, n, shape):
    """scenes that are NOT the cover image - small spheres scattered in a volume (no shared slab: the 3-axis cell tables), on a plane, on a
    vertical wall (shared x), in a thin column, in one tight cluster far from a second one - at sizes across the one-pass / multi-pass and full-copy / hybrid
    boundaries, rendered by the production instantiations (no counters) on the two-dispatch path (8 spp) and the single dispatch (2 spp), against the oracle."""
    rng = np.random.default_rng(4000 + n)
    sp = np.zeros(n, rt.sphere_dtype); mt = np.zeros(n, rt.material_dtype)
    c = rng.uniform(-12, 12, (n, 3))
    if shape == "volume": c[:, 1] = rng.uniform(0.3, 7.0, n)
    elif shape == "plane": c[:, 1] = 0.2
    elif shape == "wall": c[:, 0] = 1.0; c[:, 1] = rng.uniform(0.3, 9.0, n)
    elif shape == "column": c[:, 0] = rng.uniform(-0.3, 0.3, n); c[:, 2] = rng.uniform(-0.3, 0.3, n); c[:, 1] = rng.uniform(0.3, 12.0, n)
    else: c = np.where((np.arange(n) % 2 == 0)[:, None], rng.uniform(-1, 1, (n, 3)) + (0, 1.5, 0), rng.uniform(-1, 1, (n, 3)) + (30, 20, -25))
    sp["center"] = c
    sp["radius"] = 0.2 if shape == "plane" else rng.uniform(0.1, 0.3, n)
    sp["center"][0] = (0, -1000, 0); sp["radius"][0] = 1000
    mt["type"] = rng.choice([0, 0, 0, 0, 1, 2], n); mt["color"] = rng.uniform(0.2, 1, (n, 3)); mt["param"] = np.where(mt["type"] == 2, 1.5, 0.2); mt["texId"] = -1
    mt["type"][0] = 0
    nx, ny = 80, 48
    cam = rt.make_camera((13, 2, 3), (0, 0.5, 0), (0, 1, 0), 30.0, nx / ny, 0.1, 10.0)
    for ns in (8, 2):
        ref, _ = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 50)
        got, _ = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50)
        assert np.array_equal(_bits(got), _bits(ref)), (ns, np.count_nonzero(_bits(got) != _bits(ref)))


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_spheres_on_a_plane_share_a_box_slab(rt, O, axis):
    """Equal spheres whose centres lie in a plane normal to `axis`: every group box has the same extent on that axis, which the kernel
    then evaluates once per pass (group_needs_shared<AX>, one instantiation per axis; random scenes take the general loop)."""
    rng = np.random.default_rng(77 + axis)
    n = 200
    sp = np.zeros(n, rt.sphere_dtype)
    mt = np.zeros(n, rt.material_dtype)
    c = rng.uniform(-5, 5, (n, 3))
    c[:, axis] = 0.25
    sp["center"] = c
    sp["radius"] = 0.25
    sp["center"][0] = (0, -500.5, 0); sp["radius"][0] = 500               # one big sphere: handled apart, does not break the shared extent
    mt["type"] = rng.integers(0, 3, n)
    mt["color"] = rng.uniform(0.1, 1, (n, 3))
    mt["param"] = np.where(mt["type"] == rt.RT_GLASS, 1.5, rng.uniform(0, 0.3, n))
    mt["texId"] = -1
    nx, ny, ns = 96, 64, 3
    cam = rt.make_camera((9, 3, 7), (0, 0, 0), (0, 1, 0), 35.0, nx / ny, 0.05, 10.0)
    ref, cnt = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 20, counters=True)
    got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 20, counters=1, variant=0)
    assert np.array_equal(_bits(got), _bits(ref)), np.count_nonzero(_bits(got) != _bits(ref))
    assert st.rays == cnt.rays


@pytest.mark.parametrize("view,n", [("level", 430), ("below", 430), ("inside", 430), ("far", 430), ("steep", 430), ("no_big", 430),
                                    ("level", 1500), ("inside", 1500), ("far", 3000), ("no_big", 3000)])
def test_cell_table_prefilter_of_the_group_boxes(rt, O, view, n, monkeypatch):
    """Spheres resting on a horizontal plane in <= 32 groups (the benchmark's shape): the group box tests run behind the cell-table prefilter
    (group_needs_cells, rt_kernels_spheres.hip).  Views chosen against it: rays that run inside the slab of the small spheres, level with it
    (the slab interval is long, or unbounded when nothing big is hit: end points at infinity), from below the plane, from inside the slab,
    from 2000 units away (the margin m exceeds a cell), straight down, and a scene without any big sphere (no bound at all); 430 spheres (one
    word per cell set, one pass) and 1500 / 3000 (several words, a pass of 32 groups per word).  Bit-exact against the oracle, with the
    prefilter and without (RT_BOX_CELLS=0), ray counts equal."""
    rng = np.random.default_rng(4242)
    sp = np.zeros(n, rt.sphere_dtype)
    mt = np.zeros(n, rt.material_dtype)
    half = 9.0 * (n / 430.0) ** 0.5                      # (the same density at every size; 1500 / 3000 spheres: 94 / 188 groups = 3 / 6 words per cell set)
    c = rng.uniform(-half, half, (n, 3))
    c[:, 1] = 0.2
    sp["center"] = c
    sp["radius"] = 0.2
    first = 0
    if view != "no_big":
        sp["center"][0] = (0, -1000, 0); sp["radius"][0] = 1000
        sp["center"][1] = (0, 1, 0); sp["radius"][1] = 1.0
        first = 2
    mt["type"] = rng.integers(0, 3, n)
    mt["color"] = rng.uniform(0.1, 1, (n, 3))
    mt["param"] = np.where(mt["type"] == rt.RT_GLASS, 1.5, rng.uniform(0, 0.3, n))
    mt["texId"] = -1
    assert first <= 2
    nx, ny, ns, depth = 96, 33, 3, 12
    cams = {
        "level":  rt.make_camera((12, 0.2, 1), (0, 0.2, 0), (0, 1, 0), 30.0, nx / ny, 0.0, 10.0),
        "below":  rt.make_camera((6, -3.0, 2), (0, 0.2, 0), (0, 1, 0), 50.0, nx / ny, 0.0, 10.0),
        "inside": rt.make_camera((0.37, 0.21, 0.11), (5, 0.2, 3), (0, 1, 0), 80.0, nx / ny, 0.0, 1.0),
        "far":    rt.make_camera((2000, 300, 500), (0, 0, 0), (0, 1, 0), 0.6, nx / ny, 0.0, 2000.0),
        "steep":  rt.make_camera((0.5, 30, 0.5), (0, 0, 0), (0, 0, 1), 35.0, nx / ny, 0.1, 30.0),
        "no_big": rt.make_camera((12, 0.2, 1), (0, 0.2, 0), (0, 1, 0), 30.0, nx / ny, 0.0, 10.0),
    }
    cam = cams[view]
    ref, cnt = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, depth, counters=True)
    assert cnt.hits > 0.2 * nx * ny * ns
    boxes = {}
    for cells in ("1", "0"):
        monkeypatch.setenv("RT_BOX_CELLS", cells)
        got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, depth, counters=1, variant=0)
        assert np.array_equal(_bits(got), _bits(ref)), (view, cells, np.count_nonzero(_bits(got) != _bits(ref)))
        assert st.rays == cnt.rays
        boxes[cells] = st.box_tests
    assert boxes["1"] < 0.7 * boxes["0"], boxes            # the prefilter was really in front of the box tests


def test_camera_inside_scene_and_odd_image_sizes(rt, O):
    """Camera inside the sphere cloud (rays start inside group boxes), image sizes that are not multiples of 8."""
    rng = np.random.default_rng(77)
    sp, mt = _random_scene(rt, rng, 200, 1, 0)
    for nx, ny in ((37, 21), (8, 8), (9, 1), (1, 9)):
        cam = rt.make_camera((0.3, 0.4, 0.2), (2, 0.2, 1), (0, 1, 0), 70.0, nx / ny, 0.0, 1.0)
        ref, _ = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, 4, 12)
        got, _ = _render_gpu(rt, sp, mt, cam, nx, ny, 4, 12)
        assert np.array_equal(_bits(got), _bits(ref)), (nx, ny)


def test_camera_inside_big_spheres_and_small_t_min(rt, O):
    """The big spheres (always-tested class) reach the group culling only through a conservative bound of their closest hit
    and are resolved exactly through the candidate list: origins INSIDE big spheres (near root negative, far root accepted),
    grazing rays, more than 32 big spheres (two bound chunks), and t_min values down to 0 (no self-hit rejection margin)."""
    rng = np.random.default_rng(99)
    sp, mt = _random_scene(rt, rng, 260, 1, 0)
    sp["radius"][1:36] = rng.uniform(2.0, 4.0, 35)          # 36 big spheres around and over the camera
    sp["center"][1] = (0.2, 0.3, 0.1); sp["radius"][1] = 3.0
    mt["type"][1] = rt.RT_GLASS; mt["param"][1] = 1.5
    nx, ny, ns = 64, 48, 4
    cam = rt.make_camera((0.3, 0.4, 0.2), (2, 0.2, 1), (0, 1, 0), 70.0, nx / ny, 0.0, 1.0)
    for t_min in (0.001, 1e-5, 0.0, 0.25):
        o = O.default_options(True)
        o.t_min = t_min
        ref, cnt = O.render(O.sphere_scene(sp, mt), cam, o, nx, ny, ns, 12, counters=True)
        got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 12, counters=1, t_min=t_min)
        assert np.array_equal(_bits(got), _bits(ref)), (t_min, np.count_nonzero(_bits(got) != _bits(ref)))
        assert st.rays == cnt.rays, t_min


def test_in_process_multi_device_path(rt, O):
    """rt_render_options.devices[]: several devices inside ONE process, each rendering its interleaved stripes on its
    own stream, gathered with hipMemcpy2DAsync into the one pinned framebuffer.  The GPU box has a single MI355X, so
    the device list names it twice / three times — every code path of the multi-device runtime except peer placement
    is exercised, and the image must not change by a bit."""
    nx, ny, ns = 168, 100, 2          # ny not a multiple of the stripe height
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    ref, _ = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 50)
    for devs, stripe in (([0, 0], 8), ([0, 0, 0], 16), ([0, 0, 0, 0, 0], 8)):
        got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50, devices=devs, stripe_rows=stripe)
        assert np.array_equal(_bits(got), _bits(ref)), devs
        assert st.samples == nx * ny * ns and st.num_launches == len(devs)
    # devices x processes: 2 devices in each of 2 partition members
    parts = [_render_gpu(rt, sp, mt, cam, nx, ny, ns, 50, devices=[0, 0], part_rank=r, part_world=2)[0] for r in range(2)]
    merged = np.zeros_like(ref)
    for r in range(2):
        for k in range((ny + 7) // 8):
            if (k % 4) // 2 == r:                    # effective world 4: rank_eff = part_rank*2 + device
                merged[k * 8:(k + 1) * 8] = parts[r][k * 8:(k + 1) * 8]
    assert np.array_equal(_bits(merged), _bits(ref))


def test_external_framebuffer(rt, O, tmp_path):
    """setExternalFramebuffer: the D2H stripe copies land in caller-owned (here: memory-mapped, as the ranks of
    bench.py share it) memory; only the stripes this partition member owns are written."""
    nx, ny, ns = 120, 72, 2
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    ref, _ = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 50)
    path = str(tmp_path / "fb.npy")
    np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(ny, nx, 3)).flush()
    ext = np.load(path, mmap_mode="r+")
    fb = rt.initRendererSpheres(sp, mt, cam, nx, ny, 50)
    rt.setExternalFramebuffer(ext)
    o = rt.getDefaultRenderOptions(True)
    for r in range(3):
        rt.setRenderOptions(o, part_rank=r, part_world=3)
        rt.runRenderer(ns, 8, 8)
    assert np.array_equal(_bits(np.array(ext)), _bits(ref))
    assert not np.array(fb).any()                 # the library's own framebuffer was not the target
    rt.setExternalFramebuffer(None)
    rt.setRenderOptions(o, part_rank=0, part_world=1)
    rt.runRenderer(ns, 8, 8)
    assert np.array_equal(_bits(np.array(fb)), _bits(ref))
    rt.cleanupRenderer()


def test_external_framebuffer_that_cannot_be_page_locked(rt, O, tmp_path, capfd):
    """setExternalFramebuffer when hipHostRegister refuses the caller's memory (forced: RT_EXT_FB_NO_REGISTER=1; on a node: a locked-memory limit): a
    warning, and the renderer copies this member's stripes from its compact device buffer into the pageable memory instead of exiting - the same image,
    on the two-dispatch frame (ns = 8) and on the single dispatch (ns = 2), for every member of a 3-way partition."""
    import os
    nx, ny = 120, 72
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    path = str(tmp_path / "fb.npy")
    np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(ny, nx, 3)).flush()
    ext = np.load(path, mmap_mode="r+")
    fb = rt.initRendererSpheres(sp, mt, cam, nx, ny, 50)
    os.environ["RT_EXT_FB_NO_REGISTER"] = "1"
    try:
        rt.setExternalFramebuffer(ext)
    finally:
        del os.environ["RT_EXT_FB_NO_REGISTER"]
    assert "could not page-lock" in capfd.readouterr().err
    o = rt.getDefaultRenderOptions(True)
    for ns in (8, 2):
        ref, _ = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 50)
        ext[:] = 0
        for r in range(3):
            rt.setRenderOptions(o, part_rank=r, part_world=3)
            rt.runRenderer(ns, 8, 8)
        assert np.array_equal(_bits(np.array(ext)), _bits(ref)), ns
    assert not np.array(fb).any()
    rt.setExternalFramebuffer(None)
    rt.cleanupRenderer()


def test_config5_image_size_crops(rt, O):
    """BASELINE config-5 geometry (3840x2160, 488 spheres) at 1 spp on one GPU: the oracle checks four 24x16 crops,
    and the frame must be complete (a pixel is black only when its single path ran into maxDepth inside glass:
    a handful per frame; an unwritten stripe or tile would be thousands)."""
    nx, ny, ns = 3840, 2160, 1
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50)
    assert st.samples == nx * ny * ns
    black = got.sum(axis=2) == 0
    assert black.mean() < 1e-3, black.sum()
    ys, xs = np.nonzero(black)
    if len(ys):                      # and the oracle agrees that such a pixel is black
        y, x = int(ys[0]), int(xs[0])
        ref1, _ = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 50, region=(x, y, x + 1, y + 1))
        assert not ref1[y, x].any()
    sc = O.sphere_scene(sp, mt)
    for (x0, y0) in ((0, 0), (1900, 1000), (3816, 2144), (2500, 700)):
        ref, _ = O.render(sc, cam, O.default_options(True), nx, ny, ns, 50, region=(x0, y0, x0 + 24, y0 + 16))
        assert np.array_equal(_bits(got[y0:y0 + 16, x0:x0 + 24]), _bits(ref[y0:y0 + 16, x0:x0 + 24])), (x0, y0)


@pytest.mark.parametrize("spi", [0, 1, 3, 100])
def test_counter_rng_sample_chunks(rt, O, spi):
    """RT_RNG_COUNTER: one RNG stream per (pixel, sample), so a pixel's samples are independent work items and the
    persistent kernel splits them into chunks of `samples_per_item` over different lanes.  Every sample is the oracle's
    sample bit for bit; only the order of the final additions differs (chunk sums are added in chunk order instead of one
    running sum): tolerance 2e-6 relative + 1e-7 absolute per channel (7 float additions of <= 1 ulp each)."""
    nx, ny, ns = 128, 80, 7
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    o = O.default_options(True)
    o.rng = rt.RT_RNG_COUNTER
    ref, cnt = O.render(O.sphere_scene(sp, mt), cam, o, nx, ny, ns, 50, counters=True)
    got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50, rng=rt.RT_RNG_COUNTER, samples_per_item=spi, counters=1)
    assert st.rays == cnt.rays                     # the very same paths
    if spi >= ns:
        assert np.array_equal(_bits(got), _bits(ref))
    else:
        assert np.all(np.abs(got - ref) <= 2e-6 * np.abs(ref) + 1e-7)


def test_preset_materials_in_a_frame(rt, O):
    """The dormant presets as sphere materials in a whole frame.  Presets without libm calls: bit-exact frame.
    With checker / tinted glass / subsurface spheres: >= 99.5 % of channels within 1e-5 relative (OCML vs glibc ulps)."""
    nx, ny, ns = 128, 80, 4
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    small = np.arange(1, 485)
    exact_kinds = np.array([rt.RT_FLOOR_COAT, rt.RT_FLOOR_DIFFUSE, rt.RT_MODEL_COAT, rt.RT_MODEL_DIFFUSE, rt.RT_MODEL_GLOSSY, rt.RT_MODEL_GLASS])
    m1 = mt.copy()
    m1["type"][small] = exact_kinds[small % len(exact_kinds)]
    m1["type"][485] = rt.RT_MODEL_GLASS
    ref, _ = O.render(O.sphere_scene(sp, m1), cam, O.default_options(True), nx, ny, ns, 50)
    got, _ = _render_gpu(rt, sp, m1, cam, nx, ny, ns, 50)
    assert np.array_equal(_bits(got), _bits(ref))
    m2 = mt.copy()
    m2["type"][0] = rt.RT_FLOOR_CHECKER
    m2["type"][485] = rt.RT_MODEL_SSS
    m2["type"][small[::3]] = rt.RT_MODEL_TINTEDGLASS
    ref, _ = O.render(O.sphere_scene(sp, m2), cam, O.default_options(True), nx, ny, ns, 50)
    got, _ = _render_gpu(rt, sp, m2, cam, nx, ny, ns, 50)
    rel = np.abs(got - ref) <= 1e-5 * np.maximum(np.abs(ref), 1e-3)
    assert rel.mean() >= 0.995, rel.mean()


def _big_scene(rt, rng, n):
    sp = np.zeros(n, rt.sphere_dtype)
    mt = np.zeros(n, rt.material_dtype)
    sp["center"] = rng.uniform(-20, 20, (n, 3)) * (1, 0.2, 1)
    sp["radius"] = rng.uniform(0.1, 0.4, n)
    mt["type"] = rng.integers(0, 3, n)
    mt["color"] = rng.uniform(0.1, 1, (n, 3))
    mt["param"] = np.where(mt["type"] == rt.RT_GLASS, 1.5, 0.1)
    mt["texId"] = -1
    return sp, mt


@pytest.mark.parametrize("n,nx,ny,ns", [(2100, 96, 64, 9), (4000, 96, 64, 4), (20000, 48, 32, 4)])
def test_scene_sizes_up_to_20000_spheres(rt, O, n, nx, ny, ns):
    """Scenes of any size (the reference's scan takes any list, intersections.h:85-104), one per scene form of the persistent kernel beyond the full LDS
    copy: 2100 spheres - the hybrid copy (what a sphere test reads in the LDS, what only a hit reads in global memory) under a 16-wave workgroup;
    4000 - the hybrid copy under an 8-wave workgroup (its 44 KB of scratch leave room for 96 KB of test data); 20000 - every array read from global
    memory, window-relative pair entries beyond 1024 groups.  All single dispatch.  Bit-exact against the oracle, equal ray counts, culling on and off."""
    rng = np.random.default_rng(5 + n)
    sp, mt = _big_scene(rt, rng, n)
    cam = rt.make_camera((9, 3, 7), (0, 0, 0), (0, 1, 0), 35.0, nx / ny, 0.05, 10.0)
    ref, cnt = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 20, counters=True)
    for variant in ((0, 1 << 26) if n <= 4000 else (0,)):
        got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 20, counters=1, variant=variant)
        assert np.array_equal(_bits(got), _bits(ref)), (n, variant, np.count_nonzero(_bits(got) != _bits(ref)))
        assert st.rays == cnt.rays
    if n > 2100:                                                                       # the counter stream (sample chunks) on the global path too
        o = O.default_options(True)
        o.rng = rt.RT_RNG_COUNTER
        ref2, cnt2 = O.render(O.sphere_scene(sp, mt), cam, o, nx, ny, ns, 20, counters=True)
        got2, st2 = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 20, rng=rt.RT_RNG_COUNTER, samples_per_item=2, counters=1)
        assert st2.rays == cnt2.rays                                                   # the very same paths; chunk sums are added in chunk order:
        assert np.all(np.abs(got2 - ref2) <= 2e-6 * np.abs(ref2) + 1e-7), n            # the tolerance of test_counter_rng_sample_chunks


def test_empty_scene_is_refused(rt):
    """The refusal follows the kernels.cu:27-38 convention - message on stderr, exit(99) - instead of a launch failure."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r); import cuda_raytracing_optimized_amd as rt; "
            "sp = np.zeros(0, rt.sphere_dtype); mt = np.zeros(0, rt.material_dtype); "
            "cam = rt.make_camera((9, 3, 7), (0, 0, 0), (0, 1, 0), 35.0, 1.5, 0.05, 10.0); "
            "rt.initRendererSpheres(sp, mt, cam, 96, 64, 20)" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 99 and "empty scene" in r.stderr, (r.returncode, r.stderr[-300:])


@pytest.mark.parametrize("dist,vfov", [(300.0, 2.5), (3000.0, 0.25), (30000.0, 0.025)])
def test_far_camera_culling_is_still_exact(rt, O, dist, vfov):
    """Camera at 10x .. 1000x the scene extent.  Far from the scene the reference's OWN fp32 discriminant b*b - a*c
    (intersections.h:87-91) is so inexact (error ~ eps |oc|^2) that it reports hits for rays that geometrically miss a sphere by
    more than any fixed box inflation - the culling must keep those groups (per-ray margin, make_box_ray): bit-exact against
    the oracle, and equal to the brute-force scan (variant bit 26), at every distance."""
    nx, ny, ns = 96, 64, 3
    sp, mt, _ = rt.scene_random_spheres(nx, ny)
    d = np.array([13.0, 2.0, 3.0]); d /= np.linalg.norm(d)
    cam = rt.make_camera(tuple(d * dist), (0, 0, 0), (0, 1, 0), vfov, nx / ny, 0.0, dist)
    ref, cnt = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 50, counters=True)
    assert cnt.hits > 0.5 * nx * ny * ns                       # the scene fills the frame
    for variant in (0, 1 << 26):
        got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, 50, counters=1, variant=variant)
        assert np.array_equal(_bits(got), _bits(ref)), (variant, np.count_nonzero(_bits(got) != _bits(ref)))
        assert st.rays == cnt.rays


@pytest.mark.parametrize("scene,rr", [("three", 0), ("random", 0), ("random", 1)])
def test_reference_stats_counters_for_sphere_scenes(rt, O, scene, rr):
    """The reference's `#ifdef STATS` ray statistics (kernels.cu:47-67,399-432,514-531) on SPHERE scenes, getRenderStats().ref_stats against the oracle's
    count: primary / secondary rays, hits and misses by kind, low-power rays, paths cut at maxDepth, Russian-roulette kills, NaN samples - through the
    single dispatch (4 spp) and the two-dispatch cost-ordered frame (12 spp).  maxDepth 6 makes EXCEED_MAX_BOUNCE common."""
    nx, ny = 160, 96
    sp, mt, cam = rt.scene_three_spheres(nx, ny) if scene == "three" else rt.scene_random_spheres(nx, ny)
    for ns, depth in ((4, 50), (12, 6)):
        o = O.default_options(True)
        o.rr = rr
        ref, cnt = O.render(O.sphere_scene(sp, mt), cam, o, nx, ny, ns, depth, counters=True)
        got, st = _render_gpu(rt, sp, mt, cam, nx, ny, ns, depth, counters=1, rr=rr)
        assert np.array_equal(_bits(got), _bits(ref))
        for k in range(18):
            assert int(st.ref_stats[k]) == int(cnt.ref_stats[k]), (ns, depth, rt.RT_STAT_NAMES[k], int(st.ref_stats[k]), int(cnt.ref_stats[k]))
        assert st.ref_stats[rt.RT_STAT_PRIMARY] == nx * ny * ns and st.ref_stats[rt.RT_STAT_PRIMARY] + st.ref_stats[rt.RT_STAT_SECONDARY] == st.rays
        assert st.ref_stats[rt.RT_STAT_SECONDARY_MESH] == st.ref_stats[rt.RT_STAT_SECONDARY] and st.ref_stats[rt.RT_STAT_SECONDARY_NOHIT] == 0
        if depth == 6:
            assert st.ref_stats[rt.RT_STAT_EXCEED_MAX_BOUNCE] > 0
        if rr and depth == 50:
            assert st.ref_stats[rt.RT_STAT_RUSSIAN_KILL] > 0


def test_statistical_parity_of_the_other_modes_against_a_converged_image(rt):
    """the modes that are NOT bit-exact - the counter RNG stream (another, equally valid sequence of samples) and the FAST fp build (same
    stream, chaotic fp divergence) - are held to the metric main.cpp:108-128 uses: RMSE against a converged image.  The converged image is the
    reference stream in the PARITY build at 16384 spp (bit-identical to the oracle / the reference's arithmetic by every other test of this file).
    Every mode's RMSE against it must fall like 1 / sqrt(n) and stay within 13 % of the reference stream's own RMSE at the same n: the modes
    estimate the same image with the same variance.  (The reference stream's n-spp image is a PREFIX of the converged one, an independent stream's is
    not: sqrt(1/n + 1/N) against sqrt(1/n - 1/N) is +6.5 % at n = 1024, N = 16384 by itself.  Measured: 1.3 / 2.4 / 9.3 % for the counter stream,
    0.5 / 0.5 / 0.7 % for the FAST build.)"""
    nx, ny = 160, 96
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    fb = rt.initRendererSpheres(sp, mt, cam, nx, ny, 50)
    o = rt.getDefaultRenderOptions(True)

    def render(ns, **kw):
        rt.setRenderOptions(o, rng=kw.get("rng", rt.RT_RNG_REFERENCE_STREAM), fp=kw.get("fp", rt.RT_FP_PARITY))
        rt.runRenderer(ns, 8, 8)
        return np.array(fb, copy=True)
    converged = render(16384)
    rows = {}
    for name, kw in (("reference", {}), ("counter", dict(rng=rt.RT_RNG_COUNTER)), ("fast", dict(fp=rt.RT_FP_FAST)),
                     ("fast+counter", dict(rng=rt.RT_RNG_COUNTER, fp=rt.RT_FP_FAST))):
        rows[name] = [rt.rmse(render(n, **kw), converged) for n in (64, 256, 1024)]
    rt.cleanupRenderer()
    print("RMSE vs the 16384-spp reference-stream image at 64 / 256 / 1024 spp:", {k: [round(x, 5) for x in v] for k, v in rows.items()})
    ref = rows["reference"]
    assert 1.7 < ref[0] / ref[1] < 2.3 and 1.6 < ref[1] / ref[2] < 2.4           # ~ 1 / sqrt(n) (the 1024-spp image shares its samples with the converged one)
    for name in ("counter", "fast", "fast+counter"):
        for k in range(3):
            assert abs(rows[name][k] / ref[k] - 1.0) < 0.13, (name, k, rows[name][k], ref[k])
