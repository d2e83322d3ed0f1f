#!/usr/bin/env python3
"""bench.py — Msamples/s of the render() hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--fp parity|fast]

A "step" is one full frame of the workload rendered through the C-ABI (init* once, then runRenderer
per step; scene and camera already resident in HBM).

  N = 1   headline = BASELINE.json configs[1] (C2): random-spheres (488 spheres), 1200x800, 100 spp,
          maxDepth 50.  The other single-GPU configs of BASELINE.json ride in the same JSON line under
          "other_configs": C3 (1200x800x1000), C4 (staircase mesh 1920x1080x256 through initRenderer,
          NEE + RR, with its own BVH-form roofline) and C5's workload (3840x2160x4096) on this one GPU.
  N > 1   STRONG scaling of a FIXED image: BASELINE.json configs[4] (C5), random-spheres 3840x2160 at
          4096 spp (34 G samples per step: 4.2 s on one GPU, ~0.5 s on eight).  The image is cut into
          interleaved 8-row stripes, rank r renders stripes k = r (mod N) — no collective on the data
          path — and the stripes are gathered on the host: the ranks share one framebuffer in /dev/shm,
          page-locked by every rank, and each rank's finished pixels (stored by the kernel itself, over
          the bus) land in it directly.  The barrier and the max / sum of the timing go through /dev/shm
          too (multigpu.ShmComm): no RCCL anywhere.  The line carries "gather_ok": the shared framebuffer
          equals rank 0's single-GPU render of the same frame bit for bit.  The same partitioned job on C2 (1200x800x100: 2 ms
          of kernel per GPU at N = 8, i.e. launch + D2H) is recorded under "other_configs" for the
          north star's "1200x800x100 at 1/2/4/8 GPUs", and rank 0 renders the whole C5 frame alone
          once ("single_gpu_same_workload") so the line carries its own strong-scaling denominator.
          Launched by the driver as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.

run_job() below is THE partitioned job: stripe partition, shared host framebuffer, barrier-bracketed
timing, max/sum over ranks.  tests/test_multigpu_gloo.py drives this same function on CPU (world sizes
2-4, over the /dev/shm communicator and over gloo) with a backend that renders the stripes with the CPU oracle.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_SPHERES = 488
PEAK_FP32_VALU_TFLOPS = 157.3       # MI355X vector fp32 peak, /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (vector)"
PEAK_L2_GATHER_TBPS = 18.8          # same guide, "Indexed rows: gather into LDS": rows served from the XCDs' L2, chip-wide 16.8-18.8 TB/s
PEAK_HBM_TBPS = 8.0
FLOPS_PER_TEST = 18                 # sphereHit discriminant path with `a` hoisted
FLOPS_PER_RAY = 80                  # per-ray set-up + shading
FLOPS_PER_BOX = 20                  # slab test of one group box (15 flop + compares)

# BASELINE.json configs[1..4]
WORKLOADS = {
    "C2": dict(kind="spheres", nx=1200, ny=800, spp=100, depth=50,
               name="random-spheres (488 spheres, LCG seed 0) 1200x800 100spp maxDepth 50, gradient sky"),
    "C3": dict(kind="spheres", nx=1200, ny=800, spp=1000, depth=50,
               name="random-spheres (488 spheres) 1200x800 1000spp maxDepth 50"),
    "C4": dict(kind="mesh", nx=1920, ny=1080, spp=256, depth=64, detail=4,
               name="procedural staircase mesh (36380 triangles, 32768-node BVH of rtBuildBvh, 5 per leaf) 1920x1080 256spp maxDepth 64, "
                    "NEE + Russian roulette, constant sky (kernels.cu HEAD defaults) through initRenderer"),
    "C5": dict(kind="spheres", nx=3840, ny=2160, spp=4096, depth=50,
               name="random-spheres (488 spheres, LCG seed 0) 3840x2160 4096spp maxDepth 50, gradient sky"),
}


# ------------------------------------------------------------------------------------------------------
# communication: a barrier and max / sum of a few scalars over the ranks.  Default at N > 1: a file in /dev/shm
# (cuda-raytracing-optimized_amd/multigpu.py ShmComm) - the job needs no RCCL on its data path (host gather) and none on
# its control path either.  RT_BENCH_BACKEND=nccl | gloo selects torch.distributed instead (DistComm).
# ------------------------------------------------------------------------------------------------------

class LocalComm:
    rank, world = 0, 1

    def barrier(self):
        pass

    def reduce(self, values, op):
        return list(values)


class DistComm:
    """barrier + max/sum over ranks; the data path (the framebuffer) never goes through here."""

    def __init__(self, dist, device):
        self.dist, self.device = dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def barrier(self):
        self.dist.barrier()

    def reduce(self, values, op):
        import torch
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return t.tolist()


# ------------------------------------------------------------------------------------------------------
# backend: the HIP renderer behind the C-ABI.  (tests/test_multigpu_gloo.py has the CPU-oracle twin.)
# ------------------------------------------------------------------------------------------------------

class HipBackend:
    def __init__(self, fp="parity", rng="reference", variant=0):
        import cuda_raytracing_optimized_amd as rt
        self.rt, self.fp, self.rng, self.variant = rt, fp, rng, variant
        self.opt = None
        self.w = None

    def open(self, w, rank, world, shared_fb):
        rt = self.rt
        self.w = w
        if w["kind"] == "spheres":
            sp, mt, cam = rt.scene_random_spheres(w["nx"], w["ny"])
            self.fb = rt.initRendererSpheres(sp, mt, cam, w["nx"], w["ny"], w["depth"])
            self.opt = rt.getDefaultRenderOptions(True)
        else:
            tris, mats = rt.scene_staircase_procedural(w["detail"])
            hm = rt.HostMesh.build(tris, 5)
            cam = rt.staircase_camera(w["nx"], w["ny"])
            ks, keep = rt.make_kernel_scene(hm, mats)
            self.fb = rt.initRenderer(ks, cam, w["nx"], w["ny"], w["depth"], keepalive=keep)
            self.opt = rt.getDefaultRenderOptions(False)
        rt.setRenderOptions(self.opt, fp=rt.RT_FP_FAST if self.fp == "fast" else rt.RT_FP_PARITY,
                            rng=rt.RT_RNG_COUNTER if self.rng == "counter" else rt.RT_RNG_REFERENCE_STREAM,
                            variant=self.variant, part_rank=rank, part_world=world, stripe_rows=8)
        if shared_fb is not None:
            rt.setExternalFramebuffer(shared_fb)     # every rank delivers its stripes into the one shared framebuffer (copied from its compact device
                                                     # framebuffer behind the kernel; RT_FB_DIRECT=1: finished pixels stored straight into it over the bus)

    def step(self, spp=None):
        self.rt.runRenderer(spp or self.w["spp"], 8, 8)      # blocking: kernel(s) incl. the delivery of this rank's stripes = the host gather
        return self.rt.getRenderStats().kernel_ms

    def counted(self, spp):
        """One untimed run with the device counters on: the inputs of the algorithmic work figures."""
        rt = self.rt
        rt.setRenderOptions(self.opt, counters=1)
        rt.runRenderer(spp, 8, 8)
        st = rt.getRenderStats()
        rt.setRenderOptions(self.opt, counters=0)
        return dict(rays=st.rays, exec_tests=st.exec_tests, node_visits=st.node_visits, prim_tests=st.prim_tests,
                    box_tests=getattr(st, "box_tests", 0), shadow_rays=getattr(st, "shadow_rays", 0), spp=spp)

    def device_sync(self):
        import torch
        torch.cuda.synchronize()

    def image(self):
        return np.array(self.fb, copy=True)

    def close(self):
        self.rt.cleanupRenderer()


# ------------------------------------------------------------------------------------------------------
# THE job: partition -> (shared framebuffer) -> warm-up -> K timed steps between barriers -> max over ranks
# ------------------------------------------------------------------------------------------------------

def run_job(backend, comm, w, steps, warmup, tag="job", count_spp=None, warmup_spp=None, keep_image=False):
    """Renders workload `w` `steps` times on comm.world ranks (interleaved 8-row stripes, host gather into one shared
    framebuffer, no collective on the data path) and returns, on every rank, the whole-job figures:
    elapsed = MAX over ranks of the barrier-bracketed wall time of the K steps, kernel_ms = MAX over ranks of the mean
    HIP-event kernel time, counters = SUM over ranks, scaled from `count_spp` to the workload's spp."""
    from cuda_raytracing_optimized_amd import multigpu
    rank, world = comm.rank, comm.world
    shared = None
    if world > 1:
        shared = multigpu.SharedFramebuffer(tag, w["nx"], w["ny"], rank, comm.barrier)
    try:
        backend.open(w, rank, world, shared.array if shared else None)
        cnt = backend.counted(count_spp or w["spp"])
        for _ in range(warmup):
            backend.step(warmup_spp)
        comm.barrier(); backend.device_sync()
        t0 = time.perf_counter()
        kernel_ms = []
        for _ in range(steps):
            kernel_ms.append(backend.step())
        comm.barrier(); backend.device_sync()
        elapsed = time.perf_counter() - t0
        image = None
        if keep_image:
            comm.barrier()
            image = np.array(shared.array) if shared else backend.image()
        backend.close()
    finally:
        if shared:
            shared.close(comm.barrier)
    scale = w["spp"] / float(cnt["spp"])
    keys = ("rays", "exec_tests", "node_visits", "prim_tests", "box_tests", "shadow_rays")
    mx = comm.reduce([elapsed, float(np.mean(kernel_ms))], "max")
    sm = comm.reduce([cnt[k] * scale for k in keys] + [cnt["rays"], cnt["exec_tests"]], "sum")
    samples = w["nx"] * w["ny"] * w["spp"]
    out = dict(elapsed=mx[0], kernel_ms=mx[1], samples=samples, steps=steps, nx=w["nx"], ny=w["ny"], spp=w["spp"],
               value=samples * steps / mx[0] / 1e6, ms_per_step=mx[0] / steps * 1e3,
               counters={k: sm[i] for i, k in enumerate(keys)}, counted_spp=cnt["spp"],
               exec_tests_per_ray=(sm[-1] / sm[-2]) if sm[-2] else None, image=image)
    return out


def sphere_roofline(job, world, traffic=None):
    """fp32 VALU roofline of the sphere kernel, per launch on one GPU.

    `frac` = flops the kernel EXECUTES / kernel time / peak: 18 x executed sphere tests + 20 x group-box tests + 80 x rays, all three counted on
    the device (an untimed run of the same frame with the counters on; the ray count equals the oracle's).  It cannot exceed 1.
    `effective_frac` is SURVEY.md §8d's figure - the flops the reference's brute-force scan would need for the same rays, rays x (18 x 488 + 80),
    over the same time: a rate of useful work, not of hardware use (the exact culling executes ~33 of the 488 tests per ray), and it may
    exceed 1 (it does on C5)."""
    c = job["counters"]
    rays = c["rays"] / world
    eff_flops = rays * (FLOPS_PER_TEST * N_SPHERES + FLOPS_PER_RAY)
    eff = eff_flops / (job["kernel_ms"] * 1e-3) / 1e12
    ex_flops = (FLOPS_PER_TEST * c["exec_tests"] + FLOPS_PER_BOX * c["box_tests"] + FLOPS_PER_RAY * c["rays"]) / world
    ex = ex_flops / (job["kernel_ms"] * 1e-3) / 1e12
    r = {"bound": "valu", "achieved": ex, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
         "frac": ex / PEAK_FP32_VALU_TFLOPS,
         "frac_definition": "EXECUTED flops / kernel time / peak: 18 x executed sphere tests + 20 x group-box tests + 80 x rays (device counters of "
                            "an untimed run of the same frame, scaled by spp; rays equal the oracle's count)",
         "flops_per_launch": ex_flops,
         "executed_per_ray": {"sphere_tests": job["exec_tests_per_ray"], "box_tests": (c["box_tests"] / c["rays"]) if c["rays"] else None,
                              "of_spheres": N_SPHERES},
         "effective_frac": eff / PEAK_FP32_VALU_TFLOPS, "effective_tflops": eff, "effective_flops_per_launch": eff_flops,
         "effective_definition": "SURVEY.md §8d: the reference's brute-force scan for the same rays, rays x (18 x 488 + 80) flops, over the same "
                                 "kernel time - useful work per second, NOT hardware utilisation (may exceed 1)",
         "kernel": "k_render_spheres_queue", "kernel_ms_avg": job["kernel_ms"],
         "algorithmic_hbm_bytes": job["nx"] * job["ny"] * 12,
         "note": "one frame = one 'launch': with the reference RNG stream the persistent kernel is dispatched twice per frame (first 2 samples, then "
                 "the cost-ordered rest) and kernel_ms_avg is the HIP-event time over both. fp32 VALU bound: no GEMM shape, no MFMA (157.3 TFLOP/s "
                 "is also the dense fp32 MFMA peak of MI355X)"}
    r.update(traffic or {"traffic": None})
    return r


def mesh_roofline(job, world, traffic=None):
    """BVH-path roofline: gather bytes out of L2, 48 B per internal-node visit (the child pair),
    64 B per triangle test, 64 B closest-hit refetch per ray; bound = the L2-served gather rate of the guide."""
    c = job["counters"]
    by = (48.0 * c["node_visits"] + 64.0 * c["prim_tests"] + 64.0 * c["rays"]) / world
    fl = (30.0 * c["node_visits"] + 51.0 * c["prim_tests"] + 150.0 * c["rays"]) / world
    achieved = by / (job["kernel_ms"] * 1e-3) / 1e12
    r = {"bound": "l2-gather", "achieved": achieved, "peak": PEAK_L2_GATHER_TBPS, "unit": "TB/s", "frac": achieved / PEAK_L2_GATHER_TBPS,
         "bytes_per_launch": by, "flops_per_launch": fl, "valu_tflops": fl / (job["kernel_ms"] * 1e-3) / 1e12,
         "kernel": "k_render_mesh_queue", "kernel_ms_avg": job["kernel_ms"],
         "per_sample": {k: c[k] / job["samples"] for k in ("rays", "shadow_rays", "node_visits", "prim_tests")},
         "note": "algorithmic gather bytes = 48 x node visits + 64 x triangle tests + 64 x rays (device counters, equal to the oracle's) / kernel time; "
                 "peak = chip-wide L2-served row-gather rate (MI355X_MICROARCH.md, 'Indexed rows': measured there on 1,152-byte rows, a generous "
                 "denominator for 48-byte records). The rays touch 1.5 MB of node records and 1.7 MB of compact leaf records (48 B per real triangle): "
                 "one XCD's 4 MB L2 holds them - `traffic` (2 x FETCH_SIZE + WRITE_SIZE, the bytes that leave L2) is compared with THESE gather bytes, "
                 "not with the framebuffer"}
    r.update(traffic or {"traffic": None})
    if r.get("traffic"):
        r["traffic_over_algorithmic"] = r["traffic"] / by
    return r


# ------------------------------------------------------------------------------------------------------
# HBM-side traffic: measured by tools/measure_traffic.py (rocprofv3 FETCH_SIZE / WRITE_SIZE in separate passes
# around tools/one_frame.py) and committed as profiles/traffic.json TOGETHER with the hash of the kernel sources
# it was taken on; reported only while the sources still hash to that value (a changed kernel -> traffic null).
# ------------------------------------------------------------------------------------------------------

def kernel_source_hash():
    """Hash of the CODE of the kernel sources (// comments and blank space do not count)."""
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cuda-raytracing-optimized_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            txt = open(os.path.join(d, f), encoding="utf-8", errors="replace").read()
            code = "\n".join(ln for ln in (re.sub(r"\s+", " ", re.sub(r"//.*", "", ln)).strip() for ln in txt.splitlines()) if ln)
            h.update(f.encode()); h.update(code.encode())
    return h.hexdigest()[:16]


def committed_traffic(config, algorithmic_bytes):
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(p):
        return {"traffic": None, "traffic_source": "profiles/traffic.json absent"}
    t = json.load(open(p))
    if t.get("kernel_source_hash") != kernel_source_hash():
        return {"traffic": None, "traffic_source": "profiles/traffic.json was taken on other kernel sources (hash mismatch): not reported"}
    e = t.get(config)
    if not e:
        return {"traffic": None, "traffic_source": f"profiles/traffic.json has no entry for {config}"}
    out = {"traffic": e["bytes_per_frame"], "traffic_unit": "bytes/frame", "traffic_over_algorithmic": e["bytes_per_frame"] / algorithmic_bytes,
           "traffic_source": f"committed_profile: profiles/traffic.json ({e.get('how', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2')}), "
                             f"kernel sources {t['kernel_source_hash']}"}
    if e.get("pmc"):          # the hardware's own view of the same frame (SQ counters of the committed PMC passes; profiles/<tag>_<config>_summary.txt)
        out["hardware"] = e["pmc"]
    return out


def cpu_baseline(w):
    """The reference's own header-only code as a single-threaded host loop (oracle/_ref), or our C
    restatement of it when the shim is absent, timed on a bounded sample of the SAME workload:
    the full frame, first CPU_SPP samples of every pixel stream."""
    import cuda_raytracing_optimized_amd as rt
    from oracle import oracle as O
    nx, ny = w["nx"], w["ny"]
    cpu_spp = int(os.environ.get("RT_BENCH_CPU_SPP", "8"))       # ~13 s of single-threaded CPU work
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    opt = O.default_options(True)
    t0 = time.perf_counter()
    if O.have_ref():
        O.ref_render_spheres(sp, mt, cam, opt, nx, ny, cpu_spp, w["depth"])
        kind = "reference"
    else:
        O.render(O.sphere_scene(sp, mt), cam, opt, nx, ny, cpu_spp, w["depth"])
        kind = "port"
    dt = time.perf_counter() - t0
    samples = nx * ny * cpu_spp
    return {"value": samples / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": kind,
            "sample": f"{nx}x{ny} full frame, first {cpu_spp} spp of every pixel stream "
                      f"({samples / 1e6:.2f} M of the {nx * ny * w['spp'] / 1e6:.0f} M samples), {dt:.1f} s single-threaded",
            "host_cpus": os.cpu_count()}


def verify_gather(gathered, single, world, stripe=8):
    """The image the ranks of an N > 1 job delivered into the shared framebuffer against the same frame rendered by ONE GPU: every stripe of
    every rank, bit for bit (pixel seeds depend on the global pixel id only: the partition must be invisible)."""
    g, s1 = np.ascontiguousarray(gathered).view(np.uint32), np.ascontiguousarray(single).view(np.uint32)
    same_rows = (g == s1).all(axis=(1, 2))
    ny = g.shape[0]
    bad_stripes = sorted({int(r) // stripe for r in np.nonzero(~same_rows)[0]})
    return {"gather_ok": bool(same_rows.all()) and not bool(np.isnan(gathered).any()),
            "gather_check": f"whole {g.shape[1]}x{ny} image of the {world}-rank job, all {(ny + stripe - 1) // stripe} stripes, bit for bit against the "
                            f"single-GPU render of the same frame (other_configs.single_gpu_same_workload); NaN-free",
            "gather_bad_stripes": bad_stripes[:16], "gather_bad_ranks": sorted({k % world for k in bad_stripes})}


def brief(job, extra=None):
    d = {"value": job["value"], "unit": "Msamples/s", "ms_per_step": job["ms_per_step"], "frame_ms_kernel": job["kernel_ms"],
         "steps": job["steps"], "rays_per_sample": job["counters"]["rays"] / job["samples"]}
    d.update(extra or {})
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--fp", choices=["parity", "fast"], default=os.environ.get("RT_BENCH_FP", "parity"))
    ap.add_argument("--variant", type=int, default=int(os.environ.get("RT_BENCH_VARIANT", "0")))
    ap.add_argument("--rng", choices=["reference", "counter"], default="reference",
                    help="reference = the reference's per-pixel xorshift stream (the contract workload); counter = per-sample "
                         "stream, samples split over lanes (a different, equally valid estimate of the same image)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--spp", type=int, default=0, help="experiments only: overrides the headline workload's spp")
    ap.add_argument("--max-depth", type=int, default=0, help="experiments only: overrides the headline workload's maxDepth")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the render path has no CPU fallback")
    # One rank per GPU (on a box with fewer GPUs than ranks - a rehearsal - the ranks share the GPUs that exist, round robin).
    backend_name = os.environ.get("RT_BENCH_BACKEND", "shm")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    comm = LocalComm()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend_name == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
            comm = DistComm(dist, "cuda")
        elif backend_name == "gloo":
            dist.init_process_group(backend="gloo")
            comm = DistComm(dist, "cpu")
        else:
            from cuda_raytracing_optimized_amd import multigpu
            comm = multigpu.ShmComm(rank, world)

    hip = HipBackend(args.fp, args.rng, args.variant)
    tag = f"bench_{os.environ.get('MASTER_PORT', '0')}"
    mode = f"{args.rng} RNG stream, fp {args.fp}"
    default_mode = args.rng == "reference" and args.fp == "parity" and args.variant == 0

    def with_overrides(w):
        w = dict(w)
        if args.spp:
            w["spp"] = args.spp
        if args.max_depth:
            w["depth"] = args.max_depth
        return w

    others = {}
    gather = None
    if world == 1:
        w = with_overrides(WORKLOADS["C2"])
        job = run_job(hip, comm, w, args.steps, args.warmup, tag)
        metric = "Msamples/s (pixels x spp / s), random-spheres 1200x800x100spp"
        scaling = "weak"                     # one GPU: nothing is partitioned (the N > 1 lines say "strong")
        partition = "single GPU"
        roof = sphere_roofline(job, 1, committed_traffic("C2", w["nx"] * w["ny"] * 12) if default_mode else None)
        if not args.no_other_configs:
            # C3: the same frame at 1000 spp
            j3 = run_job(hip, comm, WORKLOADS["C3"], 2, 1, tag, count_spp=8, warmup_spp=16)
            others["C3"] = brief(j3, {"workload": WORKLOADS["C3"]["name"] + ", " + mode,
                                      "roofline": {k: v for k, v in sphere_roofline(j3, 1, committed_traffic("C3", 1200 * 800 * 12) if default_mode else None).items()
                                                   if k in ("bound", "achieved", "peak", "unit", "frac", "effective_frac", "executed_per_ray", "traffic", "traffic_over_algorithmic", "traffic_source")}})
            # C4: triangle mesh + BVH through the reference's own entry point
            j4 = run_job(HipBackend(args.fp, "reference", 0), comm, WORKLOADS["C4"], 3, 1, tag, count_spp=4, warmup_spp=16)
            w4 = WORKLOADS["C4"]
            others["C4"] = brief(j4, {"workload": w4["name"] + f", fp {args.fp}",
                                      "roofline": mesh_roofline(j4, 1, committed_traffic("C4", w4["nx"] * w4["ny"] * 12) if args.fp == "parity" else None)})
            # C5's workload on this one GPU (the denominator of the N > 1 strong-scaling lines)
            j5 = run_job(hip, comm, WORKLOADS["C5"], 1, 1, tag, count_spp=4, warmup_spp=16)
            others["C5_one_gpu"] = brief(j5, {"workload": WORKLOADS["C5"]["name"] + ", " + mode + "; the whole frame on ONE GPU",
                                              "roofline": {k: v for k, v in sphere_roofline(j5, 1, committed_traffic("C5", 3840 * 2160 * 12) if default_mode else None).items()
                                                           if k in ("bound", "achieved", "peak", "unit", "frac", "effective_frac", "executed_per_ray", "traffic", "traffic_over_algorithmic", "traffic_source")}})
    else:
        w = with_overrides(WORKLOADS["C5"])
        job = run_job(hip, comm, w, args.steps, args.warmup, tag, count_spp=4, warmup_spp=16, keep_image=True)
        metric = f"Msamples/s (pixels x spp / s), random-spheres 3840x2160x{w['spp']}spp split over {world} GPUs (BASELINE.json configs[4])"
        scaling = "strong"
        partition = f"{world} x interleaved 8-row stripes of the fixed 3840x2160 image, host gather into one shared pinned framebuffer, no collective"
        roof = sphere_roofline(job, world)
        if not args.no_other_configs:
            # the north star's "1200x800x100 at 1/2/4/8 GPUs": same partitioned job on C2
            j2 = run_job(hip, comm, WORKLOADS["C2"], max(args.steps, 5), 1, tag + "_c2")
            others["C2_partitioned"] = brief(j2, {"workload": WORKLOADS["C2"]["name"] + ", " + mode + f"; the fixed 1200x800 image over {world} GPUs",
                                                  "note": "strong scaling of a 16 ms frame: per-GPU kernel time is a few ms, the rest is launch, D2H and barrier"})
            # the mesh / BVH path shards by the same stripes (tests/test_gpu_parity_mesh.py: the partition is invisible): C4 over the ranks
            j4 = run_job(HipBackend(args.fp, "reference", 0), comm, WORKLOADS["C4"], 2, 1, tag + "_c4", count_spp=4, warmup_spp=16)
            others["C4_partitioned"] = brief(j4, {"workload": WORKLOADS["C4"]["name"] + f", fp {args.fp}; the fixed 1920x1080 image over {world} GPUs"})
            # the same C5 frame on rank 0's GPU alone: this line's own strong-scaling denominator
            solo = None
            if rank == 0:
                js = run_job(hip, LocalComm(), w, 1, 1, tag + "_solo", count_spp=4, warmup_spp=16, keep_image=True)
                solo = brief(js, {"workload": w["name"] + "; whole frame on rank 0's GPU alone while the other ranks wait"})
                gather = verify_gather(job["image"], js["image"], world)
            comm.barrier()
            if solo:
                others["single_gpu_same_workload"] = solo

    if rank == 0:
        out = {
            "metric": metric,
            "value": job["value"], "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": job["ms_per_step"], "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["name"] + ", " + mode, "image": [w["nx"], w["ny"]], "spp": w["spp"], "max_depth": w["depth"],
                       "spheres": N_SPHERES, "partition": partition, "fp_mode": args.fp, "rng": args.rng, "kernel_variant": args.variant},
            "frame_ms_kernel": job["kernel_ms"],
            "rays_per_sample": job["counters"]["rays"] / job["samples"],
            "executed_sphere_tests_per_ray": job["exec_tests_per_ray"],
            "roofline": roof,
        }
        if world > 1:
            out.update(gather if gather else {"gather_ok": None, "gather_check": "not run (--no-other-configs: no single-GPU render to compare with)"})
        if others:
            out["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out), flush=True)

    if world > 1:
        if isinstance(comm, DistComm):
            dist.destroy_process_group()
        else:
            comm.close()


if __name__ == "__main__":
    main()
