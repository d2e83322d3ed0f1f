"""CPU, world_size 2, 3, 4 and 8 (gloo and /dev/shm): the N>1 path of bench.py.  The workers call bench.run_job() — THE partitioned job
of the benchmark: stripe partition, framebuffer shared by the ranks, barrier-bracketed timing, max / sum over ranks
through bench.DistComm — with a backend that renders the rank's stripes with the CPU oracle (test infrastructure
standing in for the GPU, which this container does not have; on the GPU box the backend is bench.HipBackend).
The assembled image must equal the single-process image bit for bit, because pixel seeds depend on the global
pixel id only, and the whole-job counters must equal the single-process counters."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleBackend:
    """bench.HipBackend's interface on the CPU oracle: renders this rank's stripes, then gathers them on the host."""

    def __init__(self):
        import cuda_raytracing_optimized_amd as rt
        from cuda_raytracing_optimized_amd import multigpu
        from oracle import oracle as O
        self.rt, self.O, self.multigpu = rt, O, multigpu

    def open(self, w, rank, world, shared_fb):
        assert w["kind"] == "spheres"
        self.w, self.rank, self.world, self.shared = w, rank, world, shared_fb
        self.sp, self.mt, self.cam = self.rt.scene_random_spheres(w["nx"], w["ny"])
        self.fb = np.zeros((w["ny"], w["nx"], 3), np.float32)

    def _render(self, spp, counters=False):
        w, O = self.w, self.O
        sc = O.sphere_scene(self.sp, self.mt)
        opt = O.default_options(True)
        rays = tests = 0
        for k in range(self.rank, (w["ny"] + 7) // 8, self.world):              # this rank's stripes only
            _, c = O.render(sc, self.cam, opt, w["nx"], w["ny"], spp, w["depth"],
                            region=(0, k * 8, w["nx"], min(w["ny"], k * 8 + 8)), fb=self.fb, counters=counters)
            if counters:
                rays += c.rays; tests += c.prim_tests
        if self.shared is not None:                                              # host gather: plain memcpy per stripe
            for k in range(self.rank, (w["ny"] + 7) // 8, self.world):
                self.shared[k * 8:k * 8 + 8] = self.fb[k * 8:k * 8 + 8]
        return rays, tests

    def step(self, spp=None):
        self._render(spp or self.w["spp"])
        return 1.0

    def counted(self, spp):
        rays, tests = self._render(spp, counters=True)
        return dict(rays=rays, exec_tests=tests, node_visits=0, prim_tests=tests, box_tests=0, shadow_rays=0, spp=spp)

    def device_sync(self):
        pass

    def image(self):
        return self.fb.copy()

    def close(self):
        pass


def _worker(rank, world, port, w, out_path):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    comm = bench.DistComm(dist, "cpu")
    job = bench.run_job(OracleBackend(), comm, w, steps=2, warmup=1, tag=f"test_{port}", keep_image=True)
    if rank == 0:
        np.save(out_path, job["image"])
        np.save(out_path + ".cnt.npy", np.array([job["counters"]["rays"], job["counters"]["prim_tests"], job["value"], job["elapsed"]]))
    assert job["samples"] == w["nx"] * w["ny"] * w["spp"] and job["steps"] == 2
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nx,ny", [(2, 64, 44), (3, 40, 64), (8, 40, 136)])
def test_bench_run_job_partitions_and_gathers_the_single_process_image(tmp_path, world, nx, ny, rt, O):
    w = dict(kind="spheres", nx=nx, ny=ny, spp=2, depth=50, name="test")
    out = str(tmp_path / "fb.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, out), nprocs=world, join=True)
    got = np.load(out)
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    ref, cnt = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, 2, 50, counters=True)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    rays, tests, value, elapsed = np.load(out + ".cnt.npy")
    assert rays == cnt.rays and tests == cnt.prim_tests                           # SUM over ranks = the whole job
    assert elapsed > 0 and abs(value - nx * ny * 2 * 2 / elapsed / 1e6) < 1e-9 * value
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("rt_fb_test_")]   # the shared framebuffer is unlinked


def _shm_worker(rank, world, port, w, out_path):
    """The same job with bench.py's DEFAULT communication at N > 1: barrier and max / sum through /dev/shm (multigpu.ShmComm), no torch.distributed."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_PORT"] = str(port)
    import bench
    from cuda_raytracing_optimized_amd import multigpu
    comm = multigpu.ShmComm(rank, world, timeout=120.0)
    assert comm.reduce([rank, 10 * rank, 1.5], "sum") == [sum(range(world)), 10 * sum(range(world)), 1.5 * world]
    assert comm.reduce([rank, -rank], "max") == [world - 1, 0]
    job = bench.run_job(OracleBackend(), comm, w, steps=2, warmup=1, tag=f"test_{port}", keep_image=True)
    if rank == 0:
        np.save(out_path, job["image"])
        np.save(out_path + ".cnt.npy", np.array([job["counters"]["rays"], job["counters"]["prim_tests"], job["value"], job["elapsed"]]))
    comm.close()


@pytest.mark.parametrize("world,nx,ny", [(2, 64, 44), (4, 48, 72), (8, 40, 136)])
def test_bench_run_job_over_the_shm_barrier(tmp_path, world, nx, ny, rt, O):
    """bench.py --gpus N as the driver launches it uses no RCCL at all: the framebuffer is gathered on the host and the barrier / reductions
    of the timing go through a /dev/shm file.  Same job, same checks as the gloo variant above, plus bench.verify_gather()."""
    import bench
    w = dict(kind="spheres", nx=nx, ny=ny, spp=2, depth=50, name="test")
    out = str(tmp_path / "fb.npy")
    mp.spawn(_shm_worker, args=(world, _free_port(), w, out), nprocs=world, join=True)
    got = np.load(out)
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    ref, cnt = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, 2, 50, counters=True)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    rays, tests, value, elapsed = np.load(out + ".cnt.npy")
    assert rays == cnt.rays and tests == cnt.prim_tests
    assert elapsed > 0 and abs(value - nx * ny * 2 * 2 / elapsed / 1e6) < 1e-9 * value
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("rt_fb_test_") or f.startswith("rt_comm_")]      # both shared files are unlinked
    # the N > 1 line's image check: the gathered image against a single-GPU render, stripe by stripe
    v = bench.verify_gather(got, ref, world)
    assert v["gather_ok"] is True and v["gather_bad_stripes"] == []
    broken = got.copy()
    broken[8 * 3 + 2, 5, 1] += 1.0                        # one pixel of stripe 3 (rank 3 % world)
    v = bench.verify_gather(broken, ref, world)
    assert v["gather_ok"] is False and v["gather_bad_stripes"] == [3] and v["gather_bad_ranks"] == [3 % world]
    broken = got.copy()
    broken[0, 0, 0] = np.nan                              # a lost pixel (the renderer poisons the framebuffer with NaN before each frame)
    assert bench.verify_gather(broken, broken, world)["gather_ok"] is False


def _dying_worker(rank, world, port, ppid_tag):
    """Rank `world - 1` dies (as the renderer does on a HIP error: exit(99)) after the first barrier; the others must notice within seconds."""
    import sys
    import time
    sys.path.insert(0, ROOT)
    from cuda_raytracing_optimized_amd import multigpu
    comm = multigpu.ShmComm(rank, world, tag=ppid_tag, timeout=60.0)
    comm.barrier()
    if rank == world - 1:
        os._exit(99)
    t0 = time.monotonic()
    try:
        comm.reduce([1.0], "sum")
    except multigpu.ShmComm.PeerDied as e:
        assert f"rank {world - 1}" in str(e) and time.monotonic() - t0 < 10.0
        comm.close()                                    # must not hang either
        os._exit(7)
    os._exit(0)


def test_a_dead_rank_ends_the_job_with_a_message_not_a_timeout():
    """a rank that dies inside the job (rt_fail -> exit(99)) left the others in a 900 s barrier.  Now every waiting rank checks the pids
    of the ranks it waits for: PeerDied within a second, a non-zero exit of every rank, nothing left in /dev/shm."""
    import time
    ctx = mp.get_context("spawn")
    world, tag = 3, f"dying_{os.getpid()}"
    t0 = time.monotonic()
    procs = [ctx.Process(target=_dying_worker, args=(r, world, 0, tag)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
    assert [p.exitcode for p in procs] == [7, 7, 99]
    assert time.monotonic() - t0 < 45.0
    assert not [f for f in os.listdir("/dev/shm") if f == f"rt_comm_{tag}.bin"]


def test_run_job_single_rank_needs_no_shared_framebuffer(rt, O):
    import bench
    w = dict(kind="spheres", nx=32, ny=24, spp=1, depth=50, name="test")
    job = bench.run_job(OracleBackend(), bench.LocalComm(), w, steps=1, warmup=0, keep_image=True)
    sp, mt, cam = rt.scene_random_spheres(32, 24)
    ref, _ = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), 32, 24, 1, 50)
    assert np.array_equal(job["image"].view(np.uint32), ref.view(np.uint32))


def test_partition_covers_every_row_once():
    from cuda_raytracing_optimized_amd import multigpu
    for ny in (1, 7, 8, 9, 800, 1128, 2160, 2264, 803):
        for world in (1, 2, 3, 4, 8):
            rows = np.concatenate([multigpu.stripe_rows(r, world, ny) for r in range(world)])
            assert sorted(rows.tolist()) == list(range(ny)), (ny, world)


def test_bench_workloads_are_the_baseline_configs():
    import json
    import bench
    cfg = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    assert "1200" in cfg[1] and "100spp" in cfg[1] and (bench.WORKLOADS["C2"]["nx"], bench.WORKLOADS["C2"]["ny"], bench.WORKLOADS["C2"]["spp"]) == (1200, 800, 100)
    assert "1000spp" in cfg[2] and bench.WORKLOADS["C3"]["spp"] == 1000
    assert "1920" in cfg[3] and (bench.WORKLOADS["C4"]["nx"], bench.WORKLOADS["C4"]["ny"], bench.WORKLOADS["C4"]["spp"]) == (1920, 1080, 256)
    assert "3840" in cfg[4] and (bench.WORKLOADS["C5"]["nx"], bench.WORKLOADS["C5"]["ny"], bench.WORKLOADS["C5"]["spp"]) == (3840, 2160, 4096)
