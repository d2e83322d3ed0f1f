"""CPU, world_size 2 (gloo): the N>1 path of bench.py — stripe partition, host-side gather into the framebuffer
shared by the ranks, barrier + max-over-ranks timing.  The stripes are rendered by the CPU oracle here (test
infrastructure standing in for the GPU, which this container does not have); the assembled image must equal the
single-process image bit for bit, because pixel seeds depend on the global pixel id only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nx, ny, ns, out_path):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cuda_raytracing_optimized_amd as rt
    from cuda_raytracing_optimized_amd import multigpu
    from oracle import oracle as O
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    rows = multigpu.stripe_rows(rank, world, ny)
    shared = multigpu.SharedFramebuffer(f"test_{port}", nx, ny, rank, dist.barrier)
    fb = np.zeros((ny, nx, 3), np.float32)
    sc = O.sphere_scene(sp, mt)
    opt = O.default_options(True)
    for k in range(rank, (ny + 7) // 8, world):                      # this rank's stripes only
        O.render(sc, cam, opt, nx, ny, ns, 50, region=(0, k * 8, nx, min(ny, k * 8 + 8)), fb=fb)
    shared.gather(fb, rank, world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        np.save(out_path, np.array(shared.array))
        assert t.item() == float(world)
    shared.close(dist.barrier)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nx,ny", [(2, 64, 44), (3, 40, 64)])
def test_stripes_gather_to_the_single_process_image(tmp_path, world, nx, ny):
    import sys
    sys.path.insert(0, ROOT)
    import cuda_raytracing_optimized_amd as rt
    from oracle import oracle as O
    ns = 2
    out = str(tmp_path / "fb.npy")
    mp.spawn(_worker, args=(world, _free_port(), nx, ny, ns, out), nprocs=world, join=True)
    got = np.load(out)
    sp, mt, cam = rt.scene_random_spheres(nx, ny)
    ref, _ = O.render(O.sphere_scene(sp, mt), cam, O.default_options(True), nx, ny, ns, 50)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_partition_covers_every_row_once():
    import sys
    sys.path.insert(0, ROOT)
    from cuda_raytracing_optimized_amd import multigpu
    for ny in (1, 7, 8, 9, 800, 1128, 2264, 803):
        for world in (1, 2, 3, 4, 8):
            rows = np.concatenate([multigpu.stripe_rows(r, world, ny) for r in range(world)])
            assert sorted(rows.tolist()) == list(range(ny)), (ny, world)
    assert multigpu.image_size(1) == (1200, 800) and multigpu.image_size(4) == (2400, 1600)
    for n in (2, 8):
        nx, ny = multigpu.image_size(n)
        assert nx % 8 == 0 and ny % 8 == 0 and abs(nx * ny / (960000 * n) - 1) < 0.01
