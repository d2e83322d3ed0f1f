"""Multi-GPU plumbing for one node: the image is cut into interleaved row stripes, one process per GPU renders
its stripes, and the stripes are gathered ON THE HOST into one framebuffer shared by the ranks (a memory-mapped
file in /dev/shm) — no collective on the data path.  torch.distributed is used only for the
barrier / max-over-ranks timing the bench contract asks for.

Nothing here touches the GPU: the same code runs in the world-size-2 gloo tests on CPU."""
import os

import numpy as np

STRIPE_ROWS = 8


def image_size(n_gpus, base=(1200, 800)):
    """Weak-scaling workload: 3:2 image with ~base pixels PER GPU, both sides multiples of 8."""
    if n_gpus == 1:
        return base
    nx = int(round(base[0] * n_gpus ** 0.5 / 8.0)) * 8
    ny = int(round(nx * base[1] / base[0] / 8.0)) * 8
    return nx, ny


def stripe_rows(rank, world, ny, stripe=STRIPE_ROWS):
    """Global row indices owned by `rank`: stripes k = rank, rank+world, ... of `stripe` rows each
    (the same partition rt_render_options.part_rank/part_world/stripe_rows select in the renderer)."""
    nstripes = (ny + stripe - 1) // stripe
    rows = [np.arange(k * stripe, min(ny, k * stripe + stripe)) for k in range(rank, nstripes, world)]
    return np.concatenate(rows) if rows else np.zeros(0, np.int64)


class SharedFramebuffer:
    """One (ny, nx, 3) float32 framebuffer visible to every rank of the node."""

    def __init__(self, tag, nx, ny, rank, barrier):
        self.path = f"/dev/shm/rt_fb_{tag}.npy"
        self.rank = rank
        if rank == 0:
            np.lib.format.open_memmap(self.path, mode="w+", dtype=np.float32, shape=(ny, nx, 3)).flush()
        barrier()
        self.array = np.load(self.path, mmap_mode="r+")

    def gather(self, fb, rank, world, stripe=STRIPE_ROWS):
        """Host-side gather of this rank's stripes: one contiguous memcpy per stripe into the shared mapping.
        (With rt.setExternalFramebuffer(self.array) the renderer's device-to-host copies land here directly
        and this call is not needed.)"""
        ny = fb.shape[0]
        for k in range(rank, (ny + stripe - 1) // stripe, world):
            self.array[k * stripe:k * stripe + stripe] = fb[k * stripe:k * stripe + stripe]

    def close(self, barrier):
        barrier()
        del self.array
        if self.rank == 0:
            try:
                os.unlink(self.path)
            except OSError:
                pass
