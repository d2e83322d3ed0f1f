"""Multi-GPU plumbing for one node: the image is cut into interleaved row stripes, one process per GPU renders
its stripes, and the stripes are gathered ON THE HOST into one framebuffer shared by the ranks (a memory-mapped
file in /dev/shm) — no collective on the data path.  torch.distributed is used only for the
barrier / max-over-ranks timing the bench contract asks for (bench.py run_job).

Nothing here touches the GPU: the same code runs in the world-size-2 gloo tests on CPU."""
import os

import numpy as np

STRIPE_ROWS = 8


def stripe_rows(rank, world, ny, stripe=STRIPE_ROWS):
    """Global row indices owned by `rank`: stripes k = rank, rank+world, ... of `stripe` rows each
    (the same partition rt_render_options.part_rank/part_world/stripe_rows select in the renderer)."""
    nstripes = (ny + stripe - 1) // stripe
    rows = [np.arange(k * stripe, min(ny, k * stripe + stripe)) for k in range(rank, nstripes, world)]
    return np.concatenate(rows) if rows else np.zeros(0, np.int64)


class SharedFramebuffer:
    """One (ny, nx, 3) float32 framebuffer visible to every rank of the node: a raw (header-less) file in /dev/shm,
    so the mapping — which every rank page-locks with hipHostRegister (setExternalFramebuffer) — starts on a page."""

    def __init__(self, tag, nx, ny, rank, barrier):
        self.path = f"/dev/shm/rt_fb_{tag}.raw"
        self.rank = rank
        self.array = None
        if rank == 0:
            m = np.memmap(self.path, dtype=np.float32, mode="w+", shape=(ny, nx, 3))
            m.flush()
            del m
        barrier()
        self.array = np.memmap(self.path, dtype=np.float32, mode="r+", shape=(ny, nx, 3))
        assert self.array.ctypes.data % 4096 == 0

    def gather(self, fb, rank, world, stripe=STRIPE_ROWS):
        """Host-side gather of this rank's stripes: one contiguous memcpy per stripe into the shared mapping.
        (With rt.setExternalFramebuffer(self.array) the renderer's device-to-host copies land here directly
        and this call is not needed.)"""
        ny = fb.shape[0]
        for k in range(rank, (ny + stripe - 1) // stripe, world):
            self.array[k * stripe:k * stripe + stripe] = fb[k * stripe:k * stripe + stripe]

    def close(self, barrier):
        try:
            barrier()
        finally:
            self.array = None
            if self.rank == 0:
                try:
                    os.unlink(self.path)
                except OSError:
                    pass
