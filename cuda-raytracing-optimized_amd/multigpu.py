"""Multi-GPU plumbing for one node: the image is cut into interleaved row stripes, one process per GPU renders
its stripes, and the stripes are gathered ON THE HOST into one framebuffer shared by the ranks (a memory-mapped
file in /dev/shm) — no collective on the data path.  The control path - the barrier and the max / sum over
ranks of the timing figures the bench contract asks for (bench.py run_job) - goes through /dev/shm as well (ShmComm below):
neither path needs RCCL, as BASELINE.json's north star states ("host-side gather, no RCCL").

Nothing here touches the GPU: the same code runs in the multi-process tests on CPU."""
import os
import struct
import time

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


class ShmComm:
    """barrier() and reduce(values, "max" | "sum") over the ranks of ONE node through a small file in /dev/shm: no RCCL, no sockets.

    Layout: a 64-byte header (magic, world) and one 4 KB slot per rank: slot[0] = the rank's barrier generation, slot[1] = number of values,
    slot[2..] = its float64 values, slot[511] = its process id.  Every word has exactly one writer (its rank), so no atomic read-modify-write is
    needed: a barrier is "bump my generation, wait until every rank's generation has reached mine".  A rank that waits checks twice a second that
    the ranks it waits for are still alive (the pid in their slot: gone, or a zombie, means the rank died - the renderer ends the process on a HIP
    error, kernels.cu:27-38 - and no generation will ever come): it raises PeerDied at once instead of sitting out the timeout, so that the job ends
    non-zero with a message whatever the launcher does about the other ranks.  The timeout (480 s) stays below the driver's 600 s limit for a bench run.  The name carries MASTER_PORT and the parent process id,
    which the ranks of one launch share (torch.distributed.run's agent, or the test's spawning process), so a launch never attaches to the
    file of another one; rank 0 creates the file under a temporary name and renames it into place, the others wait for it."""

    MAGIC = 0x52544D4D          # "RTMM"
    SLOT = 4096
    PID_WORD = 511

    class PeerDied(RuntimeError):
        pass

    def __init__(self, rank, world, tag=None, timeout=480.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        tag = tag if tag is not None else f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        self.path = f"/dev/shm/rt_comm_{tag}.bin"
        size = 64 + self.SLOT * self.world
        if self.rank == 0:
            tmp = self.path + f".tmp{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(struct.pack("<II", self.MAGIC, self.world) + bytes(size - 8))
            os.replace(tmp, self.path)
        else:
            t0 = time.monotonic()
            fresh = time.time() - 600.0                         # (a file left behind by a crashed job of long ago is not this job's)

            def ready():
                try:
                    st = os.stat(self.path)
                except OSError:
                    return False
                return st.st_size == size and st.st_mtime >= fresh
            while not ready():
                if time.monotonic() - t0 > self.timeout:
                    raise TimeoutError(f"ShmComm: rank {self.rank} waited {self.timeout} s for {self.path}")
                time.sleep(0.001)
        self.mm = np.memmap(self.path, dtype=np.uint8, mode="r+", shape=(size,))
        magic, w = struct.unpack("<II", bytes(self.mm[:8]))
        if magic != self.MAGIC or w != self.world:
            raise RuntimeError(f"ShmComm: {self.path} belongs to another job (magic {magic:#x}, world {w})")
        self.slots = [self.mm[64 + self.SLOT * r:64 + self.SLOT * (r + 1)].view(np.float64) for r in range(self.world)]
        self.slots[self.rank][self.PID_WORD] = float(os.getpid())
        self.gen = 0
        self.barrier()

    @staticmethod
    def _alive(pid):
        """False when the process is gone or a zombie (exited, not yet reaped by its launcher)."""
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        except PermissionError:
            return True
        try:
            with open(f"/proc/{pid}/stat", "rb") as f:
                return f.read().rsplit(b")", 1)[1].split()[0] != b"Z"
        except OSError:
            return True

    def _check_peers(self, behind):
        for r in behind:
            pid = int(self.slots[r][self.PID_WORD])
            if pid > 0 and not self._alive(pid):
                raise ShmComm.PeerDied(f"ShmComm: rank {r} (pid {pid}) died before barrier {self.gen}; rank {self.rank} gives up")

    def barrier(self):
        self.gen += 1
        self.slots[self.rank][0] = float(self.gen)
        t0 = time.monotonic()
        spins = 0
        next_check = t0 + 0.5
        while True:
            if all(s[0] >= self.gen for s in self.slots):
                return
            spins += 1
            if spins > 2000:                                    # a rank that renders for seconds: stop burning a core
                time.sleep(0.0002)
                now = time.monotonic()
                if now >= next_check:
                    next_check = now + 0.5
                    behind = [r for r, s in enumerate(self.slots) if s[0] < self.gen]
                    self._check_peers(behind)
                    if now - t0 > self.timeout:
                        raise TimeoutError(f"ShmComm: barrier {self.gen} timed out after {self.timeout} s waiting for ranks {behind}")

    def reduce(self, values, op):
        vals = [float(v) for v in values]
        assert len(vals) <= self.PID_WORD - 2
        mine = self.slots[self.rank]
        mine[1] = float(len(vals))
        mine[2:2 + len(vals)] = vals
        self.barrier()                                          # everybody has written
        cols = np.array([s[2:2 + len(vals)] for s in self.slots])
        assert all(int(s[1]) == len(vals) for s in self.slots)
        out = (cols.max(axis=0) if op == "max" else cols.sum(axis=0)).tolist()
        self.barrier()                                          # everybody has read: the slots may be rewritten
        return out

    def close(self):
        try:
            self.barrier()
        except (TimeoutError, ShmComm.PeerDied):
            pass                                                # (a dead peer was reported where it was first seen)
        finally:
            self.slots = None
            self.mm = None
            if self.rank == 0:
                try:
                    os.unlink(self.path)
                except OSError:
                    pass
