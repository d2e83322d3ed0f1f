"""From <RT_WAVE_DEBUG>.px: how well does the phase-1 estimate (3x3-smoothed rays of the first 2 samples) predict a pixel's total rays?"""
import sys
import numpy as np
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
a = np.fromfile(sys.argv[1], dtype=np.float32).reshape(-1, 4)
ny = len(a) // nx
rays = a[:, 2].reshape(ny, nx)
p1 = a[:, 3].reshape(ny, nx)
pad = np.pad(p1, 1, mode="constant")
cnt = np.pad(np.ones_like(p1), 1, mode="constant")
s = sum(pad[1 + dy:1 + dy + ny, 1 + dx:1 + dx + nx] for dy in (-1, 0, 1) for dx in (-1, 0, 1))
n = sum(cnt[1 + dy:1 + dy + ny, 1 + dx:1 + dx + nx] for dy in (-1, 0, 1) for dx in (-1, 0, 1))
est = s / (n * 2.0)          # rays per sample, smoothed
own = p1 / 2.0
print("total rays %.0f" % rays.sum())
for thr in (500, 700, 1000, 1500, 2000, 3000):
    sel = rays > thr
    print("pixels with > %4d rays: %6d (rays in them %.1f %%) | est: p10 %.1f median %.1f p90 %.1f | own: median %.1f" %
          (thr, sel.sum(), 100 * rays[sel].sum() / rays.sum(), *np.percentile(est[sel], [10, 50, 90]), np.median(own[sel])))
for lo, hi in [(1, 1.06), (1.06, 1.5), (1.5, 2.25), (2.25, 3.5), (3.5, 6), (6, 8), (8, 10), (10, 12), (12, 15), (15, 20), (20, 100)]:
    sel = (est >= lo) & (est < hi)
    if sel.any():
        r = rays[sel]
        print("est in [%5.2f,%5.2f): %7d px, rays mean %6.0f p50 %6.0f p90 %6.0f p99 %6.0f max %6.0f" % (lo, hi, sel.sum(), r.mean(), *np.percentile(r, [50, 90, 99]), r.max()))
