"""Reads <RT_WAVE_DEBUG>.px (per pixel: grabbed ms, finished ms, rays in total, rays in phase 1) and prints who ends the frame."""
import sys
import numpy as np
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
a = np.fromfile(sys.argv[1], dtype=np.float32).reshape(-1, 4)
grab, fin, rays, p1 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
print("pixels", len(a), "frame end %.2f ms" % fin.max(), "rays: mean %.0f max %.0f" % (rays.mean(), rays.max()))
order = np.argsort(fin)[::-1]
print("last pixels to finish:")
for k in order[:25]:
    print("  px (%4d,%4d) grabbed %6.2f finished %6.2f  rays %5.0f  phase-1 rays %3.0f  us/ray %.2f" %
          (k % nx, k // nx, grab[k], fin[k], rays[k], p1[k], 1e3 * (fin[k] - grab[k]) / max(rays[k] - p1[k], 1)))
for lo, hi in [(0, 1), (1, 5), (5, 10), (10, 15), (15, 20), (20, 21), (21, 22), (22, 30)]:
    sel = (grab >= lo) & (grab < hi)
    if sel.any():
        print("grabbed in [%2d,%2d) ms: %7d px, rays mean %6.0f p99 %6.0f max %6.0f, duration mean %.2f max %.2f ms, us/ray mean %.1f" %
              (lo, hi, sel.sum(), rays[sel].mean(), np.percentile(rays[sel], 99), rays[sel].max(), (fin - grab)[sel].mean(), (fin - grab)[sel].max(),
               1e3 * ((fin - grab)[sel].sum() / np.maximum(rays - p1, 1)[sel].sum())))
print("longest pixels:")
for k in np.argsort(rays)[::-1][:8]:
    print("  px (%4d,%4d) grabbed %6.2f finished %6.2f  rays %5.0f  phase-1 rays %3.0f  us/ray %.2f" % (k % nx, k // nx, grab[k], fin[k], rays[k], p1[k], 1e3 * (fin[k] - grab[k]) / max(rays[k] - p1[k], 1)))
late = fin > np.percentile(fin, 99.9)
print("latest 0.1%% pixels: grabbed mean %.2f (min %.2f max %.2f), rays mean %.0f, phase-1 rays mean %.1f" % (grab[late].mean(), grab[late].min(), grab[late].max(), rays[late].mean(), p1[late].mean()))
