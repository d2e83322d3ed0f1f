"""Reads the RT_WAVE_DEBUG dump (8 x u64 per wave) and prints a time line of the persistent kernel."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
sec, steps = a[65534].astype(np.float64), a[65533].astype(np.float64)
csec, cspa = a[65532].astype(np.float64), a[65531].astype(np.float64)     # the same for the waves with >= 2000 sparse-form iterations
a = a[:65531]
a = a[a[:, 7] == 1]
t0 = a[:, 0].min()
start = (a[:, 0] - t0) / 1e5      # ms (100 MHz)
tex = np.where(a[:, 1] > 0, (a[:, 1].astype(np.int64) - int(t0)) / 1e5, np.nan)
end = (a[:, 2] - t0) / 1e5
print("waves", len(a), "kernel span ms %.2f" % end.max())
print("start   ms: min %.2f max %.2f" % (start.min(), start.max()))
print("exhaust ms: min %.2f median %.2f max %.2f" % (np.nanmin(tex), np.nanmedian(tex), np.nanmax(tex)))
print("end     ms: p10 %.2f median %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(end, [10, 50, 90, 99, 100])))
print("iterations per wave: mean %.0f max %d | sparse-form iterations: mean %.0f max %d | sparse-form rays mean %.0f max %d" %
      (a[:, 3].mean(), a[:, 3].max(), a[:, 4].mean(), a[:, 4].max(), a[:, 5].mean(), a[:, 5].max()))
hist, edges = np.histogram(end, bins=20)
for h, e in zip(hist, edges):
    print("  end in [%6.2f, ..) ms: %5d waves" % (e, h))
late = np.argsort(end)[-5:]
for w in late:
    print("  late wave: end %.2f exhausted %.2f iters %d sparse_iters %d sparse_rays %d" % (end[w], tex[w], a[w, 3], a[w, 4], a[w, 5]))

early = end < 1.0
print("early waves:", early.sum(), "their iters: mean %.1f max %d" % (a[early, 3].mean() if early.any() else 0, a[early, 3].max() if early.any() else 0))
hw = a[:, 6]
xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
hwid = hw & np.uint64(0xFFFFFFFF)
# gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] (se_id 3 bits on gfx90a+)
cu = (hwid >> np.uint64(8)) & np.uint64(0xF)
sh = (hwid >> np.uint64(12)) & np.uint64(0x1)
se = (hwid >> np.uint64(13)) & np.uint64(0x7)
simd = (hwid >> np.uint64(4)) & np.uint64(0x3)
key = (xcc.astype(np.int64) * 8 + se.astype(np.int64)) * 32 + sh.astype(np.int64) * 16 + cu.astype(np.int64)
uk, cnt = np.unique(key, return_counts=True)
print("distinct CUs used:", len(uk), "waves per CU: min %d max %d" % (cnt.min(), cnt.max()), "histogram", np.bincount(cnt))
for c in sorted(set(cnt)):
    sel = np.isin(key, uk[cnt == c])
    print("  CUs with %2d waves: %4d waves, median end %.1f ms, median exhausted %.1f" % (c, sel.sum(), np.median(end[sel]), np.nanmedian(tex[sel]) if np.isfinite(tex[sel]).any() else -1))
print("per XCD (XCC_ID): waves, the moment a wave first found every queue empty (median), end of its waves (median / max) - one work queue per XCD since round 4:")
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print("  XCD %d: %4d waves, exhausted %.2f ms, end median %.2f max %.2f" % (x, m.sum(), np.nanmedian(tex[m]) if np.isfinite(tex[m]).any() else -1, np.median(end[m]), end[m].max()))

if sec.sum() > 0:
    names = ["ray set-up (unit dir)", "big spheres", "group boxes + pair list", "pair rounds (phase 1)", "candidates (exact tail)", "shade", "sparse scan", "refill / start pixel+sample"]
    print("section cycles, all waves (share of the instrumented total):")
    for n_, c in zip(names, sec):
        print("  %-30s %5.1f %%" % (n_, 100 * c / sec.sum()))
    print("  main steps %.3g, boost steps %.3g" % (steps[0], steps[1]))

if csec.sum() > 0 and cspa[5] > 0:
    names = ["ray set-up (unit dir)", "big spheres (dense)", "group boxes + pair list (dense)", "pair rounds (dense)", "candidates (dense)", "shade", "sparse scan (old slot)", "refill / start pixel+sample"]
    tot = cspa[6]
    print("waves with >= 2000 sparse-form iterations: %d, %.0f iterations each, %.0f cycles of wave life per iteration" % (cspa[5], cspa[4] / cspa[5], tot / cspa[4]))
    for n_, c in zip(names, csec):
        print("  %-34s %5.1f %% of their life" % (n_, 100 * c / tot))
    for n_, c in zip(["sparse: ray table + box set-up", "sparse: big spheres", "sparse: group boxes", "sparse: sphere tests + read-back"], cspa[:4]):
        print("  %-34s %5.1f %% of their life" % (n_, 100 * c / tot))
