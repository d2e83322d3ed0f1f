# tools/mesh_tail_diag.py <file>: reads the per-pixel (start, end, cost) records a -DRT_MESH_TAIL_DIAG=1 build writes (RT_MESH_DIAG_FILE=<file>) for the second
# dispatch of the mesh frame: who ends last, when did they start, how long do the costliest pixels run.  (tools/build_variant.sh taildiag -DRT_MESH_TAIL_DIAG=1)
import numpy as np, sys
a = np.fromfile(sys.argv[1], dtype=np.uint32); n = a.size // 3
se = a[:2 * n].reshape(n, 2).astype(np.int64); cost = a[2 * n:].astype(np.int64)
t0 = se[:, 0].min(); st = (se[:, 0] - t0) / 1000.0; en = (se[:, 1] - t0) / 1000.0; du = en - st
print("pixels", n, "frame end %.1f ms" % en.max())
print("cost (phase-1 'rays'): mean %.1f p50 %d p90 %d p99 %d max %d" % (cost.mean(), *np.percentile(cost, [50, 90, 99]), cost.max()))
print("item duration ms: mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % (du.mean(), *np.percentile(du, [50, 90, 99]), du.max()))
late = np.argsort(-en)[:2000]
print("the 2000 last-ending pixels: start ms p10/p50/p90 = %.1f %.1f %.1f, duration p10/p50/p90 = %.1f %.1f %.1f, cost p10/p50/p90 = %d %d %d" % (*np.percentile(st[late], [10, 50, 90]), *np.percentile(du[late], [10, 50, 90]), *np.percentile(cost[late], [10, 50, 90])))
top = np.argsort(-cost)[:2000]
print("the 2000 costliest pixels: start ms p10/p50/p90 = %.1f %.1f %.1f, duration = %.1f %.1f %.1f, end = %.1f %.1f %.1f" % (*np.percentile(st[top], [10, 50, 90]), *np.percentile(du[top], [10, 50, 90]), *np.percentile(en[top], [10, 50, 90])))
for lo, hi in [(0, 50), (50, 150), (150, 300), (300, 400), (400, 600)]:
    m = (st >= lo) & (st < hi) & (cost > 0)
    if m.sum(): print("started in [%d,%d) ms: %d pixels, mean cost %.1f, mean duration %.2f ms, us per cost unit %.1f" % (lo, hi, m.sum(), cost[m].mean(), du[m].mean(), 1000 * du[m].sum() / cost[m].sum()))
for q in [50, 90, 99, 99.9]:
    th = np.percentile(cost, q); m = cost >= th
    print("cost >= p%s (%d): n %d, duration mean %.1f max %.1f ms; start mean %.1f; end max %.1f" % (q, th, m.sum(), du[m].mean(), du[m].max(), st[m].mean(), en[m].max()))
m = st < 100
print("corr(cost, duration | start<100ms) = %.3f" % np.corrcoef(cost[m], du[m])[0, 1])
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
for T in ([430, 450, 470, 490, 510, 530] if len(sys.argv) <= 2 else [round(en.max() * f, 1) for f in (0.7, 0.8, 0.85, 0.9, 0.95)]):
    m = en > T
    print("ending after %d ms: %d pixels; of them started before 10 ms: %d; start p50 %.1f; cost p50 %d" % (T, m.sum(), (m & (st < 10)).sum(), np.median(st[m]) if m.sum() else -1, np.median(cost[m]) if m.sum() else -1))
m = en > (470 if len(sys.argv) <= 2 else en.max() * 0.9)
idx = np.nonzero(m)[0][np.argsort(-en[m])][:40]
for i in idx: print("  px (%4d,%4d) start %.1f dur %.1f end %.1f cost %d" % (i % nx, i // nx, st[i], du[i], en[i], cost[i]))
