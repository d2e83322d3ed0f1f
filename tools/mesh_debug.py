"""Reads the RT_WAVE_DEBUG dump of the mesh kernel (16 counters summed over the waves) and prints where the cycles and lanes go."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64)[:16].astype(np.float64)
names = ["process", "refill", "node loop", "leaf"]
cyc, act, it = a[0:4], a[4:8], a[8:12]
print("waves", int(a[12]), "cycles per wave %.3g" % (cyc.sum() / a[12]))
for k in range(4):
    print("%-10s %5.1f %% of cycles | steps %.3g | active lanes per step %.1f | cycles per step %.0f" %
          (names[k], 100 * cyc[k] / cyc.sum(), it[k], act[k] / max(it[k], 1), cyc[k] / max(it[k], 1)))
