"""Reads bench.py JSON lines on stdin and prints `value ms_per_step frame_ms_kernel` for each (sweep helper)."""
import json
import sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print("%.1f Msamples/s  %.3f ms/step  kernel %.3f ms" % (d["value"], d["ms_per_step"], d.get("frame_ms_kernel", float("nan"))))
