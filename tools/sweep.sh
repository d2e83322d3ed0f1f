#!/bin/bash
# tools/sweep.sh "<variant list>" [fp] — bench.py over kernel variants in one go (A/B on one device)
FP=${2:-parity}
EXTRA=${3:-}
for v in $1; do
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --fp $FP --variant $v $EXTRA 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('variant', $v, 'fp', '$FP', 'Msamples/s %.1f' % d['value'], 'kernel_ms %.2f' % d['frame_ms_kernel'], 'frac %.3f' % d['roofline']['frac'], 'exec tests/ray %.0f' % (d.get('executed_sphere_tests_per_ray_rank0') or 0))"
done
