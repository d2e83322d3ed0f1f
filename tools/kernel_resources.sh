#!/bin/bash
# tools/kernel_resources.sh [spheres|mesh] [parity|fast] — register / spill / LDS figures of every kernel of one TU, read from
# the code object's own metadata (device-only assembly; the numbers rocprofv3's trace prints for VGPR/LDS are not these).
set -e
TU=${1:-spheres}; MODE=${2:-parity}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${3:-/tmp/kres_${TU}_${MODE}.s}
DEF=-DRT_MODE_PARITY; FP="-ffp-contract=off"
if [ "$MODE" = fast ]; then DEF=-DRT_MODE_FAST; FP="-ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt"; fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 $EXTRA $DEF $FP -fno-slp-vectorize -fno-vectorize --cuda-device-only -S \
    "$ROOT/cuda-raytracing-optimized_amd/csrc/rt_kernels_${TU}.hip" -o "$OUT"
python3 - "$OUT" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
meta = txt[txt.rfind("amdhsa.kernels:"):]
for blk in re.split(r"\n  - ", meta)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    body = txt[txt.find(name + ":"):]
    body = body[:body.find(".end_amdhsa_kernel") if ".end_amdhsa_kernel" in body else len(body)]
    cnt = lambda pat: len(re.findall(pat, body))
    print(f"{name[:110]}\n    vgpr {g('vgpr_count')} (spill {g('vgpr_spill_count')})  sgpr {g('sgpr_count')} (spill {g('sgpr_spill_count')})  "
          f"lds {g('group_segment_fixed_size')}  scratch {g('private_segment_fixed_size')}  "
          f"v_readlane/v_writelane {cnt(r'v_(read|write)lane_b32')}  s_waitcnt {cnt(r's_waitcnt')}  lines {body.count(chr(10))}")
PY
