# tools/sweep_scene_sizes_ab.sh lib... — tools/sweep_scene_sizes.py under each build of librt_mi355x.so (one GPU call)
for L in "$@"; do echo "== $L"; RT_LIB=$L python3 - <<'PY'
import os, sys, runpy
sys.path.insert(0, os.getcwd())
import cuda_raytracing_optimized_amd as rt
rt.RENDERER_LIB = os.path.abspath(os.environ["RT_LIB"])
runpy.run_path("tools/sweep_scene_sizes.py", run_name="__main__")
PY
done
