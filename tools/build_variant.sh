#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags...] — builds librt_mi355x.so from the working tree with extra -D flags into build/ab/<name>.so
# (objects under build/ab/obj_<name>/): the A/B unit of tools/ab_spheres.sh / tools/ab_mesh.sh.  The tree's own library is not touched.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
O=build/ab/obj_$NAME
mkdir -p $O
make -s -j8 OBJ=$O HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $*" \
     $O/renderer.o $O/probe_parity.o $O/probe_fast.o $O/spheres_parity.o $O/spheres_fast.o $O/mesh_parity.o $O/mesh_fast.o
hipcc --offload-arch=gfx950 -shared -fPIC $O/renderer.o $O/probe_parity.o $O/probe_fast.o $O/spheres_parity.o $O/spheres_fast.o $O/mesh_parity.o $O/mesh_fast.o -o build/ab/$NAME.so
rm -rf $O
echo "built build/ab/$NAME.so"
