#!/bin/bash
# tools/profile_mesh.sh <tag> — tools/profile.sh for the mesh kernel (tools/bench_mesh.py, staircase 1920x1080x32spp)
PROF_PROG=tools/bench_mesh.py bash "$(dirname "$0")/profile.sh" "$1" --spp 32 --steps 2
