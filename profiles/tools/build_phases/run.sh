#!/bin/bash
# Where one thread's build of one 32,768-triangle mesh spends its cycles (the multi-frame job's unit of host work).
# Builds an instrumented copy of the builder (rdtsc around each phase) with a small main() and runs it three times;
# counts are thousands of TSC cycles per build, averaged over 20 builds after 10 warm-ups.
# Usage: profiles/tools/build_phases/run.sh [rt_bvh.cpp] [grid] [soup]   (default: the tree's own source, 128 -> 32,768
# triangles of a plate in row order; "soup": as many small triangles at random places)
set -e
here="$(cd "$(dirname "$0")" && pwd)"
src="${1:-$here/../../../raytracer_amd/csrc/rt_bvh.cpp}"
tmp="$(mktemp -d)"
cp "$here/../../../raytracer_amd/csrc/rt_bvh.hpp" "$here/main.cpp" "$tmp/"
python3 "$here/instrument.py" "$tmp/cur.cpp" "$src"
/opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I"$tmp" -o "$tmp/prof" "$tmp/main.cpp" "$tmp/cur.cpp" -lpthread
for i in 1 2 3; do RT_HOST_THREADS=1 "$tmp/prof" ${2:-128} $3; echo; done
rm -rf "$tmp"
