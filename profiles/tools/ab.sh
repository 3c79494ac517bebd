#!/bin/bash
# A/B of prebuilt library variants on ONE box: raytracer_amd/librtgpu_<NAME>.so are copied over librtgpu.so in turn.
# Usage: bash profiles/tools/ab.sh NAME1 NAME2 ... (each is measured twice, interleaved; kernel alone + whole step)
cd "${GRAFT_REPO_ROOT:-.}"
cp raytracer_amd/librtgpu.so /tmp/librtgpu_keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp raytracer_amd/librtgpu_$v.so raytracer_amd/librtgpu.so
    echo -n "$v "; bash profiles/tools/c4_quick.sh RT_NO_LOOKAHEAD=1 || exit 1
    echo -n "$v "; STEPS=12 bash profiles/tools/c4_long.sh || exit 1
  done
done
cp /tmp/librtgpu_keep.so raytracer_amd/librtgpu.so
